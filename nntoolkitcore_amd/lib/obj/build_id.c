const char *nntk_build_source_hash(void) { return "7e08cf5985201671"; }
