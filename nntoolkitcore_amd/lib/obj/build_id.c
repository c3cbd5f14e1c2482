const char *nntk_build_source_hash(void) { return "d237a3674eb64242"; }
