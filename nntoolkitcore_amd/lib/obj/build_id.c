const char *nntk_build_source_hash(void) { return "a8ef4e03155e491e"; }
