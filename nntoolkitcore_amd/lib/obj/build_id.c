const char *nntk_build_source_hash(void) { return "6e0eac64d3350386"; }
