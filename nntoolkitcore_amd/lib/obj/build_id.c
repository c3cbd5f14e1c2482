const char *nntk_build_source_hash(void) { return "4338d8fc158faf39"; }
