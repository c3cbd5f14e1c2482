const char *nntk_build_source_hash(void) { return "297ae9bd0d034931"; }
