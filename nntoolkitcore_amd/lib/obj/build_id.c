const char *nntk_build_source_hash(void) { return "726e45298310815f"; }
