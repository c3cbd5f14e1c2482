const char *nntk_build_source_hash(void) { return "aa70aa6219aa52c3"; }
