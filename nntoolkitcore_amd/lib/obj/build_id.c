const char *nntk_build_source_hash(void) { return "8e92769b938acdf1"; }
