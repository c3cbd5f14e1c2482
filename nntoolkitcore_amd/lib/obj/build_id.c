const char *nntk_build_source_hash(void) { return "163ccbb2551cf199"; }
