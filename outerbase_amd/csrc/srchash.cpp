// Content hashes of the library's sources, set by the Makefile at build time.
#include "../../include/obhip.h"

extern "C" const char *obhip_source_hash(int which) { return which == 1 ? OBHIP_HASH_GRAM : OBHIP_HASH_ALL; }
