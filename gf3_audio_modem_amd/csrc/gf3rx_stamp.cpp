// libgf3rx -- the build stamp: SHA-256 of the sources and flags this binary was built from
// (gf3_audio_modem_amd/build.py passes it in; "unknown" for a build made by hand).  The loader compares it with the
// sources as they are now -- the library carries its own stamp, no side file -- and finds it by the marker without
// loading the library.  Its own translation unit, so that a change anywhere recompiles this file and nothing else.
#include "gf3rx.h"

#ifndef GF3_SRC_HASH
#define GF3_SRC_HASH "unknown"
#endif
extern "C" const char gf3_src_hash_marker[] = "GF3_SRC_HASH=" GF3_SRC_HASH;
extern "C" const char* gf3_source_hash(void) { return gf3_src_hash_marker + 13; }
extern "C" const char* gf3_version(void) { return GF3RX_VERSION; }
