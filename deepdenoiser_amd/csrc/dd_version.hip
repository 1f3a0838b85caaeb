// dd_version(): library name, target and the hash of the sources this binary was built from.
// deepdenoiser_amd/build.py passes -DDD_SOURCE_HASH="<first 16 hex digits of sha256 over csrc/* and include/dd_hip.h>"; deepdenoiser_amd/_lib.py
// recomputes the hash from the sources next to the library at load time and refuses a stale binary (the .so travels prebuilt to the GPU box).
#ifndef DD_SOURCE_HASH
#define DD_SOURCE_HASH "unknown"
#endif

extern "C" const char* dd_version(void) { return "libdd_hip 0.3 (gfx950) src=" DD_SOURCE_HASH; }
