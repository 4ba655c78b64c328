// ABI bookkeeping for libvist3a_hip.so
#include "common.h"
#include "../../include/vist3a_hip.h"
extern "C" int v3a_abi_version(void) { return 21; }
extern "C" const char* v3a_build_info(void) { return "libvist3a_hip gfx950 " __DATE__ " " __VERSION__; }
