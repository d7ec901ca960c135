// Library identity for the C-ABI (include/cetpick_hip.h).
#include "common.h"

extern "C" int mi_abi_version(void) { return 1; }
extern "C" const char* mi_build_arch(void) { return "gfx950"; }
