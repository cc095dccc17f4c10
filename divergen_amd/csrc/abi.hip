#include "dgx_common.h"
extern "C" const char* dgx_build_arch(void) { return "gfx950"; }
extern "C" int dgx_abi_version(void) { return 3; }
