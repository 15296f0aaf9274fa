// version.hip -- library identification.
#include "vfi_common.h"

extern "C" const char* vfi_version(void) { return "vfi_hip 0.1.0 gfx950"; }
