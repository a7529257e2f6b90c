// Library identification.
#include "resel_common.h"

extern "C" int resel_abi_version(void) { return 2; }
extern "C" const char* resel_build_info(void) { return "resel_hip gfx950 (CDNA4, wave64) built " __DATE__ " " __TIME__; }
