// Library identity entry points.
#include "sei_common.h"
#include <string.h>

extern "C" int sei_abi_version(void) { return SEI_ABI_VERSION; }

extern "C" int sei_build_target(char *name, int n) {
    SEI_REQUIRE(name && n > 0);
    strncpy(name, "gfx950", (size_t)n);
    name[n - 1] = 0;
    return SEI_OK;
}
