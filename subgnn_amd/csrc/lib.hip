// libsubgnn_hip.so: ABI version and error reporting.
#include "common.h"
#include <string.h>

static char g_last_error[256] = "";

void sgnn_set_last_error(hipError_t e) {
    strncpy(g_last_error, hipGetErrorString(e), sizeof(g_last_error) - 1);
    g_last_error[sizeof(g_last_error) - 1] = 0;
}

extern "C" int sgnn_abi_version(void) { return SGNN_ABI_VERSION; }
extern "C" const char* sgnn_last_error(void) { return g_last_error; }
