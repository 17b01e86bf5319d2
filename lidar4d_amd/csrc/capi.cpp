// Error reporting + version of the C ABI (include/lidar4d_hip.h).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#include "../../include/lidar4d_hip.h"

static thread_local char g_err[512] = "";

extern "C" void l4d_set_error(int code, const char* where) {
  const char* txt = code > 1 ? hipGetErrorString((hipError_t)code) : "invalid argument";
  snprintf(g_err, sizeof(g_err), "%s: %s (code %d)", where, txt, code);
}

extern "C" const char* l4d_last_error(void) { return g_err; }

extern "C" int l4d_version(void) { return L4D_ABI_VERSION; }
