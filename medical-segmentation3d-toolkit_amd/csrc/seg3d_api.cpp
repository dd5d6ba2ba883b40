// seg3d_api.cpp -- library-level entry points of libseg3d_hip.so: error reporting and version/introspection.
// The compute entry points live next to their kernels (*.hip); every one of them is declared in
// include/seg3d_hip.h, returns 0 on success and a negative code + thread-local message on failure.
#include "seg3d_common.h"
#include "seg3d_hip.h"
#include <string.h>

static thread_local char g_err[512] = "";

void seg3d_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* seg3d_last_error(void) { return g_err; }

extern "C" int seg3d_abi_version(void) { return SEG3D_ABI_VERSION; }

extern "C" const char* seg3d_target_arch(void) { return "gfx950"; }

// Number of HIP devices visible, or a negative error code; performs no other GPU work.
extern "C" int seg3d_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    seg3d_set_error("hipGetDeviceCount failed: %s", hipGetErrorString(e));
    return SEG3D_ERR_LAUNCH;
  }
  return n;
}
