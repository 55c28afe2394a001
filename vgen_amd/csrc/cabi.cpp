// cabi.cpp — error plumbing + version for libvgen_hip.so (see include/vgen_hip.h).
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "../../include/vgen_hip.h"

static thread_local char g_err[512] = "ok";

void vgen_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int vgen_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    vgen_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

// per-device launch state (common.h)
int vgen_device_slot() {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0) d = 0;
  return d < 16 ? d : 15;
}

int vgen_device_cus() {
  static int cus[16] = {0};
  const int d = vgen_device_slot();
  if (cus[d] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || n <= 0) n = 256;
    cus[d] = n;
  }
  return cus[d];
}

extern "C" int vgen_version(void) { return VGEN_ABI_VERSION; }
extern "C" const char* vgen_last_error(void) { return g_err; }
