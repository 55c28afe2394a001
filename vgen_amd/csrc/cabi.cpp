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

extern "C" int vgen_version(void) { return VGEN_ABI_VERSION; }
extern "C" const char* vgen_last_error(void) { return g_err; }
