// Error plumbing + ABI version of libvlaser_hip.so (no C++ exceptions cross the ABI).
#include <stdarg.h>
#include <stdio.h>

#include "../../include/vlaser_hip.h"

static thread_local char g_err[512] = "";

void vlaser_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* vlaser_last_error(void) { return g_err; }
extern "C" int vlaser_abi_version(void) { return 8; }

// ---- CU-masked streams (ABI 7) -- see include/vlaser_hip.h
#include <hip/hip_runtime.h>
#include <stdint.h>
extern "C" int vlaser_stream_create_cumask(int first_cu, int n_cus, vl_stream_t* out) {
  hipDeviceProp_t prop;
  int dev = 0;
  if (!out || hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
    vlaser_set_error("vlaser_stream_create_cumask: no device / null result pointer");
    return -1;
  }
  const int total = prop.multiProcessorCount;
  if (first_cu < 0 || n_cus < 1 || first_cu + n_cus > total || total > 1024) {
    vlaser_set_error("vlaser_stream_create_cumask: CUs [%d, %d) outside the device's %d", first_cu, first_cu + n_cus, total);
    return -1;
  }
  uint32_t mask[32] = {0};
  for (int i = first_cu; i < first_cu + n_cus; ++i) mask[i >> 5] |= 1u << (i & 31);
  hipStream_t s = nullptr;
  const hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)((total + 31) / 32), mask);
  if (e != hipSuccess) {
    vlaser_set_error("hipExtStreamCreateWithCUMask: %s", hipGetErrorString(e));
    return -1;
  }
  *out = (vl_stream_t)s;
  return 0;
}
extern "C" int vlaser_stream_destroy(vl_stream_t s) {
  const hipError_t e = hipStreamDestroy((hipStream_t)s);
  if (e != hipSuccess) { vlaser_set_error("hipStreamDestroy: %s", hipGetErrorString(e)); return -1; }
  return 0;
}
