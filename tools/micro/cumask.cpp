// Lab helper: HIP streams restricted to a CU mask (hipExtStreamCreateWithCUMask), for tools/micro/cumask_lab.py.
// Build: hipcc -shared -fPIC cumask.cpp -o lab_build/libcumask.so
#include <hip/hip_runtime.h>
#include <stdint.h>
extern "C" int cumask_stream_create(const uint32_t* mask, int words, void** out) {
  hipStream_t s;
  hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask);
  if (e != hipSuccess) return (int)e;
  *out = (void*)s;
  return 0;
}
extern "C" int cumask_stream_destroy(void* s) { return (int)hipStreamDestroy((hipStream_t)s); }
