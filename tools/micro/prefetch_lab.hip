// Lab: a paced "touch" of a buffer (reads it with few workgroups, keeps nothing) -- pulls expert weights into the Infinity Cache ahead of the Euler phase's launches.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o libprefetch_lab.so prefetch_lab.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
// progress: optional device int32 the compute chain advances (layer-steps completed); the toucher of item `idx` waits until progress >= idx - ahead (bounded spin)
__global__ __launch_bounds__(256) void touch_kernel(const u32x4* __restrict__ p, long long n16, unsigned* __restrict__ sink, const volatile int* progress, int idx, int ahead,
                                                    long long spin_ticks) {
  if (progress) {
    if (threadIdx.x == 0) {
      const long long t0 = wall_clock64();
      while (*progress < idx - ahead && wall_clock64() - t0 < spin_ticks) __builtin_amdgcn_s_sleep(32);
    }
    __syncthreads();
  }
  unsigned acc = 0;
  const long long st = (long long)gridDim.x * 256;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * st < n16; i += 4 * st) {          // four 16-byte loads in flight per lane
    const u32x4 v0 = p[i], v1 = p[i + st], v2 = p[i + 2 * st], v3 = p[i + 3 * st];
    acc ^= v0[0] ^ v1[1] ^ v2[2] ^ v3[3];
  }
  for (; i < n16; i += st) acc ^= p[i][0];
  if (acc == 0x12345678u) sink[0] = acc;        // never true in practice: keeps the loads
}
extern "C" int touch(const void* p, long long bytes, int n_wg, unsigned* sink, const int* progress, int idx, int ahead, long long spin_ticks, hipStream_t s) {
  hipLaunchKernelGGL(touch_kernel, dim3(n_wg), dim3(256), 0, s, (const u32x4*)p, bytes / 16, sink, progress, idx, ahead, spin_ticks);
  return (int)hipGetLastError();
}
__global__ void bump_kernel(int* progress, int v) { if (threadIdx.x == 0) *progress = v; }
extern "C" int bump(int* progress, int v, hipStream_t s) { hipLaunchKernelGGL(bump_kernel, dim3(1), dim3(64), 0, s, progress, v); return (int)hipGetLastError(); }
