// Do kernels from two HIP streams overlap on this box?  A = HBM-bound streaming kernel (G blocks, grid-stride over 1 GiB),
// B = a chain of 200 short compute-bound kernels (256 blocks).  Times A alone, B alone, and A || B on two streams.
// Build: hipcc --offload-arch=gfx950 -O3 overlap_lab.hip -o overlap_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ __launch_bounds__(256) void stream_kernel(float4* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { float4 v = p[i]; v.x += 1.f; p[i] = v; }
}
__global__ __launch_bounds__(512) void spin_kernel(float* out, int iters) {
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  for (int i = 0; i < iters; ++i) a = a * b + 0.5f;
  if (a == 123.456f) out[0] = a;
}
int main() {
  hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  const size_t n = (size_t)1 << 26;      // 1 GiB of float4
  float4* buf; CK(hipMalloc(&buf, n * 16)); CK(hipMemset(buf, 0, n * 16));
  float* out; CK(hipMalloc(&out, 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timed = [&](const char* name, int G, bool runA, bool runB) {
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, s1));
      CK(hipStreamWaitEvent(s2, e0, 0));
      if (runA) for (int k = 0; k < 10; ++k) hipLaunchKernelGGL(stream_kernel, dim3(G), dim3(256), 0, s1, buf, n);
      if (runB) for (int k = 0; k < 400; ++k) hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(512), 0, s2, out, 1500);
      hipEvent_t eb; CK(hipEventCreate(&eb)); CK(hipEventRecord(eb, s2)); CK(hipStreamWaitEvent(s1, eb, 0));
      CK(hipEventRecord(e1, s1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
      CK(hipEventDestroy(eb));
    }
    printf("%-28s G=%5d : %.3f ms\n", name, G, best);
  };
  for (int G : {4096, 1024, 512, 256}) {
    timed("A alone (4 x 2 GiB traffic)", G, true, false);
    timed("B alone (200 spin kernels)", G, false, true);
    timed("A || B", G, true, true);
  }
  return 0;
}
