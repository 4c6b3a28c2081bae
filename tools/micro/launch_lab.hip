// Floor of a dependent kernel chain inside a HIP graph on this box: N launches of a trivial kernel (G workgroups x T threads), time per launch.
// Build: hipcc --offload-arch=gfx950 -O3 launch_lab.hip -o launch_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void trivial(float* p, int n) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.0f; }
__global__ void trivial_lds(float* p, int n) {
  extern __shared__ float sm[];
  const int i = blockIdx.x * blockDim.x + threadIdx.x; sm[threadIdx.x] = (float)i; __syncthreads();
  if (i < n) p[i] += sm[(threadIdx.x + 1) % blockDim.x];
}
int main() {
  hipStream_t s; CK(hipStreamCreate(&s));
  float* buf; CK(hipMalloc(&buf, 1 << 22)); CK(hipMemset(buf, 0, 1 << 22));
  CK(hipFuncSetAttribute((const void*)trivial_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
  const int N = 1000;
  struct Cfg { int g, t, lds; } cfgs[] = {{1, 64, 0}, {16, 256, 0}, {256, 256, 0}, {256, 512, 0}, {1025, 64, 0}, {256, 512, 64 * 1024}, {256, 512, 140 * 1024}, {2048, 256, 0}};
  for (auto c : cfgs) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < N; ++i) {
      if (c.lds) hipLaunchKernelGGL(trivial_lds, dim3(c.g), dim3(c.t), c.lds, s, buf, c.g * c.t);
      else hipLaunchKernelGGL(trivial, dim3(c.g), dim3(c.t), 0, s, buf, c.g * c.t);
    }
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("grid %5d x %4d threads, %3d KB LDS: %.2f us per launch in a dependent chain\n", c.g, c.t, c.lds / 1024, ms * 1e3 / (5 * N));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  return 0;
}
