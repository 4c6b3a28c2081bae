// Register layout probe of v_mfma_f32_4x4x4_16b_bf16 (16 blocks of 4x4x4) on gfx950: A one-hot at (lane la, element ea), B[lane][e] = lane * 4 + e + 1 -> which
// (lane, reg) of D receive which B element.  Build: hipcc --offload-arch=gfx950 -O2 mfma4_probe.hip -o mfma4_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__device__ short bf(float f) { return (short)(__float_as_uint(f) >> 16); }
__global__ void k(int la, int ea, float* out) {
  const int l = threadIdx.x;
  s16x4 a = {0, 0, 0, 0}, b;
  if (l == la) a[ea] = bf(1.0f);
  for (int e = 0; e < 4; ++e) b[e] = bf((float)(l * 4 + e + 1));
  f32x4 d = {0, 0, 0, 0};
  d = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a, b, d, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = d[r];
}
int main() {
  float* d; hipMalloc(&d, 256 * 4); float h[256];
  for (int la : {0, 1, 2, 5, 62}) for (int ea = 0; ea < 4; ++ea) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, la, ea, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("A one-hot lane %2d elem %d ->", la, ea);
    for (int i = 0; i < 256; ++i) if (h[i] != 0.f) printf("  D[lane %2d][reg %d] = B[lane %d][elem %d]", i / 4, i % 4, ((int)h[i] - 1) / 4, ((int)h[i] - 1) % 4);
    printf("\n");
  }
  return 0;
}
