// Probe for a fused (attention + o_proj) launch of the expert's layer-step: would G workgroups that EACH pull the whole kv group's K / V (199 KB, the
// same lines for every workgroup of a group: L2 hits after the first touch per XCD) + a 24 KB weight slice, reduce through LDS and store a few bytes
// fit in ~6 us, against attn_skinny + o_proj = 5.65 + 4.88 us today?  A producer kernel rewrites part of the K / V region first (as the qkv launch
// does), so the consumer's lines are cold in the other XCDs' L2s exactly as in the chain.
// Build: hipcc --offload-arch=gfx950 -O3 attn_o_probe.hip -o attn_o_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__global__ void producer(u32x4* kv, int n16) {          // 128 workgroups rewrite 4 rows' worth (8 KB) + touch nothing else
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n16) kv[i] = u32x4{(unsigned)i, 1u, 2u, 3u};
}
// consumer: workgroup b reads group (b & 1)'s region of `bytes_kv` bytes (all waves, 16 B per lane, KV_IT iterations), plus `bytes_w` of its own weights,
// XOR-reduces through LDS and stores 64 B.
__global__ __launch_bounds__(512) void consumer(const u32x4* __restrict__ kv, const u32x4* __restrict__ w, float* __restrict__ out, int kv16, int w16) {
  __shared__ unsigned red[512];
  const int grp = blockIdx.x & 1;
  const u32x4* base = kv + (size_t)grp * kv16;
  u32x4 acc = {0, 0, 0, 0};
  for (int i = threadIdx.x; i < w16; i += 512) { const u32x4 v = w[(size_t)blockIdx.x * w16 + i]; acc ^= v; }
#pragma unroll 4
  for (int i = threadIdx.x; i < kv16; i += 512) { const u32x4 v = base[i]; acc ^= v; }
  red[threadIdx.x] = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
  __syncthreads();
  if (threadIdx.x < 16) {
    unsigned r = 0;
    for (int j = 0; j < 32; ++j) r ^= red[threadIdx.x * 32 + j];
    out[blockIdx.x * 16 + threadIdx.x] = (float)r;
  }
}
int main() {
  hipStream_t s; CK(hipStreamCreate(&s));
  const int kv_bytes = 199 * 1024, NL = 28;
  u32x4* kv; CK(hipMalloc(&kv, (size_t)NL * 2 * kv_bytes)); CK(hipMemset(kv, 1, (size_t)NL * 2 * kv_bytes));
  u32x4* w; CK(hipMalloc(&w, (size_t)NL * 256 * 32 * 1024)); CK(hipMemset(w, 2, (size_t)NL * 256 * 32 * 1024));
  float* out; CK(hipMalloc(&out, 1 << 20));
  struct Cfg { int g, wkb; } cfgs[] = {{96, 24}, {144, 16}, {48, 48}, {24, 96}, {96, 0}};
  for (auto c : cfgs) {
    for (int with_kv = 1; with_kv >= 0; --with_kv) {
      hipGraph_t g; hipGraphExec_t ge;
      CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      for (int rep = 0; rep < 10; ++rep)
        for (int l = 0; l < NL; ++l) {          // every layer has its own K / V and weights (cold, as in the chunk)
          hipLaunchKernelGGL(producer, dim3(128), dim3(64), 0, s, kv + (size_t)l * 2 * kv_bytes / 16, 512);
          hipLaunchKernelGGL(consumer, dim3(c.g), dim3(512), 0, s, kv + (size_t)l * 2 * kv_bytes / 16, w + (size_t)l * 256 * 2048, out, with_kv ? kv_bytes / 16 : 64,
                             c.wkb * 64);
        }
      CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
      CK(hipEventRecord(e0, s));
      for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ge, s));
      CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("%3d workgroups x 512 thr, %2d KB weights each, K/V %s: %.2f us per (producer + consumer) pair\n", c.g, c.wkb, with_kv ? "199 KB per workgroup" : "not read           ",
             ms * 1e3 / (5 * 10 * NL));
      CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
  }
  return 0;
}
