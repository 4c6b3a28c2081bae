// Microbenchmark: does the MFMA-fragment access pattern (16 rows x 64 B per wave load) stream slower than a
// contiguous 1 KB per wave load?  27.5 MB per launch, 256 blocks x 512 threads, 12 x 16 B loads in flight per lane.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// MODE 0: contiguous: wave-load i reads 1 KB at base + i*1KB (lane*16)
// MODE 1: fragment: lane (r = l&15, g = l>>4) reads 16 B of row r at col g*8 (+32 per step); rows are K*2 bytes apart
template <int MODE>
__global__ __launch_bounds__(512) void k(const u4* __restrict__ w, float* out, int K, int units_per_block, int n_units) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nsteps = K / 8 / 32;  // per wave
  unsigned acc = 0;
  for (int ui = 0; ui < units_per_block; ++ui) {
    const int unit = blockIdx.x * units_per_block + ui;
    if (unit >= n_units) break;
    u4 v[16];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      if (s < nsteps) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          size_t off16;  // in 16-byte units
          if (MODE == 0) {
            off16 = ((size_t)unit * 32 * K * 2 + ((size_t)(wave * nsteps + s) * 2 + t) * 1024) / 16 + lane;
          } else {
            const int row = unit * 32 + t * 16 + (lane & 15);
            off16 = ((size_t)row * K * 2 + (size_t)(wave * nsteps + s) * 64 + (lane >> 4) * 16) / 16;
          }
          v[s * 2 + t] = __builtin_nontemporal_load(w + off16);
        }
      }
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) if (s < 2 * nsteps) acc += v[s].x ^ v[s].w;
  }
  if (acc == 0x12345678) out[0] = 1.f;
}

template <class F>
float time_graph(hipStream_t s, int n, int reps, F launch) {
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
  for (int i = 0; i < n; ++i) launch(i);
  hipStreamEndCapture(s, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipGraphLaunch(ge, s); hipStreamSynchronize(s);
  hipEventRecord(e0, s);
  for (int r = 0; r < reps; ++r) hipGraphLaunch(ge, s);
  hipEventRecord(e1, s); hipStreamSynchronize(s);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / (reps * n);
}

int main() {
  hipStream_t s; CK(hipStreamCreate(&s));
  float* b; CK(hipMalloc(&b, 1024));
  const size_t big = (size_t)1 << 31;
  u4* w; CK(hipMalloc(&w, big)); CK(hipMemset(w, 1, big));
  struct { int N, K; const char* name; } cases[] = {{17920, 768, "gate/up 17920x768"}, {2048, 768, "qkv 2048x768"}, {768, 8960 / 7, "down slice 768x1280 (x7 splits)"}};
  for (auto c : cases) {
    const int n_units = c.N / 32;
    const size_t bytes = (size_t)c.N * c.K * 2;
    const int nk = (int)(big / bytes) > 28 ? 28 : (int)(big / bytes);
    for (int blocks : {64, 128, 256, 512}) {
      if (blocks > n_units) continue;
      const int upb = (n_units + blocks - 1) / blocks;
      float t0 = time_graph(s, nk, 10, [&](int i) { hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(512), 0, s, w + (size_t)i * bytes / 16, b, c.K, upb, n_units); });
      float t1 = time_graph(s, nk, 10, [&](int i) { hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(512), 0, s, w + (size_t)i * bytes / 16, b, c.K, upb, n_units); });
      printf("%-34s blocks %3d: contiguous %.2f us (%.0f GB/s)   fragment-pattern %.2f us (%.0f GB/s)\n", c.name, blocks, t0, bytes / t0 / 1e3, t1, bytes / t1 / 1e3);
    }
  }
  return 0;
}
