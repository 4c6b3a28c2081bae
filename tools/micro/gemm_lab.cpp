// Torch-free GEMM lab (MI355X): times vlaser_gemm (C ABI) on the path's shapes for every tile configuration (force_bm), inside
// a HIP graph cycling over 8 weight buffers, and checks each configuration against the 64-row register-staged kernel.
// Build: hipcc --offload-arch=gfx950 -O3 gemm_lab.cpp -I../../include -L../../vlaser_amd/csrc -lvlaser_hip -Wl,-rpath,'$ORIGIN/../../vlaser_amd/csrc' -o gemm_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "vlaser_hip.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void fill_bf16(unsigned short* p, size_t n, unsigned seed, float scale) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    float v = ((h & 0xffff) / 32768.0f - 1.0f) * scale;
    p[i] = (unsigned short)(__float_as_uint(v) >> 16);
  }
}
__global__ void max_diff(const unsigned short* a, const unsigned short* b, size_t n, float* out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  float m = 0.f;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float x = __uint_as_float((unsigned)a[i] << 16), y = __uint_as_float((unsigned)b[i] << 16);
    m = fmaxf(m, fabsf(x - y));
  }
  atomicMax((int*)out, __float_as_int(m));
}

struct Shape { int M, N, K; const char* name; int epi; };

int main(int argc, char** argv) {
  hipStream_t s; CK(hipStreamCreate(&s));
  std::vector<Shape> shapes = {
      {384, 2048, 1536, "llm qkv", VL_EPI_NONE}, {384, 17920, 1536, "llm gate/up", VL_EPI_SWIGLU}, {1025, 3072, 1024, "vit qkv", VL_EPI_NONE},
      {1025, 4096, 1024, "vit fc1", VL_EPI_BIAS_GELU}, {560, 17920, 1536, "sft gate/up", VL_EPI_NONE}, {560, 2048, 1536, "sft qkv", VL_EPI_NONE},
      {560, 8960, 1536, "sft dgrad wdown", VL_EPI_NONE}, {560, 1536, 2048, "sft dgrad qkv", VL_EPI_NONE}, {13 * 1025, 4096, 1024, "vit fc1 x13", VL_EPI_BIAS_GELU},
      {3408, 37888, 3584, "8B gate/up", VL_EPI_SWIGLU},
      // fixed cost vs per-K-step cost vs epilogue cost of a single-round grid (run with the filter "ksw")
      {1025, 4096, 256, "ksw fc1 none", VL_EPI_NONE}, {1025, 4096, 512, "ksw fc1 none", VL_EPI_NONE}, {1025, 4096, 1024, "ksw fc1 none", VL_EPI_NONE},
      {1025, 4096, 2048, "ksw fc1 none", VL_EPI_NONE}, {1025, 4096, 256, "ksw fc1 gelu", VL_EPI_BIAS_GELU}, {1025, 4096, 1024, "ksw fc1 gelu", VL_EPI_BIAS_GELU},
      {1025, 4096, 1024, "ksw fc1 f32", VL_EPI_F32}, {384, 17920, 1536, "ksw gu none", VL_EPI_NONE}, {384, 17920, 1536, "ksw gu swiglu", VL_EPI_SWIGLU},
      {384, 17920, 512, "ksw gu none", VL_EPI_NONE}, {384, 17920, 512, "ksw gu swiglu", VL_EPI_SWIGLU}};
  std::vector<Shape> pshapes = {{1025, 1024, 1024, "vit proj", 4}, {1025, 1024, 4096, "vit fc2", 4}, {384, 1536, 1536, "llm o", 3}, {384, 1536, 8960, "llm down", 7},
                                {560, 1536, 8960, "sft down", 5}, {560, 1536, 17920, "sft dgrad gu", 10}};
  const int cfgs[] = {64, 128, 1100, 1200, 1300, 1440, 1500, 0};
  const int NL = 8;
  float* dmax; CK(hipMalloc(&dmax, 4));
  auto run = [&](const Shape& sh, int splits) {
    const int M = sh.M, N = sh.N, K = sh.K;
    unsigned short *x, *w[NL], *out, *ref, *bias; float* part = nullptr;
    CK(hipMalloc(&x, (size_t)M * K * 2)); CK(hipMalloc(&out, (size_t)M * N * 4 + 64)); CK(hipMalloc(&ref, (size_t)M * N * 4 + 64)); CK(hipMalloc(&bias, (size_t)N * 2));
    if (splits) CK(hipMalloc(&part, (size_t)splits * M * N * 4 * 2));
    fill_bf16<<<1024, 256, 0, s>>>(x, (size_t)M * K, 1, 1.0f);
    fill_bf16<<<64, 256, 0, s>>>(bias, (size_t)N, 77, 0.5f);
    for (int i = 0; i < NL; ++i) { CK(hipMalloc(&w[i], (size_t)N * K * 2)); fill_bf16<<<1024, 256, 0, s>>>(w[i], (size_t)N * K, 100 + i, 0.03f); }
    const double fl = 2.0 * M * N * K;
    for (int cfg : cfgs) {
      VlaserGemmArgs a; memset(&a, 0, sizeof a);
      a.A = x; a.M = M; a.N = N; a.K = K; a.lda = K; a.ldw = K; a.ldo = sh.epi == VL_EPI_SWIGLU ? N / 2 : N; a.bias = bias; a.force_bm = cfg;
      int epi = sh.epi;
      if (splits) { epi = VL_EPI_PARTIAL; a.out_f32 = part; a.k_splits = splits; }
      auto launch = [&](int i, unsigned short* o) { a.W = w[i % NL]; a.out = o; return vlaser_gemm(epi, &a, s); };
      if (launch(0, cfg == 64 ? ref : out)) { printf("%-16s cfg %4d: %s\n", sh.name, cfg, vlaser_last_error()); continue; }
      CK(hipStreamSynchronize(s));
      float md = 0.f;
      if (cfg != 64) {
        CK(hipMemsetAsync(dmax, 0, 4, s));
        if (splits) max_diff<<<512, 256, 0, s>>>((unsigned short*)part, (unsigned short*)part, 1, dmax);   // partial slabs: timed only
        else max_diff<<<512, 256, 0, s>>>(out, ref, (size_t)M * a.ldo, dmax);
        CK(hipMemcpyAsync(&md, dmax, 4, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
      }
      hipGraph_t g; hipGraphExec_t ge;
      CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      for (int i = 0; i < NL; ++i) launch(i, out);
      CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
      const int reps = 10;
      CK(hipEventRecord(e0, s));
      for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s));
      CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double us = ms * 1e3 / (reps * NL);
      printf("%-16s M=%5d N=%5d K=%5d S=%2d cfg %4d: %8.2f us %7.1f TF  maxdiff %.4g\n", sh.name, M, N, K, splits, cfg, us, fl / us / 1e6, md);
      CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    CK(hipFree(x)); CK(hipFree(out)); CK(hipFree(ref)); CK(hipFree(bias)); if (part) CK(hipFree(part));
    for (int i = 0; i < NL; ++i) CK(hipFree(w[i]));
  };
  const char* only = argc > 1 ? argv[1] : nullptr;
  for (auto& sh : shapes) if (only ? strstr(sh.name, only) != nullptr : strncmp(sh.name, "ksw", 3) != 0) run(sh, 0);
  for (auto& sh : pshapes) if (!only || strstr(sh.name, only)) run(sh, sh.epi);
  return 0;
}
