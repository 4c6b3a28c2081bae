// Torch-free harness for hardware counters on the MFMA GEMM (rocprofv3 --pmc aborts inside torch on this image).
// Launches vlaser_gemm(EPI_NONE) through the C ABI on two shapes of the path, cycling over 8 weight buffers:
//   A: ViT qkv      M=1025 N=3072  K=1024   (one workgroup per CU regime)
//   B: SFT gate/up  M=560  N=17920 K=1536   (multi-wave grid)
// Usage: gemm_pmc [A|B] [rounds]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../include/vlaser_hip.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static unsigned short bf16(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16); }
int main(int argc, char** argv) {
  const char which = argc > 1 ? argv[1][0] : 'A';
  const int rounds = argc > 2 ? atoi(argv[2]) : 4;
  const int M = which == 'A' ? 1025 : 560, N = which == 'A' ? 3072 : 17920, K = which == 'A' ? 1024 : 1536, NL = 8;
  std::vector<unsigned short> hw((size_t)N * K), hx((size_t)M * K);
  unsigned s = 777u;
  for (auto& v : hw) { s = s * 1664525u + 1013904223u; v = bf16(((int)(s >> 16) % 2001 - 1000) * 3e-5f); }
  for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = bf16(((int)(s >> 16) % 2001 - 1000) * 1e-3f); }
  std::vector<void*> W(NL);
  for (int i = 0; i < NL; ++i) { CK(hipMalloc(&W[i], hw.size() * 2)); CK(hipMemcpy(W[i], hw.data(), hw.size() * 2, hipMemcpyHostToDevice)); }
  void *x, *out;
  CK(hipMalloc(&x, hx.size() * 2)); CK(hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
  CK(hipMalloc(&out, (size_t)M * N * 2));
  hipStream_t st; CK(hipStreamCreate(&st));
  VlaserGemmArgs a; memset(&a, 0, sizeof(a));
  a.A = x; a.out = out; a.M = M; a.N = N; a.K = K; a.lda = K; a.ldw = K; a.ldo = N; a.k_splits = 1;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int r = 0; r < rounds + 1; ++r) {
    if (r == 1) CK(hipEventRecord(e0, st));
    for (int i = 0; i < NL; ++i) {
      a.W = W[i];
      if (vlaser_gemm(VL_EPI_NONE, &a, (vl_stream_t)st) != 0) { fprintf(stderr, "%s\n", vlaser_last_error()); return 2; }
    }
  }
  CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  const double fl = 2.0 * M * N * K, us = ms * 1e3 / (rounds * NL);
  printf("shape %c M=%d N=%d K=%d: %.2f us/launch (eager, host gaps included) -> %.0f TFLOP/s\n", which, M, N, K, us, fl / us / 1e6);
  return 0;
}
