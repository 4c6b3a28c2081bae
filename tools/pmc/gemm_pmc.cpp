// Torch-free harness for hardware counters on the MFMA GEMMs (rocprofv3 --pmc aborts inside torch on this image).
// Launches the GEMM entry points through the C ABI on shapes of the path, cycling over 8 weight buffers:
//   A: ViT qkv            M=1025 N=3072  K=1024   NT, 144x128 tile (6 waves, 4 stages; r03)
//   B: SFT gate/up fwd    M=560  N=17920 K=1536   NT, 192x256 tile (r03)
//   C: LLM prefill o_proj M=384  N=1536  K=1536   NT, 64x64 tile (8 stages; r03)
//   D: SFT gate/up wgrad  out 17920x1536, contraction 576   TN form (vlaser_gemm_tn_lds), 256x256 tile, with the r04 sum-of-squares slots
//   E: SFT down dgrad     M=560  N=8960  K=1536   NN form (vlaser_gemm_nn, weights as stored), 128x256 tile
// Usage: gemm_pmc [A|B|C|D|E] [rounds]
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
  int M, N, K;
  switch (which) {
    case 'A': M = 1025; N = 3072; K = 1024; break;
    case 'B': M = 560; N = 17920; K = 1536; break;
    case 'C': M = 384; N = 1536; K = 1536; break;
    case 'D': M = 17920; N = 1536; K = 576; break;          // out [M, N] = At[K, M]^T @ Wt[K, N]
    case 'E': M = 560; N = 8960; K = 1536; break;           // out [M, N] = A[M, K] @ W[K, N]  (W = down_proj.weight [1536, 8960] as stored)
    default: fprintf(stderr, "shape A..E\n"); return 1;
  }
  const int NL = 8;
  // W: NT form [N, K]; TN form At [K, M] (the cycled operand); NN form W [K, N]
  const size_t w_elems = which == 'D' ? (size_t)K * M : (size_t)N * K, x_elems = which == 'D' ? (size_t)K * N : (size_t)M * K;
  std::vector<unsigned short> hw(w_elems), hx(x_elems);
  unsigned s = 777u;
  for (auto& v : hw) { s = s * 1664525u + 1013904223u; v = bf16(((int)(s >> 16) % 2001 - 1000) * 3e-5f); }
  for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = bf16(((int)(s >> 16) % 2001 - 1000) * 1e-3f); }
  std::vector<void*> W(NL);
  for (int i = 0; i < NL; ++i) { CK(hipMalloc(&W[i], hw.size() * 2)); CK(hipMemcpy(W[i], hw.data(), hw.size() * 2, hipMemcpyHostToDevice)); }
  void *x, *out;
  float* slots = nullptr;
  const int n_slots = 1 << 16;
  CK(hipMalloc(&x, hx.size() * 2)); CK(hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
  CK(hipMalloc(&out, (size_t)M * N * 2));
  CK(hipMalloc(&slots, n_slots * 4)); CK(hipMemset(slots, 0, n_slots * 4));
  hipStream_t st; CK(hipStreamCreate(&st));
  VlaserGemmArgs a; memset(&a, 0, sizeof(a));
  a.A = x; a.out = out; a.M = M; a.N = N; a.K = K; a.lda = K; a.ldw = which == 'E' ? N : K; a.ldo = N; a.k_splits = 1;
  auto launch = [&](void* w) -> int {
    if (which == 'D') return vlaser_gemm_tn_lds(w, x, out, M, N, K, M, N, N, 0, slots, n_slots, (vl_stream_t)st);
    a.W = w;
    return which == 'E' ? vlaser_gemm_nn(VL_EPI_NONE, &a, (vl_stream_t)st) : vlaser_gemm(VL_EPI_NONE, &a, (vl_stream_t)st);
  };
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int r = 0; r < rounds + 1; ++r) {
    if (r == 1) CK(hipEventRecord(e0, st));
    for (int i = 0; i < NL; ++i)
      if (launch(W[i]) != 0) { fprintf(stderr, "%s\n", vlaser_last_error()); return 2; }
  }
  CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  const double fl = 2.0 * M * N * K, us = ms * 1e3 / (rounds * NL);
  printf("shape %c M=%d N=%d K=%d: %.2f us/launch (eager, host gaps included) -> %.0f TFLOP/s\n", which, M, N, K, us, fl / us / 1e6);
  return 0;
}
