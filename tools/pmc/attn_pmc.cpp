// Torch-free harness for hardware counters on the prefill attention kernel (rocprofv3 --pmc aborts inside torch on this image).
//   A: ViT, 1 tile      B=1  S=1025 16/16 heads hd 64  FULL    (one 4-wave workgroup per CU)
//   B: ViT, 13 tiles    B=13 S=1025 16/16 heads hd 64  FULL    (8 waves per SIMD)
//   C: joint prefill    B=1  S=384  12/2  heads hd 128 CAUSAL
// Usage: attn_pmc [A|B|C] [rounds]
#include <hip/hip_runtime.h>
#include <cmath>
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
  const int B = which == 'B' ? 13 : 1, S = which == 'C' ? 384 : 1025, nq = which == 'C' ? 12 : 16, nkv = which == 'C' ? 2 : 16, hd = which == 'C' ? 128 : 64;
  const int Sp = (S + 63) / 64 * 64;
  const size_t nQ = (size_t)B * S * nq * hd, nK = (size_t)B * nkv * Sp * hd;
  std::vector<unsigned short> hq(nQ), hk(nK), hv(nK);
  unsigned s = 777u;
  auto fill = [&](std::vector<unsigned short>& v, float sc) { for (auto& x : v) { s = s * 1664525u + 1013904223u; x = bf16(((int)(s >> 16) % 2001 - 1000) * sc); } };
  fill(hq, 1e-3f); fill(hk, 1e-3f); fill(hv, 1e-3f);
  void *q, *k, *vt, *out;
  CK(hipMalloc(&q, nQ * 2)); CK(hipMalloc(&k, nK * 2)); CK(hipMalloc(&vt, nK * 2)); CK(hipMalloc(&out, nQ * 2));
  CK(hipMemcpy(q, hq.data(), nQ * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(k, hk.data(), nK * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(vt, hv.data(), nK * 2, hipMemcpyHostToDevice));
  hipStream_t st; CK(hipStreamCreate(&st));
  VlaserAttnArgs a; memset(&a, 0, sizeof(a));
  a.q = q; a.k = k; a.vt = vt; a.out = out; a.batch = B; a.sq = S; a.kv_len = S; a.n_q_heads = nq; a.n_kv_heads = nkv; a.head_dim = hd;
  a.q_bs = (long long)S * nq * hd; a.q_hs = hd; a.q_ss = nq * hd;
  a.k_bs = (long long)nkv * Sp * hd; a.k_hs = (long long)Sp * hd;
  a.vt_bs = (long long)nkv * hd * Sp; a.vt_hs = (long long)hd * Sp;
  a.o_bs = (long long)S * nq * hd; a.o_ss = nq * hd;
  a.ld_vt = Sp; a.scale = 1.0f / sqrtf((float)hd); a.mode = which == 'C' ? VL_ATTN_CAUSAL : VL_ATTN_FULL; a.n_splits = 1;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int NL = 8;
  for (int r = 0; r < rounds + 1; ++r) {
    if (r == 1) CK(hipEventRecord(e0, st));
    for (int i = 0; i < NL; ++i)
      if (vlaser_attn_prefill(&a, (vl_stream_t)st) != 0) { fprintf(stderr, "%s\n", vlaser_last_error()); return 2; }
  }
  CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  const double fl = 4.0 * B * nq * (double)S * S * hd * (which == 'C' ? 0.5 : 1.0), us = ms * 1e3 / (rounds * NL);
  printf("shape %c B=%d S=%d heads %d/%d hd=%d: %.2f us/launch (eager, host gaps included) -> %.0f TFLOP/s\n", which, B, S, nq, nkv, hd, us, fl / us / 1e6);
  return 0;
}
