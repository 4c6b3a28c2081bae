// Torch-free harness for hardware counters on the dominant kernel (rocprofv3 --pmc aborts inside torch on this image).
// Launches the action expert's gate/up GEMV, N = 17920, K = 768, M = 4 -- since r05 vlaser_chain_gu (csrc/chain.hip: chain_gu_kernel, every unit of a workgroup
// requested up front), with `skinny` as second argument the r01-r04 vlaser_skinny(NORM, SWIGLU) -- through the
// C ABI, cycling over 28 distinct 27.5 MB weight buffers (770 MB > the 256 MiB Infinity Cache, so every launch
// streams from HBM exactly as in the real layer sequence).  Usage: skinny_pmc [rounds] [chain16|chain|skinny]  (chain16 = 16-row lane-local units, the default path)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../include/vlaser_hip.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

static unsigned short bf16(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16); }

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 4;
  const char* which = argc > 2 ? argv[2] : "chain16";      // the default path of PiZeroInference (EULER_DEFAULT: gu16 + chain)
  const bool old_kernel = !strcmp(which, "skinny"), units16 = !strcmp(which, "chain16");
  // [nbuf] distinct weight buffers cycled (default 28 = HBM-cold; 4 = 110 MB: warm in the 256 MiB Infinity Cache; 1 = 27.6 MB: one eighth per XCD, warm in its 4 MB L2) and
  // [graph]: the NL launches captured into one hipGraph (dependent chain without host gaps) -- what the weight stream costs by where it comes from
  const int nbuf = argc > 3 ? atoi(argv[3]) : 28;
  const bool graph = argc > 4 && !strcmp(argv[4], "graph");
  const int M = 4, K = 768, N = 17920, NL = 28, NP = 3;
  const size_t wbytes = (size_t)N * K * 2;
  std::vector<unsigned short> hw((size_t)N * K);
  unsigned s = 12345u;
  for (auto& v : hw) { s = s * 1664525u + 1013904223u; v = bf16(((int)(s >> 16) % 2001 - 1000) * 3e-5f); }
  std::vector<void*> W(NL);
  for (int i = 0; i < NL; ++i) {
    if (i >= nbuf) { W[i] = W[i % nbuf]; continue; }
    CK(hipMalloc(&W[i], wbytes)); CK(hipMemcpy(W[i], hw.data(), wbytes, hipMemcpyHostToDevice));
  }
  std::vector<unsigned short> hx((size_t)M * K), hn(K, bf16(1.0f));
  for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = bf16(((int)(s >> 16) % 2001 - 1000) * 1e-3f); }
  std::vector<float> hp((size_t)NP * M * K, 0.01f);
  void *x, *nw, *hout, *out; float* parts;
  CK(hipMalloc(&x, hx.size() * 2)); CK(hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
  CK(hipMalloc(&nw, K * 2)); CK(hipMemcpy(nw, hn.data(), K * 2, hipMemcpyHostToDevice));
  CK(hipMalloc((void**)&parts, hp.size() * 4)); CK(hipMemcpy(parts, hp.data(), hp.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&hout, (size_t)M * K * 2)); CK(hipMalloc(&out, (size_t)M * (N / 2) * 2));
  hipStream_t st; CK(hipStreamCreate(&st));
  VlaserSkinnyArgs a; memset(&a, 0, sizeof(a));
  a.x = x; a.partials = parts; a.n_partials = NP; a.norm_w = nw; a.eps = 1e-6f; a.h_out = hout;
  a.M = M; a.N = N; a.K = K; a.n_valid = N; a.tiles_per_unit = units16 ? 1 : 2; a.k_splits = 1; a.out = out; a.ldo = N / 2;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipGraphExec_t gexec = nullptr;
  for (int r = 0; r < rounds + 1; ++r) {
    if (r == 1) CK(hipEventRecord(e0, st));
    if (graph && gexec) { CK(hipGraphLaunch(gexec, st)); continue; }
    if (graph && r == 1) CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < NL; ++i) {
      a.W = W[i];
      const int rc = old_kernel ? vlaser_skinny(VL_PRO_NORM, VL_SK_SWIGLU, &a, (vl_stream_t)st) : vlaser_chain_gu(&a, (vl_stream_t)st);
      if (rc != 0) { fprintf(stderr, "%s\n", vlaser_last_error()); return 2; }
    }
    if (graph && r == 1) {
      hipGraph_t g; CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&gexec, g, nullptr, nullptr, 0));
      CK(hipStreamSynchronize(st)); CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(gexec, st));
    }
  }
  CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  const double alg = (double)wbytes + M * K * 2.0 + NP * M * K * 4.0 + M * (N / 2) * 2.0 + M * K * 2.0;
  printf("%s, %d weight buffers, %s: launches %d  us/launch %.2f  algorithmic bytes/launch %.0f  -> %.0f GB/s%s\n", which, nbuf, graph ? "one hipGraph" : "eager", rounds * NL,
         ms * 1e3 / (rounds * NL), alg, alg / (ms * 1e-3 / (rounds * NL)) / 1e9, graph ? "" : " (eager launches: includes host gaps)");
  return 0;
}
