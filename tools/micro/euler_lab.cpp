// Euler-phase lab (MI355X, torch-free): the action expert's gate/up and down weight-streaming GEMVs through the C ABI, timed in a
// HIP graph over 28 distinct weight buffers (HBM-cold), alone and right behind a "touch" kernel that pulls the same bytes through
// the L2 of the XCD whose workgroup will read them -- does a warm L2 / Infinity Cache shorten the stream phase?
// Build: hipcc --offload-arch=gfx950 -O3 euler_lab.cpp -I../../include -L../../vlaser_amd/csrc -lvlaser_hip -Wl,-rpath,'$ORIGIN/../../vlaser_amd/csrc' -o euler_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <functional>
#include <vector>
#include "vlaser_hip.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(4))) unsigned int u4;

__global__ void fill_bf16(unsigned short* p, size_t n, unsigned seed, float scale) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    float v = ((h & 0xffff) / 32768.0f - 1.0f) * scale;
    p[i] = (unsigned short)(__float_as_uint(v) >> 16);
  }
}
// workgroup b touches bytes [b*chunk, (b+1)*chunk) (the slice the same-numbered workgroup of the consumer will stream)
__global__ __launch_bounds__(512) void touch(const u4* __restrict__ w, size_t chunk16, size_t total16, unsigned* sink) {
  const size_t lo = (size_t)blockIdx.x * chunk16, hi = lo + chunk16 < total16 ? lo + chunk16 : total16;
  unsigned acc = 0;
  for (size_t i = lo + threadIdx.x; i < hi; i += 512) acc ^= w[i].x;
  if (acc == 0x12345u) sink[0] = acc;
}

static float time_graph(hipStream_t s, int reps, const std::function<void()>& body, int n_items) {
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  body();
  CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
  CK(hipEventRecord(e0, s));
  for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s));
  CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  return ms * 1e3f / (reps * n_items);
}

int main() {
  hipStream_t s; CK(hipStreamCreate(&s));
  const int NL = 28, M = 4, H = 768, I = 8960;
  unsigned short *h, *nw, *hout, *act; float *parts, *pd; unsigned* sink;
  CK(hipMalloc(&h, M * H * 2)); CK(hipMalloc(&nw, H * 2)); CK(hipMalloc(&hout, 16 * H * 2)); CK(hipMalloc(&act, 16 * I * 2));
  CK(hipMalloc(&parts, 8 * 16 * H * 4)); CK(hipMalloc(&pd, 8 * 16 * H * 4)); CK(hipMalloc(&sink, 4));
  fill_bf16<<<64, 256, 0, s>>>(h, M * H, 1, 1.0f); fill_bf16<<<64, 256, 0, s>>>(nw, H, 2, 1.0f); fill_bf16<<<64, 256, 0, s>>>(act, 16 * I, 3, 1.0f);
  CK(hipMemsetAsync(parts, 0, 8 * 16 * H * 4, s));
  std::vector<unsigned short*> wgu(NL), wd(NL);
  const size_t gu_elems = (size_t)2 * I * H, d_elems = (size_t)H * I;
  for (int l = 0; l < NL; ++l) {
    CK(hipMalloc(&wgu[l], gu_elems * 2)); fill_bf16<<<1024, 256, 0, s>>>(wgu[l], gu_elems, 10 + l, 0.03f);
    CK(hipMalloc(&wd[l], d_elems * 2)); fill_bf16<<<1024, 256, 0, s>>>(wd[l], d_elems, 50 + l, 0.03f);
  }
  // evictor: a buffer larger than L2 + Infinity Cache touched between measurements is not needed: 28 x 41 MB of weights cycle through
  auto gu = [&](int l) {
    VlaserSkinnyArgs a; memset(&a, 0, sizeof a);
    a.x = h; a.partials = parts; a.n_partials = 3; a.norm_w = nw; a.eps = 1e-6f; a.h_out = hout; a.W = wgu[l]; a.M = M; a.N = 2 * I; a.K = H; a.ldw = H;
    a.n_valid = 2 * I; a.tiles_per_unit = 2; a.k_splits = 1; a.out = act; a.ldo = I;
    if (vlaser_skinny(VL_PRO_NORM, VL_SK_SWIGLU, &a, s)) { printf("gu: %s\n", vlaser_last_error()); exit(1); }
  };
  auto down = [&](int l) {
    VlaserSkinnyArgs a; memset(&a, 0, sizeof a);
    a.x = act; a.W = wd[l]; a.M = M; a.N = H; a.K = I; a.ldw = I; a.n_valid = H; a.tiles_per_unit = 1; a.k_splits = 5; a.out_f32 = pd;
    if (vlaser_skinny(VL_PRO_PLAIN, VL_SK_PARTIAL, &a, s)) { printf("down: %s\n", vlaser_last_error()); exit(1); }
  };
  auto touch_w = [&](const void* w, size_t bytes, int blocks) {
    const size_t total16 = bytes / 16, chunk16 = (total16 + blocks - 1) / blocks;
    touch<<<blocks, 512, 0, s>>>((const u4*)w, chunk16, total16, sink);
  };
  gu(0); down(0); CK(hipStreamSynchronize(s));
  const float t_gu = time_graph(s, 10, [&] { for (int l = 0; l < NL; ++l) gu(l); }, NL);
  const float t_dn = time_graph(s, 10, [&] { for (int l = 0; l < NL; ++l) down(l); }, NL);
  printf("gate/up cold: %.2f us   down cold: %.2f us\n", t_gu, t_dn);
  for (int blocks : {256, 512, 1024}) {
    const float t_t = time_graph(s, 10, [&] { for (int l = 0; l < NL; ++l) touch_w(wgu[l], gu_elems * 2, blocks); }, NL);
    const float t_tg = time_graph(s, 10, [&] { for (int l = 0; l < NL; ++l) { touch_w(wgu[l], gu_elems * 2, blocks); gu(l); } }, NL);
    const float t_td = time_graph(s, 10, [&] { for (int l = 0; l < NL; ++l) touch_w(wd[l], d_elems * 2, blocks); }, NL);
    const float t_tdd = time_graph(s, 10, [&] { for (int l = 0; l < NL; ++l) { touch_w(wd[l], d_elems * 2, blocks); down(l); } }, NL);
    printf("touch blocks %4d: touch(gu) %.2f us (%.0f GB/s), touch+gu %.2f -> warm gu %.2f us | touch(down) %.2f, touch+down %.2f -> warm down %.2f us\n", blocks, t_t,
           gu_elems * 2 / t_t / 1e3, t_tg, t_tg - t_t, t_td, t_tdd, t_tdd - t_td);
  }
  // warm both, then run both (what an idle-CU warm-up during the attention launch would give)
  const float t_both_cold = time_graph(s, 10, [&] { for (int l = 0; l < NL; ++l) { gu(l); down(l); } }, NL);
  const float t_tt = time_graph(s, 10, [&] { for (int l = 0; l < NL; ++l) { touch_w(wgu[l], gu_elems * 2, 256); touch_w(wd[l], d_elems * 2, 256); } }, NL);
  const float t_both_warm = time_graph(s, 10, [&] { for (int l = 0; l < NL; ++l) { touch_w(wgu[l], gu_elems * 2, 256); touch_w(wd[l], d_elems * 2, 256); gu(l); down(l); } }, NL);
  printf("gu+down cold %.2f us; touches %.2f us; touches+gu+down %.2f -> warm gu+down %.2f us\n", t_both_cold, t_tt, t_both_warm, t_both_warm - t_tt);
  return 0;
}
