// The <= 16-row layer-step chain (action-expert Euler steps, greedy decode), round 5: three kernels rebuilt around what the in-chain, in-kernel
// timeline showed (tools/micro/chain_timeline.py, profiles/r05a_chain_timeline.md): a dependent kernel boundary costs 1.4-1.6 us on this chip and a
// dependent load -> store kernel 1.75 us (tools/micro/launch_floor.hip) -- the r01-r04 "4.9 us floor" of a short kernel was its own serial structure:
//   * runtime-optional code paths (slab-count loops, second chunks, `if (ucount > u)` weight requests) make hipcc's waitcnt pass merge the worst case of
//     every path: `s_waitcnt vmcnt(0)` in front of units 1 and 2 of the gate/up stream (three dependent HBM round trips instead of one), in front of
//     the K / V^T requests of the attention, before the norm of every prologue;
//   * every workgroup of a consumer re-reduced the producer's fp32 split-K slabs (66-98 KB through one CU's L1 = 0.7 us before the first FLOP).
// Here every shape parameter is a template argument, every load of a kernel is issued in ONE straight-line block (small dependent data first, the
// weight stream behind it: vmcnt retires in issue order), and the down projection publishes the residual stream ONCE, as bf16:
//
//   chain_qkv  : one WAVE per 16-row unit (no LDS exchange between waves, no barrier): RMSNorm of the published bf16 residual stream, q/k/v GEMV over the
//                whole K, bias + RoPE + cache scatter (lane-local epilogue of skinny.hip's 16-row units; same packed weights)
//   chain_gu   : residual + o_proj's split-K slabs + RMSNorm -> gate/up GEMV -> SwiGLU with ALL units of the workgroup requested up front; same
//                arithmetic and summation order as skinny_kernel<NORM, SWIGLU> (bit-identical outputs)
//   chain_down : down projection WITHOUT cross-workgroup split-K: a workgroup owns 4 output columns over the whole K (v_mfma_f32_4x4x4_16b_bf16: 16
//                independent 4 x 4 x 4 blocks per instruction = 4 columns x 4 rows x 64 k), adds the residual and stores the next layer's bf16
//                residual stream -- the consumer loads 6 KB instead of 66 KB and runs no slab reduction
#include "common.h"
#include "../../include/vlaser_hip.h"

typedef __attribute__((ext_vector_type(4))) short s16x4;

#define CH_STAMP(i) do { if constexpr (DBG) { if (threadIdx.x == 0) dbg[i] = wall_clock64(); } } while (0)

__device__ __forceinline__ int ch_fdiv(int x, float inv_d) { return (int)(((float)x + 0.5f) * inv_d); }

// ------------------------------------------------------------------------------------------------------------------ gate / up
struct ChainGuP {
  const bf16_t* h_in; const float* partials; const bf16_t* norm_w; bf16_t* h_out;
  const u32x4* W; bf16_t* out;
  int M, ldo, ulo, urem, n_valid;
  float eps, inv_cpr;
  unsigned long long* dbg;
};

// NS = K / 256 (K-steps of 32 per wave), UE = units per workgroup, SP = split-K slabs of the producer, CPT = 16-byte chunks of the [M, K] residual stream per thread
// (ceil(M * K / 8 / 512)), TPU = MFMA tiles per unit: 2 = 32 packed rows [gate16 | up16] (ops.pack_gate_up), 1 = 16 rows whose lane group g holds
// [gate 2g, gate 2g+1, up 2g, up 2g+1] of the unit's 8 activation columns (ops.pack_gate_up8: 1120 instead of 560 units -> 224 workgroups x 5 instead of 187 x 3)
template <int NS, int UE, int SP, int CPT, int TPU, bool DBG>
__global__ __launch_bounds__(512) void chain_gu_kernel(ChainGuP p) {
  constexpr int K = NS * 256, cpr = K / 8, ngr = cpr / 16, gstride = (ngr + 3) & ~3, XS = K * 2 + 16, NF = TPU * NS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, g = lane >> 4;
  unsigned long long* dbg = DBG ? p.dbg + (size_t)blockIdx.x * 8 : nullptr;
  CH_STAMP(0);
  const int M = p.M;
  const int ucount = p.ulo + ((int)blockIdx.x < p.urem ? 1 : 0);
  const int ustart = (int)blockIdx.x * p.ulo + min((int)blockIdx.x, p.urem);
  const int xs_bytes = (M * XS + 15) & ~15;
  char* xs = smem;
  float* red = reinterpret_cast<float*>(smem + xs_bytes);      // [UE][8][TPU][64] f32x4 partial tiles; group sums [M][gstride] during the prologue
  float* gs = red;
  const int nch = M * cpr, slab = M * K;

  // ---- every request of the kernel, in issue order: residual chunk(s), norm weight, the producer's slabs (L2), then the whole weight stream (HBM)
  u32x4 hv[CPT], wv[CPT];
  f32x4 q[CPT][SP > 0 ? 2 * SP : 1];
  int mm[CPT], cc[CPT];
#pragma unroll
  for (int c = 0; c < CPT; ++c) {
    const int ch = min(c * 512 + tid, nch - 1);
    mm[c] = ch_fdiv(ch, p.inv_cpr);
    cc[c] = (ch - mm[c] * cpr) << 3;
    const int off = mm[c] * K + cc[c];
    hv[c] = ld_global_16(p.h_in + off);
    wv[c] = ld_global_16(p.norm_w + cc[c]);
#pragma unroll
    for (int u = 0; u < SP; ++u) {
      const float* pp = p.partials + (off + u * slab);
      q[c][2 * u] = *reinterpret_cast<const f32x4*>(pp);
      q[c][2 * u + 1] = *reinterpret_cast<const f32x4*>(pp + 4);
    }
  }
  // (hipcc's scheduler moves loads freely -- it does not know that vmcnt retires in issue order: without the fences below it put the norm-weight chunk and two
  //  fragments of unit 0 BEHIND units 1 and 2, so the norm and the first MFMAs waited for the whole stream)
  __builtin_amdgcn_sched_barrier(0);
  // All 8 waves put their (L2-resident) prologue requests into the CU's memory pipeline BEFORE any wave queues weight requests: the pipeline is FIFO per CU, and a
  // wave that started late found its 8 small requests behind 7 x 18 KB of the others' HBM requests (in-chain timeline r05b: prologue data back 3.7 us after the start
  // instead of 2.3).  The barrier waits for instruction issue only -- nobody waits for data here.
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  u32x4 w[UE][NF];
  {
    const u32x4* wp = p.W + (size_t)wave * (NF * 64) + lane;
#pragma unroll
    for (int u = 0; u < UE; ++u) {
      const u32x4* src = wp + (size_t)(ustart + min(u, ucount - 1)) * (8 * NF * 64);      // a workgroup with fewer units re-requests its last one
#pragma unroll
      for (int f = 0; f < NF; ++f) w[u][f] = __builtin_nontemporal_load(src + f * 64);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- phase 1: h = bf16(h_in + sum slabs) (slabs first, residual last), sum of squares per 16-lane group -> LDS
  u32x4 hr[CPT];
#pragma unroll
  for (int c = 0; c < CPT; ++c) {
    float v[8], sl[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[2 * j] = bf16lo_to_f32(hv[c][j]); v[2 * j + 1] = bf16hi_to_f32(hv[c][j]); }
#pragma unroll
    for (int u = 0; u < SP; ++u) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { sl[j] += q[c][2 * u][j]; sl[4 + j] += q[c][2 * u + 1][j]; }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) hr[c][j] = pack_bf16x2(sl[2 * j] + v[2 * j], sl[2 * j + 1] + v[2 * j + 1]);
    float ssq = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { const float lo = bf16lo_to_f32(hr[c][j]), hi = bf16hi_to_f32(hr[c][j]); ssq += lo * lo + hi * hi; }
    const bool on = c * 512 + tid < nch;
    if (!on) ssq = 0.f;                      // clamped duplicate chunk
    ssq = group16_sum(ssq);                 // same operand pairs as the xor butterfly of skinny_kernel: identical bits
    if ((lane & 15) == 0 && on) gs[mm[c] * gstride + (cc[c] >> 7)] = ssq;
  }
  __syncthreads();
  CH_STAMP(1);
  // ---- phase 2: row total in a fixed order, normalise the thread's own values, bf16 activations -> LDS
#pragma unroll
  for (int c = 0; c < CPT; ++c) {
    float tot = 0.f;
    const float* gr = gs + mm[c] * gstride;
#pragma unroll
    for (int i = 0; i < ngr; i += 4) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(gr + i);
#pragma unroll
      for (int j = 0; j < 4; ++j) tot += (i + j < ngr) ? t[j] : 0.f;
    }
    const float rs = rsqrtf(tot / (float)K + p.eps);
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float lo = round_bf16(bf16lo_to_f32(hr[c][j]) * rs) * bf16lo_to_f32(wv[c][j]);
      const float hi = round_bf16(bf16hi_to_f32(hr[c][j]) * rs) * bf16hi_to_f32(wv[c][j]);
      o[j] = pack_bf16x2(lo, hi);
    }
    if (c * 512 + tid < nch) *reinterpret_cast<u32x4*>(xs + mm[c] * XS + cc[c] * 2) = o;
  }
  __syncthreads();          // (the group sums alias `red`: every thread has read them before any wave writes a partial accumulator)
  CH_STAMP(2);

  // ---- units: in-workgroup split-K over the 8 waves, reduced through LDS by wave 0 (same order as skinny_kernel: bit-identical)
  const bool mok = fr < M;
  const char* xrow = xs + (mok ? fr : 0) * XS + (wave * (NS * 32) + g * 8) * 2;       // lanes of rows >= M shadow row 0: MFMA columns are independent, their results are never stored
  bf16x8 xf[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) xf[s] = as_bf16x8(*reinterpret_cast<const u32x4*>(xrow + s * 64));
  // Every unit's MFMAs as its fragments land (issue order), ALL eight waves leave their partial tiles in LDS, ONE barrier, then wave u finishes unit u: sum of the
  // eight K-slices in wave order (the order of skinny_kernel: wave 0's tile first -- identical bits) + SwiGLU + store.  r05b timeline: with wave 0 finishing the units
  // one after the other behind a barrier each, 1.6-2.1 us passed between "prologue done" and the end of the workgroup -- the weights had landed long before.
#pragma unroll
  for (int u = 0; u < UE; ++u) {
    if (u < ucount) {
      f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        a0 = mfma16(as_bf16x8(w[u][TPU * s]), xf[s], a0);
        if constexpr (TPU == 2) a1 = mfma16(as_bf16x8(w[u][2 * s + 1]), xf[s], a1);
      }
      if (u == 0) { if constexpr (DBG) asm volatile("" ::"v"(a0[0])); CH_STAMP(3); }
      float* r = red + (((u * 8 + wave) * TPU) * 64 + lane) * 4;
      *reinterpret_cast<f32x4*>(r) = a0;
      if constexpr (TPU == 2) *reinterpret_cast<f32x4*>(r + 64 * 4) = a1;
    }
  }
  __syncthreads();
  CH_STAMP(4);
  if (wave < UE && wave < ucount) {
    const int u = wave;
    f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
#pragma unroll
    for (int w2 = 0; w2 < 8; ++w2) {
      const float* r = red + (((u * 8 + w2) * TPU) * 64 + lane) * 4;
      if (w2 == 0) {
        a0 = *reinterpret_cast<const f32x4*>(r);
        if constexpr (TPU == 2) a1 = *reinterpret_cast<const f32x4*>(r + 64 * 4);
      } else {
        a0 += *reinterpret_cast<const f32x4*>(r);
        if constexpr (TPU == 2) a1 += *reinterpret_cast<const f32x4*>(r + 64 * 4);
      }
    }
    const int unit = ustart + u;
    if constexpr (TPU == 2) {
      if (mok && unit * 32 < p.n_valid) {              // unit = [gate16 | up16] -> output columns unit*16 + g*4 + j
        bf16_t* o = p.out + (size_t)fr * p.ldo + unit * 16 + g * 4;
        float r4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) r4[j] = round_bf16(silu(round_bf16(a0[j]))) * round_bf16(a1[j]);
        *reinterpret_cast<u32x2*>(o) = u32x2{pack_bf16x2(r4[0], r4[1]), pack_bf16x2(r4[2], r4[3])};
      }
    } else {
      if (mok && unit * 16 < p.n_valid) {              // lane group g: [gate 2g, gate 2g+1, up 2g, up 2g+1] -> output columns unit*8 + 2g, + 1
        bf16_t* o = p.out + (size_t)fr * p.ldo + unit * 8 + g * 2;
        const float r0 = round_bf16(silu(round_bf16(a0[0]))) * round_bf16(a0[2]);
        const float r1 = round_bf16(silu(round_bf16(a0[1]))) * round_bf16(a0[3]);
        *reinterpret_cast<uint32_t*>(o) = pack_bf16x2(r0, r1);
      }
    }
  }
  // the rounded residual stream (the down projection adds it back): workgroup 0, last -- nothing in this launch waits for these stores
  if (p.h_out && blockIdx.x == 0) {
#pragma unroll
    for (int c = 0; c < CPT; ++c)
      if (c * 512 + tid < nch) st_global_16(p.h_out + mm[c] * K + cc[c], hr[c]);
  }
  CH_STAMP(5);
}

template <int NS, int UE, int SP, int CPT, int TPU>
static int chain_gu_launch(const ChainGuP& p, int gx, int lds, hipStream_t stream) {
  if (p.dbg) {
    if (int rc = set_max_lds_once(chain_gu_kernel<NS, UE, SP, CPT, TPU, true>, lds)) return rc;
    hipLaunchKernelGGL((chain_gu_kernel<NS, UE, SP, CPT, TPU, true>), dim3(gx), dim3(512), lds, stream, p);
  } else {
    if (int rc = set_max_lds_once(chain_gu_kernel<NS, UE, SP, CPT, TPU, false>, lds)) return rc;
    hipLaunchKernelGGL((chain_gu_kernel<NS, UE, SP, CPT, TPU, false>), dim3(gx), dim3(512), lds, stream, p);
  }
  VL_LAUNCH_CHECK();
  return 0;
}

/* tpu = tiles per unit of the packed weights (2: pack_gate_up, 1: pack_gate_up8).
   ONE list of the built variants {K / 256, units per workgroup, producer slabs, tiles per unit}: the predicate and the dispatch below are both generated from it
   (the predicate admitted K = 768 / 2 slabs / 2 units per workgroup, which has no instantiation). */
#define CG_VARIANTS(X) X(3, 3, 3, 2) X(3, 3, 2, 2) X(6, 3, 2, 2) X(3, 2, 3, 2) X(6, 2, 2, 2) X(3, 5, 3, 1) X(6, 5, 2, 1)
extern "C" int vlaser_chain_gu_supported(int M, int N, int K, int n_partials, int tpu) {
  if (M < 1 || M > 16 || K % 256 || (tpu != 1 && tpu != 2) || N < 16 * tpu || N % (16 * tpu)) return 0;
  const int cpt = (M * (K / 8) + 511) / 512;
  if (cpt > 3) return 0;
  const int ns = K / 256, units = N / (16 * tpu), longest = (units + 255) / 256;
#define CG_HAS(NS_, UE_, SP_, TPU_) if (ns == NS_ && longest == UE_ && n_partials == SP_ && tpu == TPU_) return 1;
  CG_VARIANTS(CG_HAS)
#undef CG_HAS
  return 0;
}

extern "C" int vlaser_chain_gu(const VlaserSkinnyArgs* a, vl_stream_t s) {
  VL_CHECK(a && a->x && a->W && a->norm_w && a->out && a->partials, "vlaser_chain_gu: null operand");
  const int tpu = a->tiles_per_unit == 1 ? 1 : 2;
  VL_CHECK(a->tiles_per_unit == 2 || a->tiles_per_unit == 0 || a->tiles_per_unit == 1, "vlaser_chain_gu: weights packed in 32-row (pack_gate_up) or 16-row (pack_gate_up8) units");
  VL_CHECK(a->k_splits == 1 && vlaser_chain_gu_supported(a->M, a->N, a->K, a->n_partials, tpu),
           "vlaser_chain_gu: built for K = 768 / 1536, 2 or 3 producer slabs, 2-3 (32-row) or 5 (16-row) units per workgroup, M * K / 8 <= 1536 chunks (got M %d N %d K %d slabs %d)", a->M, a->N,
           a->K, a->n_partials);
  VL_CHECK(((uintptr_t)a->W & 15) == 0 && ((uintptr_t)a->x & 15) == 0 && ((uintptr_t)a->partials & 15) == 0 && ((uintptr_t)a->out & 7) == 0 && a->ldo % 4 == 0, "vlaser_chain_gu: alignment");
  const int n_valid = a->n_valid > 0 ? a->n_valid : a->N;
  VL_CHECK(n_valid % (16 * tpu) == 0, "vlaser_chain_gu: whole [gate|up] groups");
  ChainGuP p;
  p.h_in = (const bf16_t*)a->x; p.partials = a->partials; p.norm_w = (const bf16_t*)a->norm_w; p.h_out = (bf16_t*)a->h_out;
  p.W = (const u32x4*)a->W; p.out = (bf16_t*)a->out;
  p.M = a->M; p.ldo = a->ldo; p.n_valid = n_valid; p.eps = a->eps; p.inv_cpr = 8.0f / (float)a->K; p.dbg = a->dbg;
  const int units = a->N / (16 * tpu), longest = (units + 255) / 256, gx = (units + longest - 1) / longest;
  p.ulo = units / gx; p.urem = units % gx;
  const int ns = a->K / 256, cpt = (a->M * (a->K / 8) + 511) / 512;
  const int lds = ((a->M * (a->K * 2 + 16) + 15) & ~15) + longest * 8 * tpu * 64 * 16;
  hipStream_t stream = (hipStream_t)s;
#define CG_CASE(NS_, UE_, SP_, CPT_, TPU_) if (ns == NS_ && longest == UE_ && a->n_partials == SP_ && cpt == CPT_ && tpu == TPU_) return chain_gu_launch<NS_, UE_, SP_, CPT_, TPU_>(p, gx, lds, stream);
#define CG_CPT(NS_, UE_, SP_, TPU_) CG_CASE(NS_, UE_, SP_, 1, TPU_) CG_CASE(NS_, UE_, SP_, 2, TPU_) CG_CASE(NS_, UE_, SP_, 3, TPU_)
  CG_VARIANTS(CG_CPT)
#undef CG_CPT
#undef CG_CASE
  vlaser_set_error("vlaser_chain_gu: no variant for K %d, %d units per workgroup, %d slabs, %d chunks per thread, %d tiles per unit", a->K, longest, a->n_partials, cpt, tpu);
  return -1;
}

// ------------------------------------------------------------------------------------------------------------------ q / k / v
struct ChainQkvP {
  const bf16_t* h_in; const float* partials; bf16_t* h_out; const bf16_t* norm_w; const u32x4* W; const bf16_t* bias;
  bf16_t* q_out; bf16_t* k_cache; bf16_t* vt_cache; const float* rope_cos; const float* rope_sin; const int32_t* pos_ids;
  int M, n_q_heads, n_kv_heads, s_max, tok_per_batch, slot_base;
  float eps;
  unsigned long long* dbg;
};

// KS = K / 32 K-steps (24: hidden 768, 48: hidden 1536), RG = ceil(M / 4) row groups of the norm prologue.  One wave = one 16-row unit of the fused q/k/v
// matrix (head_perm16 packing: lane group g of a unit holds [d, d+1, d+64, d+65], d = 8 (unit % 8) + 2 g -- the RoPE partner of every value in the same lane).
// SP = 0: h_in is the residual stream itself; SP = 2: h = bf16(h_in + slab 0 + slab 1), the two K halves of vlaser_chain_down2 (slabs first, residual last: the
// order of every other seam here), and unit 0's wave stores h for the o_proj -> gate/up seam.
// NWV = 2 (hidden 1536, the greedy decode): two waves per unit, each over one K half -- 24 weight requests and 24 dependent MFMAs per wave instead of 48 (6.6 -> 5.x us per
// launch); both waves load the whole row (3 KB from L2) for the row's sum of squares and normalise their own half into their own LDS rows (still wave-local); wave 1 hands its
// accumulator to wave 0 through LDS (the launch's only barrier).
template <int KS, int RG, int SP, int NWV, bool DBG>
__global__ __launch_bounds__(64 * NWV) void chain_qkv_kernel(ChainQkvP p) {
  constexpr int K = KS * 32, NCH = KS / 4, KSW = KS / NWV, NCW = NCH / NWV, XS = (K / NWV) * 2 + 16;        // NCH: 16-byte chunks of a row per lane of its 16-lane group
  static_assert(NWV == 1 || ((NWV == 2 || NWV == 4) && SP == 0 && KS % (4 * NWV) == 0), "several waves: plain residual stream, whole chunks per K part");
  extern __shared__ __attribute__((aligned(16))) char smem_q[];    // [NWV][4 RG][XS] normalised activations (bf16) | NWV = 2: f32x4[64] of wave 1's accumulator
  const int lane = threadIdx.x & 63, wv = NWV == 1 ? 0 : (int)(threadIdx.x >> 6), fr = lane & 15, g = lane >> 4, r4 = lane >> 4, j16 = lane & 15;
  char* xs = smem_q + wv * (4 * RG * XS);
  const int unit = blockIdx.x, M = p.M;
  unsigned long long* dbg = DBG ? p.dbg + (size_t)blockIdx.x * 8 : nullptr;
  CH_STAMP(0);
  // ---- requests, in issue order: position id of the lane's row (cos / sin hang off it), norm weight + residual stream (L2), epilogue bias, weights (HBM)
  const int pos = p.pos_ids[min(fr, M - 1)];
  u32x4 wn[NCH], hc[RG][NCH];
  f32x4 q[RG][NCH][SP > 0 ? 2 * SP : 1];
#pragma unroll
  for (int i = 0; i < NCH; ++i) wn[i] = ld_global_16(p.norm_w + (j16 + 16 * i) * 8);
#pragma unroll
  for (int rg = 0; rg < RG; ++rg) {
    const int row = min(rg * 4 + r4, M - 1);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int off = row * K + (j16 + 16 * i) * 8;
      hc[rg][i] = ld_global_16(p.h_in + off);
#pragma unroll
      for (int u = 0; u < SP; ++u) {
        const float* pp = p.partials + (off + u * (M * K));
        q[rg][i][2 * u] = *reinterpret_cast<const f32x4*>(pp);
        q[rg][i][2 * u + 1] = *reinterpret_cast<const f32x4*>(pp + 4);
      }
    }
  }
  const u32x2 bias = *reinterpret_cast<const u32x2*>(p.bias + unit * 16 + g * 4);
  __builtin_amdgcn_sched_barrier(0);           // pin the issue order: vmcnt retires in order, the scheduler does not know
  u32x4 w[KSW];
  {
    const u32x4* src = p.W + (size_t)unit * (KS * 64) + wv * (KSW * 64) + lane;
#pragma unroll
    for (int f = 0; f < KSW; ++f) w[f] = __builtin_nontemporal_load(src + f * 64);
  }
  __builtin_amdgcn_sched_barrier(0);
  const int d = ((unit & 7) << 3) + 2 * g;
  int pos_l = pos;
  asm volatile("" : "+v"(pos_l));              // everything computed from the position id stays BEHIND the weight requests (hipcc hoisted its sign extension -- and with it
                                               // a wait for the id's round trip -- in front of them)
  const f32x2_t cs = *reinterpret_cast<const f32x2_t*>(p.rope_cos + (size_t)pos_l * 64 + d);     // behind the weights in the queue: only the epilogue needs them
  const f32x2_t sn = *reinterpret_cast<const f32x2_t*>(p.rope_sin + (size_t)pos_l * 64 + d);
  __builtin_amdgcn_sched_barrier(0);

  // ---- RMSNorm: xn = bf16(w * bf16(h * rsqrt(mean(h^2) + eps))), one row per 16-lane group and row group; wave-local (no barrier)
#pragma unroll
  for (int rg = 0; rg < RG; ++rg) {
    if constexpr (SP > 0) {
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        float sl[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < SP; ++u)
#pragma unroll
          for (int e = 0; e < 4; ++e) { sl[e] += q[rg][i][2 * u][e]; sl[4 + e] += q[rg][i][2 * u + 1][e]; }
#pragma unroll
        for (int e = 0; e < 4; ++e) hc[rg][i][e] = pack_bf16x2(sl[2 * e] + bf16lo_to_f32(hc[rg][i][e]), sl[2 * e + 1] + bf16hi_to_f32(hc[rg][i][e]));
        if (blockIdx.x == 0 && rg * 4 + r4 < M) st_global_16(p.h_out + (size_t)(rg * 4 + r4) * K + (j16 + 16 * i) * 8, hc[rg][i]);
      }
    }
    float ssq = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float lo = bf16lo_to_f32(hc[rg][i][e]), hi = bf16hi_to_f32(hc[rg][i][e]); ssq += lo * lo + hi * hi; }
    ssq = group16_sum(ssq);
    const float rs = rsqrtf(ssq / (float)K + p.eps);
    const int row = rg * 4 + r4;
#pragma unroll
    for (int i2 = 0; i2 < NCW; ++i2) {
      u32x4 hv, wv4;
      if constexpr (NWV == 1) { hv = hc[rg][i2]; wv4 = wn[i2]; }
      else {                                     // this wave's K part: chunks [wv NCW, (wv + 1) NCW) -- selected without indexing the register arrays dynamically
        hv = hc[rg][i2]; wv4 = wn[i2];
#pragma unroll
        for (int v = 1; v < NWV; ++v)
#pragma unroll
          for (int e = 0; e < 4; ++e) { hv[e] = wv == v ? hc[rg][v * NCW + i2][e] : hv[e]; wv4[e] = wv == v ? wn[v * NCW + i2][e] : wv4[e]; }
      }
      u32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float lo = round_bf16(bf16lo_to_f32(hv[e]) * rs) * bf16lo_to_f32(wv4[e]);
        const float hi = round_bf16(bf16hi_to_f32(hv[e]) * rs) * bf16hi_to_f32(wv4[e]);
        o[e] = pack_bf16x2(lo, hi);
      }
      if (row < M) *reinterpret_cast<u32x4*>(xs + row * XS + (j16 + 16 * i2) * 16) = o;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  CH_STAMP(2);
  // ---- GEMV over the whole K: two accumulators so that consecutive MFMAs do not wait for each other
  const bool mok = fr < M;
  const char* xrow = xs + (mok ? fr : 0) * XS + g * 16;
  f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
#pragma unroll
  for (int f = 0; f < KSW; f += 2) {
    const u32x4 x0 = *reinterpret_cast<const u32x4*>(xrow + f * 64), x1 = *reinterpret_cast<const u32x4*>(xrow + (f + 1) * 64);     // rows >= M shadow row 0 (never stored)
    acc0 = mfma16(as_bf16x8(w[f]), as_bf16x8(x0), acc0);
    acc1 = mfma16(as_bf16x8(w[f + 1]), as_bf16x8(x1), acc1);
  }
  f32x4 acc;
#pragma unroll
  for (int e = 0; e < 4; ++e) acc[e] = acc0[e] + acc1[e];
  if constexpr (DBG) asm volatile("" ::"v"(acc[0]));
  CH_STAMP(3);
  if constexpr (NWV > 1) {                       // out = sum of the K parts in wave order, wave 0 finishes
    f32x4* red = reinterpret_cast<f32x4*>(smem_q + NWV * (4 * RG * XS));
    if (wv != 0) red[(wv - 1) * 64 + lane] = acc;
    __syncthreads();
    if (wv != 0) return;
#pragma unroll
    for (int v = 1; v < NWV; ++v) {
      const f32x4 o = red[(v - 1) * 64 + lane];
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] += o[e];
    }
  }
  // ---- epilogue: bias + RoPE + scatter (as skinny_epilogue16<QKV_ROPE>): lane -> row m = fr, values [d, d+1, d+64, d+65] of head unit / 8
  if (mok) {
    const int m = fr, head = unit >> 3;
    const float b0 = bf16lo_to_f32(bias[0]), b1 = bf16hi_to_f32(bias[0]), b2 = bf16lo_to_f32(bias[1]), b3 = bf16hi_to_f32(bias[1]);
    const float x1[2] = {round_bf16(acc[0] + b0), round_bf16(acc[1] + b1)}, x2[2] = {round_bf16(acc[2] + b2), round_bf16(acc[3] + b3)};
    const int b = ch_fdiv(m, __builtin_amdgcn_rcpf((float)p.tok_per_batch));
    const int slot = p.slot_base >= 0 ? p.slot_base + (m - b * p.tok_per_batch) : pos_l;
    const int nq = p.n_q_heads, nkv = p.n_kv_heads;
    if (head < nq + nkv) {
      float o1[2], o2[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        o1[e] = x1[e] * cs[e] - x2[e] * sn[e];
        o2[e] = x2[e] * cs[e] + x1[e] * sn[e];
      }
      bf16_t* dst = head < nq ? p.q_out + (size_t)m * nq * 128 + head * 128 : p.k_cache + (((size_t)b * nkv + (head - nq)) * p.s_max + slot) * 128;
      *reinterpret_cast<uint32_t*>(dst + d) = pack_bf16x2(o1[0], o1[1]);
      *reinterpret_cast<uint32_t*>(dst + d + 64) = pack_bf16x2(o2[0], o2[1]);
    } else {
      bf16_t* vt = p.vt_cache + ((size_t)b * nkv + (head - nq - nkv)) * 128 * p.s_max + slot;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        vt[(size_t)(d + e) * p.s_max] = f32_to_bf16(x1[e]);
        vt[(size_t)(d + 64 + e) * p.s_max] = f32_to_bf16(x2[e]);
      }
    }
  }
  CH_STAMP(5);
}

extern "C" int vlaser_chain_qkv_supported(int M, int N, int K) { return M >= 1 && ((K == 768 && M <= 16) || (K == 1536 && M <= 8)) && N % 128 == 0; }      // (hidden 1536 x 16 rows would spill)
static int g_qkv_waves = 0;            // tools / tests: 1 forces the one-wave kernel at hidden 1536 (same-box A/B, order-of-summation tests); 0 = the defaults below
extern "C" int vlaser_chain_qkv_set_waves(int n) { const int prev = g_qkv_waves; if (n == 0 || n == 1) g_qkv_waves = n; return prev; }
/* with the two fp32 slabs of vlaser_chain_down2 as input (n_partials = 2) */
extern "C" int vlaser_chain_qkv2_supported(int M, int N, int K) { return M >= 1 && ((K == 768 && M <= 8) || (K == 1536 && M <= 4)) && N % 128 == 0; }

extern "C" int vlaser_chain_qkv(const VlaserSkinnyArgs* a, vl_stream_t s) {
  VL_CHECK(a && a->x && a->W && a->norm_w && a->bias && a->q_out && a->k_cache && a->vt_cache && a->rope_cos && a->rope_sin && a->pos_ids, "vlaser_chain_qkv: null operand");
  VL_CHECK(a->tiles_per_unit == 1 && a->k_splits == 1, "vlaser_chain_qkv: weights packed in 16-row lane-local units (pack_qkv16 + pack_skinny(..., 1, 1))");
  VL_CHECK(a->n_partials == 0 || (a->n_partials == 2 && a->partials && a->h_out), "vlaser_chain_qkv: reads the residual stream as published by vlaser_chain_down (no slabs) or x + the two slabs of vlaser_chain_down2 (then h_out receives the rounded sum)");
  VL_CHECK(a->n_partials == 0 ? vlaser_chain_qkv_supported(a->M, a->N, a->K) : vlaser_chain_qkv2_supported(a->M, a->N, a->K),
           "vlaser_chain_qkv: built for hidden 768 / 1536, M <= 16 (8 / 4 with slabs), whole heads (got M %d N %d K %d slabs %d)", a->M, a->N, a->K, a->n_partials);
  VL_CHECK(((uintptr_t)a->W & 15) == 0 && ((uintptr_t)a->x & 15) == 0 && ((uintptr_t)a->norm_w & 15) == 0 && ((uintptr_t)a->bias & 7) == 0, "vlaser_chain_qkv: alignment");
  ChainQkvP p;
  p.h_in = (const bf16_t*)a->x; p.partials = a->partials; p.h_out = (bf16_t*)a->h_out; p.norm_w = (const bf16_t*)a->norm_w; p.W = (const u32x4*)a->W; p.bias = (const bf16_t*)a->bias;
  p.q_out = (bf16_t*)a->q_out; p.k_cache = (bf16_t*)a->k_cache; p.vt_cache = (bf16_t*)a->vt_cache; p.rope_cos = a->rope_cos; p.rope_sin = a->rope_sin;
  p.pos_ids = a->pos_ids; p.M = a->M; p.n_q_heads = a->n_q_heads; p.n_kv_heads = a->n_kv_heads; p.s_max = a->s_max; p.tok_per_batch = a->tok_per_batch;
  p.slot_base = a->slot_base; p.eps = a->eps; p.dbg = a->dbg;
  const int units = a->N / 16, rg = (a->M + 3) / 4, ks = a->K / 32;
  // waves per unit (K parts): hidden 1536 two (measured in the decode step: 0.9745 -> 0.9595 ms per token), hidden 768 one
  const int nwv = (a->n_partials == 0 && ks == 48 && g_qkv_waves != 1) ? 2 : 1;
  const int lds = nwv * 4 * rg * (a->K / nwv * 2 + 16) + (nwv - 1) * 64 * 16;
  hipStream_t stream = (hipStream_t)s;
#define CQ_LAUNCH(KS_, RG_, SP_, NWV_, DBG_)                                                                            \
  do {                                                                                                                  \
    if (int rc = set_max_lds_once(chain_qkv_kernel<KS_, RG_, SP_, NWV_, DBG_>, lds)) return rc;                          \
    hipLaunchKernelGGL((chain_qkv_kernel<KS_, RG_, SP_, NWV_, DBG_>), dim3(units), dim3(64 * NWV_), lds, stream, p);     \
    VL_LAUNCH_CHECK();                                                                                                  \
    return 0;                                                                                                           \
  } while (0)
#define CQ_CASEW(KS_, RG_, SP_, NWV_) if (ks == KS_ && rg == RG_ && a->n_partials == SP_ && nwv == NWV_) { if (p.dbg) CQ_LAUNCH(KS_, RG_, SP_, NWV_, true); else CQ_LAUNCH(KS_, RG_, SP_, NWV_, false); }
#define CQ_CASE(KS_, RG_, SP_) CQ_CASEW(KS_, RG_, SP_, 1)
  CQ_CASE(24, 1, 0) CQ_CASE(24, 2, 0) CQ_CASE(24, 3, 0) CQ_CASE(24, 4, 0) CQ_CASE(48, 1, 0) CQ_CASE(48, 2, 0) CQ_CASE(24, 1, 2) CQ_CASE(24, 2, 2) CQ_CASE(48, 1, 2)
  CQ_CASEW(48, 1, 0, 2) CQ_CASEW(48, 2, 0, 2)      // (measured and not kept: four waves at hidden 1536 0.9709 vs 0.9697 ms per token; two waves at hidden 768 11.656 vs 11.645 ms per chunk)
#undef CQ_CASEW
#undef CQ_CASE
#undef CQ_LAUNCH
  vlaser_set_error("vlaser_chain_qkv: no variant");
  return -1;
}

// ------------------------------------------------------------------------------------------------------------------ down projection
struct ChainDownP {
  const bf16_t* x; const u32x4* W; const bf16_t* res; bf16_t* h_out; float* out_f32;
  int M, N, ldx;
  unsigned long long* dbg;
};

// A workgroup = CG groups of COLS (3 or 4) output columns over the WHOLE K = NW waves x NL loads x 128.  Lane (b = lane >> 2, i = lane & 3) of wave w, load l, group c
// holds W[n0 + c COLS + min(i, COLS - 1)][(w NL + l) 128 + 8 b .. + 8] (ops.pack_down4: one contiguous 16 x COLS x 16 bytes per wave-level load; with COLS = 3 the
// fourth lane of a block re-reads the third's 16 bytes) and x[row i of its row group][same k]; the two halves of the 16 bytes feed two v_mfma_f32_4x4x4_16b_bf16
// (layout probed on the GPU, tools/micro/mfma4_probe.hip: D[reg = lane & 3 of the A lane][lane of the B lane]; block b: D_b[n][m] += sum_k W[n0 + n][k] x[m][k] over
// 4 k): lane (b, m) accumulates out[m][n0 .. n0 + COLS) over its block's k, the 16 blocks are summed across the lanes, the NW waves through LDS in a fixed order.
// COLS = 3 puts N = 768 on 256 workgroups: a CU sustains ~35 GB/s of requests, so the launch lasts as long as the bytes of its busiest CU -- 54 KB of weights + the
// whole [M, K] activation (every workgroup contracts over all of K: the price of publishing the result once) instead of 72 + 71 KB on 192 CUs.
// RG = ceil(M / 4) row groups.
// KS = 2 (vlaser_chain_down2, grid.y = K half): every byte a CU pulls costs the same ~17 ns / KB whether it comes from HBM or from L2 (r05d timeline: 71 -> 125 ->
// 143 KB per CU = 2.7 -> 3.75 -> 3.95 us), and at KS = 1 more than half of a CU's bytes are the activation.  Two K halves x 128 six-column workgroups = 256
// workgroups of 54 KB weights + HALF the activation; the halves leave two fp32 slabs [2][M][N] that the q/k/v launch (one wave per unit: 24 more L2 loads per lane,
// inside the shadow of its weight round trip) and the next gate/up's residual path reduce -- no bf16 rounding here, the consumer rounds h = bf16(res + s0 + s1).
template <int NW, int NL, int RG, int COLS, int CG, int KS, bool DBG>
__global__ __launch_bounds__(NW * 64) void chain_down_kernel(ChainDownP p) {
  __shared__ __attribute__((aligned(16))) float part[NW][CG][RG][4][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = lane >> 2, i = lane & 3;
  const int M = p.M, n0 = blockIdx.x * (COLS * CG), kh = KS > 1 ? blockIdx.y : 0;
  unsigned long long* dbg = DBG ? p.dbg + ((size_t)kh * gridDim.x + blockIdx.x) * 8 : nullptr;
  CH_STAMP(0);
  // ---- requests: residual (final phase: thread t -> row t / CG... see below), activations (L2), weights (HBM)
  const int fr_row = min(tid / CG, M - 1), fr_cg = tid % CG;            // final phase: thread t < M * CG finishes row t / CG of column group t % CG
  float rres[COLS];
  if constexpr (KS == 1) {
    const bf16_t* rp = p.res + (size_t)fr_row * p.N + n0 + fr_cg * COLS;
#pragma unroll
    for (int c = 0; c < COLS; ++c) rres[c] = bf16_to_f32(rp[c]);          // unconditional (clamped): no branch in front of the load burst
  }
  u32x4 xv[RG][NL], wv[CG][NL];
#pragma unroll
  for (int rg = 0; rg < RG; ++rg) {
    const bf16_t* xr = p.x + (size_t)min(rg * 4 + i, M - 1) * p.ldx + ((kh * NW + wave) * NL) * 128 + b * 8;
#pragma unroll
    for (int l = 0; l < NL; ++l) xv[rg][l] = ld_global_16(xr + l * 128);
  }
  __builtin_amdgcn_sched_barrier(0);           // activations (L2) strictly in front of the weights (HBM) in the queue
  {
    const u32x4* src = p.W + ((((size_t)kh * gridDim.x + blockIdx.x) * NW + wave) * NL) * (CG * 16 * COLS) + b * COLS + min(i, COLS - 1);
#pragma unroll
    for (int l = 0; l < NL; ++l)
#pragma unroll
      for (int c = 0; c < CG; ++c) wv[c][l] = __builtin_nontemporal_load(src + (l * CG + c) * (16 * COLS));
  }
  __builtin_amdgcn_sched_barrier(0);
  f32x4 acc[CG][RG];
#pragma unroll
  for (int c = 0; c < CG; ++c)
#pragma unroll
    for (int rg = 0; rg < RG; ++rg) acc[c][rg] = f32x4{0, 0, 0, 0};
#pragma unroll
  for (int l = 0; l < NL; ++l) {
#pragma unroll
    for (int c = 0; c < CG; ++c) {
      const s16x4 wlo = __builtin_bit_cast(s16x4, u32x2{wv[c][l][0], wv[c][l][1]}), whi = __builtin_bit_cast(s16x4, u32x2{wv[c][l][2], wv[c][l][3]});
#pragma unroll
      for (int rg = 0; rg < RG; ++rg) {
        const s16x4 xlo = __builtin_bit_cast(s16x4, u32x2{xv[rg][l][0], xv[rg][l][1]}), xhi = __builtin_bit_cast(s16x4, u32x2{xv[rg][l][2], xv[rg][l][3]});
        acc[c][rg] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wlo, xlo, acc[c][rg], 0, 0, 0);
        acc[c][rg] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(whi, xhi, acc[c][rg], 0, 0, 0);
      }
    }
  }
  if constexpr (DBG) asm volatile("" ::"v"(acc[0][0][0]));
  CH_STAMP(3);
  // ---- sum of the 16 blocks (lanes with equal lane & 3), then of the waves
#pragma unroll
  for (int c = 0; c < CG; ++c)
#pragma unroll
    for (int rg = 0; rg < RG; ++rg) {
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[c][rg][r] = vl_blocks16_sum(acc[c][rg][r]);
      if (lane < 4) *reinterpret_cast<f32x4*>(&part[wave][c][rg][lane][0]) = acc[c][rg];       // [row i of the group][column n]
    }
  __syncthreads();
  CH_STAMP(4);
  if (tid < M * CG) {
    const int rg = fr_row >> 2, ri = fr_row & 3;
    float s4[4] = {0, 0, 0, 0};
#pragma unroll
    for (int w2 = 0; w2 < NW; ++w2) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(&part[w2][fr_cg][rg][ri][0]);
#pragma unroll
      for (int r = 0; r < 4; ++r) s4[r] += t[r];
    }
    if constexpr (KS == 1) {
      bf16_t* op = p.h_out + (size_t)fr_row * p.N + n0 + fr_cg * COLS;
#pragma unroll
      for (int c = 0; c < COLS; ++c) op[c] = f32_to_bf16(s4[c] + rres[c]);
    } else {
      float* op = p.out_f32 + ((size_t)kh * M + fr_row) * p.N + n0 + fr_cg * COLS;
#pragma unroll
      for (int c = 0; c < COLS; ++c) op[c] = s4[c];
    }
  }
  CH_STAMP(5);
}

extern "C" int vlaser_chain_down_geometry(int N, int* cols, int* groups);
extern "C" int vlaser_chain_down_supported(int M, int N, int K) {
  if (!(M >= 1 && M <= 16 && (N % 3 == 0 || N % 4 == 0) && K == 8960)) return 0;
  int c, g;
  vlaser_chain_down_geometry(N, &c, &g);
  return g == 1 || M <= 8;            // two column groups per workgroup x 16 rows would spill
}
/* column geometry of vlaser_chain_down / ops.pack_down4 for an output width N: cols per group (3 or 4) and groups per workgroup, chosen for ~256 workgroups */
extern "C" int vlaser_chain_down_geometry(int N, int* cols, int* groups) {
  const int c = (N % 3 == 0) ? 3 : 4;
  const int g = (N / c >= 512 && (N / c) % 2 == 0) ? 2 : 1;
  if (cols) *cols = c;
  if (groups) *groups = g;
  return N / (c * g);
}

/* x bf16 [M, ldx] (ldx >= K), W = ops.pack_down4(down_proj.weight), res / h_out bf16 [M, N] (may not alias): h_out = bf16(res + x @ W^T) */
extern "C" int vlaser_chain_down(const void* x, int ldx, const void* W, const void* res, void* h_out, int M, int N, int K, unsigned long long* dbg, vl_stream_t s) {
  VL_CHECK(x && W && res && h_out && res != h_out, "vlaser_chain_down: null operand (or res == h_out: every workgroup reads its residual columns while others store)");
  VL_CHECK(vlaser_chain_down_supported(M, N, K) && ldx >= K && ldx % 8 == 0, "vlaser_chain_down: built for K = 8960 (7 waves x 10 loads x 128), N %% 3 == 0 or N %% 4 == 0, M <= 16 (got M %d N %d K %d)", M, N, K);
  VL_CHECK(((uintptr_t)W & 15) == 0 && ((uintptr_t)x & 15) == 0, "vlaser_chain_down: alignment");
  ChainDownP p;
  p.x = (const bf16_t*)x; p.W = (const u32x4*)W; p.res = (const bf16_t*)res; p.h_out = (bf16_t*)h_out; p.out_f32 = nullptr; p.M = M; p.N = N; p.ldx = ldx; p.dbg = dbg;
  int cols, groups;
  const int wgs = vlaser_chain_down_geometry(N, &cols, &groups);
  const int rg = (M + 3) / 4;
  hipStream_t stream = (hipStream_t)s;
#define CD_LAUNCH(RG_, COLS_, CG_)                                                                                                       \
  do {                                                                                                                                   \
    if (dbg) hipLaunchKernelGGL((chain_down_kernel<7, 10, RG_, COLS_, CG_, 1, true>), dim3(wgs), dim3(7 * 64), 0, stream, p);              \
    else hipLaunchKernelGGL((chain_down_kernel<7, 10, RG_, COLS_, CG_, 1, false>), dim3(wgs), dim3(7 * 64), 0, stream, p);                 \
    VL_LAUNCH_CHECK();                                                                                                                   \
    return 0;                                                                                                                            \
  } while (0)
#define CD_CASE1(RG_) if (rg == RG_ && groups == 1) { if (cols == 3) CD_LAUNCH(RG_, 3, 1); else CD_LAUNCH(RG_, 4, 1); }
#define CD_CASE2(RG_) if (rg == RG_ && groups == 2) { if (cols == 3) CD_LAUNCH(RG_, 3, 2); else CD_LAUNCH(RG_, 4, 2); }
  CD_CASE1(1) CD_CASE1(2) CD_CASE1(3) CD_CASE1(4) CD_CASE2(1) CD_CASE2(2)
#undef CD_CASE1
#undef CD_CASE2
#undef CD_LAUNCH
  vlaser_set_error("vlaser_chain_down: no variant");
  return -1;
}

/* The same contraction split into two K halves (grid.y): out_f32 [2][M][N] fp32 slabs, no residual, no rounding.  W = ops.pack_down4(W, k_splits=2):
 * [K half][workgroup][7 waves][5 loads][2 groups][16 blocks][cols][8]; a workgroup owns 2 x cols output columns (cols = 3 when N % 6 == 0, else 4 with N % 8 == 0). */
extern "C" int vlaser_chain_down2_supported(int M, int N, int K) { return M >= 1 && M <= 8 && K == 8960 && (N % 6 == 0 || N % 8 == 0); }
extern "C" int vlaser_chain_down2(const void* x, int ldx, const void* W, float* out_f32, int M, int N, int K, unsigned long long* dbg, vl_stream_t s) {
  VL_CHECK(x && W && out_f32, "vlaser_chain_down2: null operand");
  VL_CHECK(vlaser_chain_down2_supported(M, N, K) && ldx >= K && ldx % 8 == 0, "vlaser_chain_down2: built for K = 8960 (2 halves x 7 waves x 5 loads x 128), N %% 6 == 0 or N %% 8 == 0, M <= 8 (got M %d N %d K %d)", M, N, K);
  VL_CHECK(((uintptr_t)W & 15) == 0 && ((uintptr_t)x & 15) == 0, "vlaser_chain_down2: alignment");
  ChainDownP p;
  p.x = (const bf16_t*)x; p.W = (const u32x4*)W; p.res = nullptr; p.h_out = nullptr; p.out_f32 = out_f32; p.M = M; p.N = N; p.ldx = ldx; p.dbg = dbg;
  const int cols = (N % 6 == 0) ? 3 : 4, wgs = N / (cols * 2), rg = (M + 3) / 4;
  hipStream_t stream = (hipStream_t)s;
#define CD2_LAUNCH(RG_, COLS_)                                                                                                           \
  do {                                                                                                                                   \
    if (dbg) hipLaunchKernelGGL((chain_down_kernel<7, 5, RG_, COLS_, 2, 2, true>), dim3(wgs, 2), dim3(7 * 64), 0, stream, p);               \
    else hipLaunchKernelGGL((chain_down_kernel<7, 5, RG_, COLS_, 2, 2, false>), dim3(wgs, 2), dim3(7 * 64), 0, stream, p);                  \
    VL_LAUNCH_CHECK();                                                                                                                   \
    return 0;                                                                                                                            \
  } while (0)
  if (rg == 1) { if (cols == 3) CD2_LAUNCH(1, 3); else CD2_LAUNCH(1, 4); }
  if (rg == 2) { if (cols == 3) CD2_LAUNCH(2, 3); else CD2_LAUNCH(2, 4); }
#undef CD2_LAUNCH
  vlaser_set_error("vlaser_chain_down2: no variant");
  return -1;
}

// ------------------------------------------------------------------------------------------------------------------ attention over the KV cache, <= 16 query tokens
// One WAVE per (kv head, key split, batch element), NC 32-key chunks each, no LDS, no barrier: Q of the kv head's (q head, token) rows (<= 32 rows = two MFMA
// tiles), then K and V^T of every chunk of the wave are requested in one burst (vlaser_attn_skinny staged the same work through 4-wave workgroups whose merge
// through LDS was 1.0 us of its 6.1 us, and its vector load of valid_len[b] put an `s_waitcnt vmcnt(0)` -- a second full round trip -- between the Q and the K / V^T
// requests).  A split leaves (m, l) fp32 and its NORMALISED output rows in bf16 (half the bytes of the fp32 partials for the 144 workgroups of the o_proj launch
// that each merge the splits of their K range); chain_oproj weights them by l 2^(m - max m).
struct ChainAttnP {
  const bf16_t* q; const bf16_t* k; const bf16_t* vt;
  float* pml; bf16_t* po;
  const int32_t* valid_len;
  long long q_bs, q_hs, q_ss, k_bs, k_hs, vt_bs, vt_hs;
  int sq, kv_len, G, ld_vt, mode, blk_start, first_tok_kv_len, n_splits, n_kv;
  float scale;
  unsigned long long* dbg;
  const float* mask; long long mask_bs, mask_rs;      // VL_ATTN_DENSE (ABI 8): fp32 additive mask, row = query TOKEN (the same for every head)
};

// DENSE: the visibility and a bias per (token, key) come from p.mask instead of the (valid_len, blk_start) descriptors; requested with the K / V^T of the chunks
template <int NC, bool DBG, bool DENSE = false>
__global__ __launch_bounds__(64) void chain_attn_kernel(ChainAttnP p) {
  constexpr int HD = 128, DC = 4, DT = 8;
  const int lane = threadIdx.x, fr = lane & 15, g = lane >> 4;
  const int kvh = blockIdx.x, split = blockIdx.y, b = blockIdx.z;
  unsigned long long* dbg = DBG ? p.dbg + (((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 : nullptr;
  CH_STAMP(0);
  const int nq = p.sq, nrows = p.G * nq;
  const bf16_t* K = p.k + (size_t)b * p.k_bs + (size_t)kvh * p.k_hs;
  const bf16_t* VT = p.vt + (size_t)b * p.vt_bs + (size_t)kvh * p.vt_hs;
  const int n_chunks = (p.kv_len + 31) >> 5;
  // ---- every request of the wave: Q rows (written by the previous launch), then K / V^T of its NC chunks (clamped: a chunk past the end is fully masked)
  u32x4 qv[2][DC];
  int row_hi2[2];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int r = min(qt * 16 + fr, nrows - 1);
    const int hg = (int)(((float)r + 0.5f) * __builtin_amdgcn_rcpf((float)nq)), tok = r - hg * nq;
    row_hi2[qt] = (tok == 0 && p.first_tok_kv_len > 0) ? p.first_tok_kv_len : 0x7fffffff;
    const bf16_t* Q = p.q + (size_t)b * p.q_bs + (size_t)(kvh * p.G + hg) * p.q_hs + (size_t)tok * p.q_ss;
#pragma unroll
    for (int dc = 0; dc < DC; ++dc) qv[qt][dc] = ld_global_16(Q + dc * 32 + g * 8);
  }
  u32x4 kf[NC][2][DC], vf[NC][DT];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int key0 = min(split * NC + c, n_chunks - 1) << 5;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int key = key0 + (fr >> 2) * 8 + t * 4 + (fr & 3);
#pragma unroll
      for (int dc = 0; dc < DC; ++dc) kf[c][t][dc] = ld_global_16(K + (size_t)min(key, p.kv_len - 1) * HD + dc * 32 + g * 8);
    }
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) vf[c][dt] = ld_global_16(VT + (size_t)(dt * 16 + fr) * p.ld_vt + key0 + g * 8);
  }
  f32x4 mk[DENSE ? 2 : 1][DENSE ? NC : 1][2];
  if constexpr (DENSE) {
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      const int r = min(qt * 16 + fr, nrows - 1);
      const int hg = (int)(((float)r + 0.5f) * __builtin_amdgcn_rcpf((float)nq)), tok = r - hg * nq;
      const float* mrow = p.mask + (size_t)b * p.mask_bs + (size_t)tok * p.mask_rs;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int key0 = min(split * NC + c, n_chunks - 1) << 5;
#pragma unroll
        for (int t = 0; t < 2; ++t) mk[qt][c][t] = *reinterpret_cast<const f32x4*>(mrow + key0 + g * 8 + t * 4);
      }
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  // (behind the requests: the two scalar round trips -- pointer, value -- of the valid length run while the vector loads are in flight)
  int lim1 = p.kv_len, lo2 = 0x7fffffff, hi2 = 0;
  if (p.mode == VL_ATTN_PREFIX && !DENSE) {
    lim1 = min(p.valid_len ? vl_sload_i32(p.valid_len + b) : p.kv_len, p.kv_len);      // scalar cache: not in the vmcnt queue
    lo2 = p.blk_start; hi2 = p.kv_len;
  }
  f32x4 o[2][DT];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int i = 0; i < DT; ++i) o[qt][i] = f32x4{0, 0, 0, 0};
  float m_run[2] = {-1.0e30f, -1.0e30f}, l_run[2] = {0.f, 0.f};
  const float sc = p.scale * 1.4426950408889634f;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int ci = split * NC + c;
    const int key0 = min(ci, n_chunks - 1) << 5;
    const bool live = ci < n_chunks;                         // wave-uniform
    bool visk[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = key0 + g * 8 + t * 4 + r;
        visk[t][r] = live & ((key < lim1) | ((key >= lo2) & (key < hi2)));        // bitwise: `&&` / `||` on lane-varying flags become exec-branch blocks
      }
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      f32x4 s[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        f32x4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int dc = 0; dc < DC; ++dc) acc = mfma16(as_bf16x8(kf[c][t][dc]), as_bf16x8(qv[qt][dc]), acc);
        s[t] = acc;
      }
      float mx = -1.0e30f;
      bool vis[2][4];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = key0 + g * 8 + t * 4 + r;
          vis[t][r] = visk[t][r] & ((key < lim1) | (key < row_hi2[qt]));
          s[t][r] *= sc;
          if constexpr (DENSE) {
            const float mv = mk[qt][c][t][r];
            vis[t][r] = vis[t][r] & (mv > -1.0e30f);
            s[t][r] = vis[t][r] ? __builtin_fmaf(mv, 1.4426950408889634f, s[t][r]) : s[t][r];
          }
          mx = vis[t][r] ? fmaxf(mx, s[t][r]) : mx;
        }
      mx = vl_xor32_max(vl_xor16_max(mx));
      const float m_new = fmaxf(m_run[qt], mx);
      const float alpha = __builtin_amdgcn_exp2f(m_run[qt] - m_new);
      m_run[qt] = m_new;
      float pv[8], psum = 0.f;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pe = vis[t][r] ? __builtin_amdgcn_exp2f(s[t][r] - m_new) : 0.f;
          psum += pe;
          pv[t * 4 + r] = pe;
        }
      l_run[qt] = l_run[qt] * alpha + psum;
      const u32x4 pk = {pack_bf16x2(pv[0], pv[1]), pack_bf16x2(pv[2], pv[3]), pack_bf16x2(pv[4], pv[5]), pack_bf16x2(pv[6], pv[7])};
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        f32x4 acc = o[qt][dt];
        acc[0] *= alpha; acc[1] *= alpha; acc[2] *= alpha; acc[3] *= alpha;
        o[qt][dt] = mfma16(as_bf16x8(vf[c][dt]), as_bf16x8(pk), acc);
      }
    }
  }
  if constexpr (DBG) asm volatile("" ::"v"(o[0][0][0]));
  CH_STAMP(1);
  // ---- the split's partial: (m, l) and the normalised rows (bf16); lane (fr, g) holds d = 16 dt + 4 g .. + 4 of row 16 qt + fr
  const size_t pidx = ((size_t)b * p.n_kv + kvh) * p.n_splits + split;
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const float l_tot = vl_xor32_sum(vl_xor16_sum(l_run[qt]));
    const int row = qt * 16 + fr;
    if (row < nrows) {
      if (g == 0) *reinterpret_cast<f32x2_t*>(p.pml + (pidx * 32 + row) * 2) = f32x2_t{m_run[qt], l_tot};
      const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
      bf16_t* dst = p.po + (pidx * 32 + row) * 128 + g * 4;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
        *reinterpret_cast<u32x2*>(dst + dt * 16) = u32x2{pack_bf16x2(o[qt][dt][0] * inv, o[qt][dt][1] * inv), pack_bf16x2(o[qt][dt][2] * inv, o[qt][dt][3] * inv)};
    }
  }
  CH_STAMP(3);
}

/* vlaser_attn_skinny's arguments (part_m = fp32 [B, n_kv, n_splits, 32, 2] (m, l) pairs, part_o = BF16 [B, n_kv, n_splits, 32, 128] normalised rows, part_l unused);
 * n_splits must equal ceil(ceil(kv_len / 32) / 2). */
extern "C" int vlaser_chain_attn_splits(int kv_len) { return (((kv_len + 31) >> 5) + 1) / 2; }
extern "C" int vlaser_chain_attn(const VlaserAttnArgs* a, vl_stream_t s) {
  VL_CHECK(a && a->q && a->k && a->vt && a->part_m && a->part_o, "vlaser_chain_attn: null pointer");
  VL_CHECK(a->head_dim == 128 && a->n_q_heads % a->n_kv_heads == 0 && a->sq >= 1 && a->sq * (a->n_q_heads / a->n_kv_heads) <= 32, "vlaser_chain_attn: head_dim 128, group * tokens <= 32");
  VL_CHECK(a->ld_vt % 32 == 0 && a->kv_len >= 1 && a->kv_len <= a->ld_vt && (a->mode == VL_ATTN_FULL || a->mode == VL_ATTN_PREFIX || a->mode == VL_ATTN_DENSE), "vlaser_chain_attn: bad cache geometry / mode");
  VL_CHECK(a->mode != VL_ATTN_DENSE || (a->mask && a->mask_rs % 4 == 0 && a->mask_rs >= ((a->kv_len + 31) & ~31) && (((uintptr_t)a->mask) & 15) == 0 && a->mask_bs % 4 == 0 && a->first_tok_kv_len == 0),
           "vlaser_chain_attn: VL_ATTN_DENSE needs a 16-byte aligned fp32 mask whose row stride is a multiple of 4 >= kv_len rounded up to 32 (and no riding first token)");
  VL_CHECK(a->n_splits == vlaser_chain_attn_splits(a->kv_len) && a->n_splits <= 16, "vlaser_chain_attn: n_splits must be ceil(chunks / 2) <= 16 (kv_len <= 1024)");
  ChainAttnP p;
  p.q = (const bf16_t*)a->q; p.k = (const bf16_t*)a->k; p.vt = (const bf16_t*)a->vt; p.pml = a->part_m; p.po = (bf16_t*)a->part_o; p.valid_len = a->valid_len;
  p.q_bs = a->q_bs; p.q_hs = a->q_hs; p.q_ss = a->q_ss; p.k_bs = a->k_bs; p.k_hs = a->k_hs; p.vt_bs = a->vt_bs; p.vt_hs = a->vt_hs;
  p.sq = a->sq; p.kv_len = a->kv_len; p.G = a->n_q_heads / a->n_kv_heads; p.ld_vt = a->ld_vt; p.mode = a->mode; p.blk_start = a->blk_start;
  p.first_tok_kv_len = a->first_tok_kv_len; p.n_splits = a->n_splits; p.n_kv = a->n_kv_heads; p.scale = a->scale; p.dbg = a->dbg;
  p.mask = a->mask; p.mask_bs = a->mask_bs; p.mask_rs = a->mask_rs;
  const dim3 grid(a->n_kv_heads, a->n_splits, a->batch);
  if (a->mode == VL_ATTN_DENSE) hipLaunchKernelGGL((chain_attn_kernel<2, false, true>), grid, dim3(64), 0, (hipStream_t)s, p);
  else if (p.dbg) hipLaunchKernelGGL((chain_attn_kernel<2, true>), grid, dim3(64), 0, (hipStream_t)s, p);
  else hipLaunchKernelGGL((chain_attn_kernel<2, false>), grid, dim3(64), 0, (hipStream_t)s, p);
  VL_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------------------------ o_proj: merge of the attention splits -> GEMV -> split-K slabs
struct ChainOprojP {
  const float* pml; const bf16_t* po; const u32x4* W; float* out;
  int M, N, n_units, G, nq, nkv, n_splits;
  float inv_cpr;
  unsigned long long* dbg;
};

// grid (N / 16 units, ks K-splits); NS = K / (ks * 256) K-steps per wave; the K range of a workgroup = whole heads.  S = attention splits REQUESTED (all up front): the
// launch has p.n_splits <= S of them (the Euler phase exactly 7 = its S; a decode step over kvmax keys 8..16) -- the requests beyond re-read the last split and get weight 0
template <int NS, int S, bool DBG>
__global__ __launch_bounds__(512) void chain_oproj_kernel(ChainOprojP p) {
  constexpr int KB = NS * 256, cpr = KB / 8, XS = KB * 2 + 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, g = lane >> 4;
  const int unit = blockIdx.x, ks = blockIdx.y, M = p.M;
  unsigned long long* dbg = DBG ? p.dbg + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 : nullptr;
  CH_STAMP(0);
  const int xs_bytes = (M * XS + 15) & ~15;
  char* xs = smem;
  float* red = reinterpret_cast<float*>(smem + xs_bytes);        // [7][64] f32x4
  const int nch = M * cpr;
  const int ch = min(tid, nch - 1);
  const int mm = ch_fdiv(ch, p.inv_cpr), j = ch - mm * cpr;
  const int k = ks * KB + j * 8, h = k >> 7, d = k & 127;
  const int b = ch_fdiv(mm, __builtin_amdgcn_rcpf((float)p.nq)), tok = mm - b * p.nq, kvh = ch_fdiv(h, __builtin_amdgcn_rcpf((float)p.G)), hg = h - kvh * p.G,
            r = hg * p.nq + tok;
  const int nsp = (S == 7) ? 7 : p.n_splits;
  const size_t pbase = ((size_t)b * p.nkv + kvh) * nsp;
  f32x2_t ml[S];
  u32x4 ov[S];
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int sc = (S == 7) ? s : min(s, nsp - 1);
    ml[s] = *reinterpret_cast<const f32x2_t*>(p.pml + ((pbase + sc) * 32 + r) * 2);
    ov[s] = ld_global_16(p.po + ((pbase + sc) * 32 + r) * 128 + d);
  }
  __builtin_amdgcn_sched_barrier(0);
  u32x4 w[NS];
  {
    const u32x4* src = p.W + (((size_t)ks * p.n_units + unit) * 8 + wave) * (NS * 64) + lane;
#pragma unroll
    for (int f = 0; f < NS; ++f) w[f] = __builtin_nontemporal_load(src + f * 64);
  }
  __builtin_amdgcn_sched_barrier(0);
  // ---- x[m][k] = sum_s w_s o_s / sum_s w_s,  w_s = l_s 2^(m_s - max m)
  float Mx = -1.0e30f;
#pragma unroll
  for (int s = 0; s < S; ++s) Mx = fmaxf(Mx, ml[s][0]);
  float Ls = 0.f, v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int s = 0; s < S; ++s) {
    float f = ml[s][1] * __builtin_amdgcn_exp2f(ml[s][0] - Mx);
    if (S != 7) f = s < nsp ? f : 0.f;
    Ls += f;
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[2 * e] += f * bf16lo_to_f32(ov[s][e]); v[2 * e + 1] += f * bf16hi_to_f32(ov[s][e]); }
  }
  const float inv = Ls > 0.f ? 1.0f / Ls : 0.f;
  if (tid < nch) *reinterpret_cast<u32x4*>(xs + mm * XS + j * 16) = u32x4{pack_bf16x2(v[0] * inv, v[1] * inv), pack_bf16x2(v[2] * inv, v[3] * inv), pack_bf16x2(v[4] * inv, v[5] * inv), pack_bf16x2(v[6] * inv, v[7] * inv)};
  // more than 512 pieces (decode batches of 6 .. 16 rows): the remaining ones in further passes of the same arithmetic, their requests no longer up front (one exposed L2 round
  // trip per pass; never taken by the Euler phase or the batch-1 decode)
  for (int c2 = tid + 512; c2 < nch; c2 += 512) {
    const int mm2 = ch_fdiv(c2, p.inv_cpr), j2 = c2 - mm2 * cpr;
    const int k2 = ks * KB + j2 * 8, h2 = k2 >> 7, d2 = k2 & 127;
    const int b2 = ch_fdiv(mm2, __builtin_amdgcn_rcpf((float)p.nq)), tok2 = mm2 - b2 * p.nq, kvh2 = ch_fdiv(h2, __builtin_amdgcn_rcpf((float)p.G)), r2 = (h2 - kvh2 * p.G) * p.nq + tok2;
    const size_t pb2 = ((size_t)b2 * p.nkv + kvh2) * nsp;
    f32x2_t ml2[S];
    u32x4 ov2[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const int sc = (S == 7) ? s : min(s, nsp - 1);
      ml2[s] = *reinterpret_cast<const f32x2_t*>(p.pml + ((pb2 + sc) * 32 + r2) * 2);
      ov2[s] = ld_global_16(p.po + ((pb2 + sc) * 32 + r2) * 128 + d2);
    }
    float Mx2 = -1.0e30f;
#pragma unroll
    for (int s = 0; s < S; ++s) Mx2 = fmaxf(Mx2, ml2[s][0]);
    float Ls2 = 0.f, v2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < S; ++s) {
      float f = ml2[s][1] * __builtin_amdgcn_exp2f(ml2[s][0] - Mx2);
      if (S != 7) f = s < nsp ? f : 0.f;
      Ls2 += f;
#pragma unroll
      for (int e = 0; e < 4; ++e) { v2[2 * e] += f * bf16lo_to_f32(ov2[s][e]); v2[2 * e + 1] += f * bf16hi_to_f32(ov2[s][e]); }
    }
    const float inv2 = Ls2 > 0.f ? 1.0f / Ls2 : 0.f;
    *reinterpret_cast<u32x4*>(xs + mm2 * XS + j2 * 16) = u32x4{pack_bf16x2(v2[0] * inv2, v2[1] * inv2), pack_bf16x2(v2[2] * inv2, v2[3] * inv2), pack_bf16x2(v2[4] * inv2, v2[5] * inv2), pack_bf16x2(v2[6] * inv2, v2[7] * inv2)};
  }
  __syncthreads();
  CH_STAMP(2);
  const bool mok = fr < M;
  const char* xrow = xs + (mok ? fr : 0) * XS + (wave * (NS * 32) + g * 8) * 2;
  f32x4 acc = {0, 0, 0, 0};
#pragma unroll
  for (int s = 0; s < NS; ++s) acc = mfma16(as_bf16x8(w[s]), as_bf16x8(*reinterpret_cast<const u32x4*>(xrow + s * 64)), acc);
  if constexpr (DBG) asm volatile("" ::"v"(acc[0]));
  CH_STAMP(3);
  if (wave != 0) *reinterpret_cast<f32x4*>(red + ((wave - 1) * 64 + lane) * 4) = acc;
  __syncthreads();
  CH_STAMP(4);
  if (wave == 0) {
#pragma unroll
    for (int w2 = 0; w2 < 7; ++w2) acc += *reinterpret_cast<const f32x4*>(red + (w2 * 64 + lane) * 4);
    if (mok) *reinterpret_cast<f32x4*>(p.out + ((size_t)ks * M + fr) * p.N + unit * 16 + g * 4) = acc;
  }
  CH_STAMP(5);
}

extern "C" int vlaser_chain_oproj_supported(int M, int N, int K, int k_splits, int attn_splits, int group) {
  if (M < 1 || M > 16 || N % 16 || K % (k_splits * 256) || K % (128 * group) || (K / k_splits) % (128) ) return 0;
  const int ns = K / (k_splits * 256);
  if (!((ns == 2 || ns == 3) && attn_splits >= 1 && attn_splits <= 16)) return 0;
  return 1;            // (M (K / k_splits / 8) > 512 pieces of the activation tile: further passes of the merge prologue)
}

/* args as vlaser_skinny(VL_PRO_ATTN, VL_SK_PARTIAL) with tiles_per_unit = 1, attn_m = the (m, l) pairs and attn_o = the bf16 rows written by vlaser_chain_attn */
extern "C" int vlaser_chain_oproj(const VlaserSkinnyArgs* a, vl_stream_t s) {
  VL_CHECK(a && a->W && a->attn_m && a->attn_o && a->out_f32, "vlaser_chain_oproj: null operand");
  VL_CHECK(a->tiles_per_unit == 1 && vlaser_chain_oproj_supported(a->M, a->N, a->K, a->k_splits, a->attn_splits, a->attn_group),
           "vlaser_chain_oproj: built for 16-row units, 2 / 3 K-steps per wave, <= 16 attention splits (got M %d N %d K %d ks %d splits %d)", a->M, a->N, a->K,
           a->k_splits, a->attn_splits);
  ChainOprojP p;
  p.pml = a->attn_m; p.po = (const bf16_t*)a->attn_o; p.W = (const u32x4*)a->W; p.out = a->out_f32;
  p.M = a->M; p.N = a->n_valid > 0 ? a->n_valid : a->N; p.n_units = a->N / 16; p.G = a->attn_group; p.nq = a->attn_nq; p.nkv = a->K / (128 * a->attn_group); p.n_splits = a->attn_splits;
  const int kb = a->K / a->k_splits, ns = kb / 256;
  p.inv_cpr = 8.0f / (float)kb; p.dbg = a->dbg;
  VL_CHECK(p.N % 16 == 0, "vlaser_chain_oproj: N must be a whole number of 16-column units");
  const int lds = ((a->M * (kb * 2 + 16) + 15) & ~15) + 7 * 64 * 16;
  const dim3 grid(a->N / 16, a->k_splits);
  hipStream_t stream = (hipStream_t)s;
#define CO_LAUNCH(NS_, S_, DBG_)                                                                          \
  do {                                                                                                    \
    if (int rc = set_max_lds_once(chain_oproj_kernel<NS_, S_, DBG_>, lds)) return rc;                      \
    hipLaunchKernelGGL((chain_oproj_kernel<NS_, S_, DBG_>), grid, dim3(512), lds, stream, p);              \
    VL_LAUNCH_CHECK();                                                                                    \
    return 0;                                                                                             \
  } while (0)
#define CO_CASE(NS_, S_) if (ns == NS_ && sreq == S_) { if (p.dbg) CO_LAUNCH(NS_, S_, true); else CO_LAUNCH(NS_, S_, false); }
  const int sreq = p.n_splits == 7 ? 7 : (p.n_splits <= 10 ? 10 : 16);
  CO_CASE(2, 7) CO_CASE(3, 7) CO_CASE(2, 10) CO_CASE(3, 10) CO_CASE(2, 16) CO_CASE(3, 16)
#undef CO_CASE
#undef CO_LAUNCH
  vlaser_set_error("vlaser_chain_oproj: no variant");
  return -1;
}
