// Fused launches of the <= 5-row Euler / decode layer-step for gfx950 (r03).
//
// One layer-step of the action expert is a chain of five all-to-all dependent GEMV-sized kernels (qkv -> attention -> o_proj -> gate/up -> down); at
// M = 4 each of them lasts 5-9 us of which only 1-3 us is its weight stream -- the rest is the kernel boundary and the prologue's dependent
// round trips.  The kernels in this file remove a boundary WITHOUT serialising the weight stream behind it:
//
//   vlaser_fused_ogu:   [attention-split merge + o_proj split-K partials]  --in-launch hand-off-->  [residual + RMSNorm + gate/up + SwiGLU]
//
// Every workgroup requests its share of the (27.5 MB) gate/up weights into REGISTERS at kernel start; the first `n_prod` workgroups first run the
// (2.4 MB) o_proj GEMV and publish their fp32 partial tiles write-through; all workgroups then wait for the producers' arrival counters, gather the
// slabs (12-36 KB, L2 / fabric) and finish.  The hop therefore overlaps the HBM stream instead of following it.
//
// Hand-off protocol (MI355X_MICROARCH.md "Workgroup dispatch, XCD placement & inter-workgroup visibility", recipe R1 in its counter form):
//   producer: payload by 16-byte sc1 (write-through) buffer stores from ONE wave -> `s_waitcnt vmcnt(0)` (inline asm) -> ONE relaxed agent-scope
//             atomic add on the arrival counter (4 counters on separate 128-byte lines, producer p uses counter p & 3);
//   consumer: ONE wave polls the 4 counters with relaxed agent-scope loads (+ s_sleep) -> __syncthreads() -> payload by sc1 buffer loads (L1 bypass;
//             the producer stored write-through, so no acquire fence is needed);
//   state:    the counters are zeroed by the CALLER on the stream before the launch (one memset per chunk over all launch slots: HIP-graph friendly);
//   liveness: producers never wait; they are the LOWEST block ids and workgroups are dispatched in id order, so a consumer can only ever wait for a
//             producer that is resident or already done -- and every spin is bounded (2 ms of wall clock): on expiry the workgroup raises the error
//             word (sync[ERR]) and finishes with whatever it has, so a protocol failure is a detectable wrong answer, never a hung GPU.
// Correctness does not depend on dispatch order, timing or workgroup -> XCD placement.
//
// Numerics: identical arithmetic, in the same order, as the two kernels it replaces (skinny_kernel<ATTN,PARTIAL> with 16-row units + skinny_kernel<NORM,SWIGLU>
// with 16-row lane-local units): tests/test_ops_gpu.py asserts bit equality.
#include "common.h"
#include "../../include/vlaser_hip_experimental.h"

#define EU_W 8
#define EU_T (EU_W * 64)
#define EU_CTR_STRIDE 32      // uint32 words between the arrival counters (128-byte lines)
#define EU_ERR (4 * EU_CTR_STRIDE)

typedef __attribute__((address_space(1))) unsigned int eu_gu32;
#define EU_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

struct FusedOguP {
  VlaserFusedOguArgs a;
  int n_units_o, n_prod;          // o_proj 16-row units (H / 16); producers = n_units_o * ks_o
  int kb_o;                       // o_proj K per producer
  float inv_cpr_o, inv_cpr_h;     // 8 / kb_o, 8 / H
  int attn_nkv;
  int n_units_gu;                 // gate/up 16-row units
  int up_lo, up_rem, uc_lo, uc_rem;   // gate/up units per producer / consumer workgroup (+1 for the first *_rem of each role)
  int xs_stride_o, xs_stride_h;   // LDS row strides in bytes
};

__device__ __forceinline__ int eu_fdiv(int x, float inv_d) { return (int)(((float)x + 0.5f) * inv_d); }
__device__ __forceinline__ unsigned long long eu_clock() { return wall_clock64(); }      // 100 MHz, constant rate
#define EU_STAMP(i) do { if (a.dbg && threadIdx.x == 0) a.dbg[(size_t)blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)

// ---- consumer half: wait for the producers, gather the o_proj slabs, residual + RMSNorm, gate/up on NU register-resident units, SwiGLU ------------------
// w[u][s]: fragments of unit (ustart + u), K-step s of this wave (u >= ucount: clamped duplicates, never stored).
template <int NSG, int KSO, int NU>
__device__ __forceinline__ void ogu_tail(const FusedOguP& p, char* smem, const u32x4 (&w)[NU][NSG], int ustart, int ucount, u32x4 hv, u32x4 wv) {
  const VlaserFusedOguArgs& a = p.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, g = lane >> 4;
  const int H = a.H, cpr = H >> 3, nch = a.M * cpr;
  char* xs = smem;                                                              // [M][xs_stride_h] normalised activations (bf16)
  const int xs_bytes = (a.M * p.xs_stride_h + 15) & ~15;
  float* red = reinterpret_cast<float*>(smem + xs_bytes);                       // [NU][EU_W][64][4] fp32
  float* gs = red;                                                              // group sums of the norm (before the MFMA partials need the space)
  // ---- 1. wait for every producer (ONE wave polls, relaxed; bounded)
  if (wave == 0) {
    eu_gu32* ctr = (eu_gu32*)(a.sync);
    const unsigned want = (unsigned)(p.n_prod >> 2);
    const unsigned long long t0 = eu_clock();
    bool all_ok;
    for (;;) {
      const unsigned v = lane < 4 ? __hip_atomic_load(ctr + lane * EU_CTR_STRIDE, EU_RLX_AGENT) : want;
      all_ok = __all(v >= want);
      if (all_ok) break;
      if (eu_clock() - t0 > 200000ull) break;                                    // 2 ms: give up, flag, finish with garbage (never hang the GPU)
      __builtin_amdgcn_s_sleep(2);
    }
    if (!all_ok && lane == 0) __hip_atomic_store(ctr + EU_ERR, 1u, EU_RLX_AGENT);
    EU_STAMP(3);
  }
  __syncthreads();
  // ---- 2. gather the KSO fp32 slabs of this thread's 8 columns (write-through by the producers -> sc1 loads, no fence) + residual -> bf16 h'
  const int ch = min(tid, nch - 1);
  const int mm = eu_fdiv(ch, p.inv_cpr_h), c = (ch - mm * cpr) << 3;
  const int off = mm * H + c;
  {
    auto rs = __builtin_amdgcn_make_buffer_rsrc(a.part_o, 0, (unsigned)(KSO * a.M * H * 4), 0x00020000);
    f32x4 q[2 * KSO];
#pragma unroll
    for (int u = 0; u < KSO; ++u) {
      const int bo = (u * a.M * H + off) * 4;
      q[2 * u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, bo, 0, 16));
      q[2 * u + 1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, bo + 16, 0, 16));
    }
    float v[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[2 * j] = bf16lo_to_f32(hv[j]); v[2 * j + 1] = bf16hi_to_f32(hv[j]); }
    float sl[8] = {0, 0, 0, 0, 0, 0, 0, 0};                                      // slabs first, residual last: the order of skinny_kernel<NORM,...>
#pragma unroll
    for (int u = 0; u < KSO; ++u) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { sl[j] += q[2 * u][j]; sl[4 + j] += q[2 * u + 1][j]; }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) hv[j] = pack_bf16x2(sl[2 * j] + v[2 * j], sl[2 * j + 1] + v[2 * j + 1]);
    if (a.dbg) { asm volatile("" ::"v"(hv[0])); EU_STAMP(4); }
  }
  // ---- 3. RMSNorm: sum of squares 16-lane group -> LDS -> fixed-order row total; every thread normalises its own 8 values
  const int ngr = cpr >> 4, gstride = (ngr + 3) & ~3;
  {
    float ssq = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { const float lo = bf16lo_to_f32(hv[j]), hi = bf16hi_to_f32(hv[j]); ssq += lo * lo + hi * hi; }
    if (tid >= nch) ssq = 0.f;
    ssq += __shfl_xor(ssq, 1, 64); ssq += __shfl_xor(ssq, 2, 64); ssq += __shfl_xor(ssq, 4, 64); ssq += __shfl_xor(ssq, 8, 64);
    if ((lane & 15) == 0 && tid < nch) gs[mm * gstride + ((tid - mm * cpr) >> 4)] = ssq;
    if (tid < nch && a.h_out != nullptr && blockIdx.x == 0) st_global_16(reinterpret_cast<bf16_t*>(a.h_out) + off, hv);
  }
  __syncthreads();
  {
    float tot = 0.f;
    const float* gr = gs + mm * gstride;
    for (int i = 0; i < ngr; i += 4) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(gr + i);
#pragma unroll
      for (int j = 0; j < 4; ++j) tot += (i + j < ngr) ? t[j] : 0.f;
    }
    const float rsq = rsqrtf(tot / (float)H + a.eps);
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float lo = round_bf16(bf16lo_to_f32(hv[j]) * rsq) * bf16lo_to_f32(wv[j]);
      const float hi = round_bf16(bf16hi_to_f32(hv[j]) * rsq) * bf16hi_to_f32(wv[j]);
      o[j] = pack_bf16x2(lo, hi);
    }
    if (tid < nch) *reinterpret_cast<u32x4*>(xs + mm * p.xs_stride_h + c * 2) = o;
  }
  __syncthreads();
  EU_STAMP(5);
  // ---- 4. gate/up: every wave its K slice of all NU units, partials through LDS, ONE barrier, wave u finishes unit u
  const char* xrow = xs + (fr < a.M ? fr : 0) * p.xs_stride_h + (wave * (NSG * 32) + g * 8) * 2;
  const bool mok = fr < a.M;
  f32x4 acc[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u) acc[u] = f32x4{0, 0, 0, 0};
#pragma unroll
  for (int s = 0; s < NSG; ++s) {
    u32x4 xv = {0, 0, 0, 0};
    if (mok) xv = *reinterpret_cast<const u32x4*>(xrow + s * 64);
    const bf16x8 xf = as_bf16x8(xv);
#pragma unroll
    for (int u = 0; u < NU; ++u) acc[u] = mfma16(as_bf16x8(w[u][s]), xf, acc[u]);
  }
#pragma unroll
  for (int u = 0; u < NU; ++u) *reinterpret_cast<f32x4*>(red + ((u * EU_W + wave) * 64 + lane) * 4) = acc[u];
  __syncthreads();
  if (wave < NU && wave < ucount) {
    const int u = wave;
    f32x4 r = *reinterpret_cast<const f32x4*>(red + ((u * EU_W + 0) * 64 + lane) * 4);
#pragma unroll
    for (int w2 = 1; w2 < EU_W; ++w2) r += *reinterpret_cast<const f32x4*>(red + ((u * EU_W + w2) * 64 + lane) * 4);   // wave 0 first, then 1 .. 7: the order of skinny_kernel
    const int unit = ustart + u, m = fr;
    if (m < a.M && unit * 16 < a.n_valid_gu) {
      bf16_t* o = reinterpret_cast<bf16_t*>(a.act) + (size_t)m * a.ld_act + unit * 8 + g * 2;
      const float r0 = round_bf16(silu(round_bf16(r[0]))) * round_bf16(r[2]);
      const float r1 = round_bf16(silu(round_bf16(r[1]))) * round_bf16(r[3]);
      *reinterpret_cast<uint32_t*>(o) = pack_bf16x2(r0, r1);
    }
  }
  EU_STAMP(6);
}

template <int NSO, int NSG, int SPL, int KSO, int NUP, int NUC>
__global__ __launch_bounds__(EU_T) void fused_ogu_kernel(FusedOguP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const VlaserFusedOguArgs& a = p.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, g = lane >> 4;
  const int bid = blockIdx.x;
  const int H = a.H, cpr = H >> 3, nch = a.M * cpr;
  EU_STAMP(0);
  // residual chunk + norm-weight chunk of this thread (both roles), requested first: small and L2 / fabric resident
  const int ch = min(tid, nch - 1);
  const int mm = eu_fdiv(ch, p.inv_cpr_h), c = (ch - mm * cpr) << 3;
  const u32x4 hv = ld_global_16(reinterpret_cast<const bf16_t*>(a.h_in) + mm * H + c);
  const u32x4 wv = ld_global_16(reinterpret_cast<const bf16_t*>(a.norm_w) + c);
  const u32x4* wgu = reinterpret_cast<const u32x4*>(a.Wgu) + (size_t)wave * (NSG * 64) + lane;      // [unit][wave][step][lane]
  constexpr size_t gu_unit_stride = (size_t)EU_W * NSG * 64;

  if (bid < p.n_prod) {
    // =========================================================================================== producer: attention merge + o_proj partial tile
    const int ks = bid / p.n_units_o, unit_o = bid - ks * p.n_units_o;
    const int ucount = p.up_lo + (bid < p.up_rem ? 1 : 0);
    const int ustart = bid * p.up_lo + min(bid, p.up_rem);
    const int kb0 = ks * p.kb_o;
    // ---- merge requests: flash-decoding partials (m, l, o) of this thread's 8 columns of the K slice, all SPL splits at once
    const int cpr_o = p.kb_o >> 3, nch_o = a.M * cpr_o;
    const int co = min(tid, nch_o - 1);
    const int mo = eu_fdiv(co, p.inv_cpr_o), jo = co - mo * cpr_o;
    const int k = kb0 + jo * 8, hh = k >> 7, d = k & 127;
    const int G = a.attn_group, nq = a.attn_nq;
    const int b = eu_fdiv(mo, __builtin_amdgcn_rcpf((float)nq)), tok = mo - b * nq, kvh = eu_fdiv(hh, __builtin_amdgcn_rcpf((float)G)), hg = hh - kvh * G,
              r = hg * nq + tok;
    const size_t pbase = ((size_t)b * p.attn_nkv + kvh) * SPL;
    float ms[SPL], ls[SPL];
    f32x4 o0[SPL], o1[SPL];
#pragma unroll
    for (int sp = 0; sp < SPL; ++sp) {
      ms[sp] = a.attn_m[(pbase + sp) * 32 + r];
      ls[sp] = a.attn_l[(pbase + sp) * 32 + r];
      const float* po = a.attn_o + ((pbase + sp) * 32 + r) * 128 + d;
      o0[sp] = *reinterpret_cast<const f32x4*>(po);
      o1[sp] = *reinterpret_cast<const f32x4*>(po + 4);
    }
    __builtin_amdgcn_sched_barrier(0);       // vmcnt retires in issue order: the weights go BEHIND the (small) partials
    // ---- o_proj weights of (ks, unit_o), this wave's K steps
    u32x4 wo[NSO];
    {
      const u32x4* src = reinterpret_cast<const u32x4*>(a.Wo) + (((size_t)ks * p.n_units_o + unit_o) * EU_W + wave) * (NSO * 64) + lane;
#pragma unroll
      for (int f = 0; f < NSO; ++f) wo[f] = __builtin_nontemporal_load(src + f * 64);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- merge -> LDS (bf16)
    char* xo = smem;                                                             // [M][xs_stride_o]
    {
      float Mx = -1.0e30f;
#pragma unroll
      for (int sp = 0; sp < SPL; ++sp) Mx = fmaxf(Mx, ms[sp]);
      float Ls = 0.f, v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int sp = 0; sp < SPL; ++sp) {
        const float f = __builtin_amdgcn_exp2f(ms[sp] - Mx);
        Ls += ls[sp] * f;
#pragma unroll
        for (int q = 0; q < 4; ++q) { v[q] += o0[sp][q] * f; v[4 + q] += o1[sp][q] * f; }
      }
      const float inv = Ls > 0.f ? 1.0f / Ls : 0.f;
      u32x4 xr;
#pragma unroll
      for (int q = 0; q < 4; ++q) xr[q] = pack_bf16x2(v[2 * q] * inv, v[2 * q + 1] * inv);
      if (tid < nch_o) *reinterpret_cast<u32x4*>(xo + mo * p.xs_stride_o + jo * 16) = xr;
    }
    __syncthreads();
    EU_STAMP(1);
    // ---- o_proj MFMA on this wave's K slice, partials through LDS
    const int xo_bytes = (a.M * p.xs_stride_o + 15) & ~15;
    float* redo = reinterpret_cast<float*>(smem + xo_bytes);                     // [EU_W - 1][64][4]
    {
      const char* xrow = xo + (fr < a.M ? fr : 0) * p.xs_stride_o + (wave * (NSO * 32) + g * 8) * 2;
      const bool mok = fr < a.M;
      f32x4 acc = {0, 0, 0, 0};
#pragma unroll
      for (int s = 0; s < NSO; ++s) {
        u32x4 xv = {0, 0, 0, 0};
        if (mok) xv = *reinterpret_cast<const u32x4*>(xrow + s * 64);
        acc = mfma16(as_bf16x8(wo[s]), as_bf16x8(xv), acc);
      }
      if (wave != 0) *reinterpret_cast<f32x4*>(redo + ((wave - 1) * 64 + lane) * 4) = acc;
      __syncthreads();
      if (wave == 0) {
#pragma unroll
        for (int w2 = 0; w2 < EU_W - 1; ++w2) acc += *reinterpret_cast<const f32x4*>(redo + (w2 * 64 + lane) * 4);
        // ---- publish: 16-byte write-through stores of the tile's rows, drain, ONE arrival
        auto rs = __builtin_amdgcn_make_buffer_rsrc(a.part_o, 0, (unsigned)(KSO * a.M * H * 4), 0x00020000);
        const int n0 = unit_o * 16 + g * 4;
        if (fr < a.M) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc), rs, ((ks * a.M + fr) * H + n0) * 4, 0, 16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add((eu_gu32*)(a.sync) + (bid & 3) * EU_CTR_STRIDE, 1u, EU_RLX_AGENT);
        EU_STAMP(2);
      }
    }
    // ---- gate/up weights of this workgroup's units, requested only NOW: vmcnt counts loads and stores in one in-order queue, so with these in flight
    // the publishing wave's `vmcnt(0)` (store acknowledged) would also have waited for ~100 KB of HBM stream.  Clamped, unconditional, straight-line.
    u32x4 w[NUP][NSG];
#pragma unroll
    for (int u = 0; u < NUP; ++u) {
      const u32x4* src = wgu + (size_t)(ustart + min(u, ucount - 1)) * gu_unit_stride;
#pragma unroll
      for (int f = 0; f < NSG; ++f) w[u][f] = __builtin_nontemporal_load(src + f * 64);
    }
    __syncthreads();                                                             // LDS is re-used by the tail
    ogu_tail<NSG, KSO, NUP>(p, smem, w, ustart, ucount, hv, wv);
  } else {
    // =========================================================================================== consumer only
    const int cb = bid - p.n_prod;
    const int ucount = p.uc_lo + (cb < p.uc_rem ? 1 : 0);
    const int ustart = p.n_prod * p.up_lo + p.up_rem + cb * p.uc_lo + min(cb, p.uc_rem);
    if (a.cons_delay > 0) {        // tuning knob: hold the bulk weight stream back so that the producers' small dependent requests do not queue behind 27 MB
      const unsigned long long t0 = eu_clock();
      while (eu_clock() - t0 < (unsigned long long)a.cons_delay) __builtin_amdgcn_s_sleep(1);
    }
    u32x4 w[NUC][NSG];
#pragma unroll
    for (int u = 0; u < NUC; ++u) {
      const u32x4* src = wgu + (size_t)(ustart + min(u, ucount - 1)) * gu_unit_stride;
#pragma unroll
      for (int f = 0; f < NSG; ++f) w[u][f] = __builtin_nontemporal_load(src + f * 64);
    }
    ogu_tail<NSG, KSO, NUC>(p, smem, w, ustart, ucount, hv, wv);
  }
}

extern "C" int vlaser_fused_ogu(const VlaserFusedOguArgs* a, vl_stream_t s) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(s);
  VL_CHECK(a && a->attn_m && a->attn_l && a->attn_o && a->Wo && a->part_o && a->h_in && a->norm_w && a->Wgu && a->act && a->sync, "vlaser_fused_ogu: null operand");
  VL_CHECK(a->M >= 1 && a->M * (a->H / 8) <= EU_T, "vlaser_fused_ogu: M * H / 8 = %d must fit one pass of %d threads", a->M * (a->H / 8), EU_T);
  VL_CHECK(a->H == 768 && a->K_o == 1536 && a->ks_o == 3 && a->attn_splits == 7, "vlaser_fused_ogu: built for the action expert's layer-step (H 768, 12 heads, 3 o_proj K slices, 7 key splits): got H %d K_o %d ks_o %d splits %d",
           a->H, a->K_o, a->ks_o, a->attn_splits);
  VL_CHECK(a->N_gu % 16 == 0 && (a->n_valid_gu <= 0 || a->n_valid_gu % 16 == 0), "vlaser_fused_ogu: gate/up rows must be whole 16-row units");
  VL_CHECK(a->attn_group >= 1 && a->attn_nq >= 1 && a->K_o % (128 * a->attn_group) == 0, "vlaser_fused_ogu: bad attention-merge arguments");
  FusedOguP p;
  p.a = *a;
  if (p.a.n_valid_gu <= 0) p.a.n_valid_gu = a->N_gu;
  p.n_units_o = a->H / 16;
  p.n_prod = p.n_units_o * a->ks_o;
  p.kb_o = a->K_o / a->ks_o;
  VL_CHECK(a->M * (p.kb_o / 8) <= EU_T && p.n_prod % 4 == 0 && p.n_prod < 256, "vlaser_fused_ogu: producer geometry");
  p.inv_cpr_o = 8.0f / (float)p.kb_o;
  p.inv_cpr_h = 8.0f / (float)a->H;
  p.attn_nkv = a->K_o / (128 * a->attn_group);
  p.n_units_gu = a->N_gu / 16;
  constexpr int NUP = 4, NUC = 5, GRID = 256;
  const int n_cons = GRID - p.n_prod;
  // consumers take NUC units each as far as they go, producers share the rest (they also stream the o_proj weights)
  int cons_units = n_cons * NUC;
  if (cons_units > p.n_units_gu) cons_units = p.n_units_gu;
  const int prod_units = p.n_units_gu - cons_units;
  p.up_lo = prod_units / p.n_prod; p.up_rem = prod_units % p.n_prod;
  p.uc_lo = cons_units / n_cons;   p.uc_rem = cons_units % n_cons;
  VL_CHECK(p.up_lo >= 1 && p.up_lo + (p.up_rem ? 1 : 0) <= NUP && p.uc_lo >= 1 && p.uc_lo + (p.uc_rem ? 1 : 0) <= NUC,
           "vlaser_fused_ogu: %d gate/up units do not fit %d producers x %d + %d consumers x %d register-resident units", p.n_units_gu, p.n_prod, NUP, n_cons, NUC);
  p.xs_stride_o = p.kb_o * 2 + 16;
  p.xs_stride_h = a->H * 2 + 16;
  const int lds_tail = ((a->M * p.xs_stride_h + 15) & ~15) + NUC * EU_W * 64 * 16;
  const int lds_prod = ((a->M * p.xs_stride_o + 15) & ~15) + (EU_W - 1) * 64 * 16;
  const int lds = lds_tail > lds_prod ? lds_tail : lds_prod;
  auto kern = fused_ogu_kernel<2, 3, 7, 3, NUP, NUC>;
  if (int rc = set_max_lds_once(kern, lds)) return rc;
  hipLaunchKernelGGL(kern, dim3(GRID), dim3(EU_T), lds, stream, p);
  VL_LAUNCH_CHECK();
  return 0;
}
