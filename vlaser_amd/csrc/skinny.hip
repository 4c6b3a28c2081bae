// Skinny (M <= 16 activation rows) weight-streaming GEMV on MFMA for gfx950 -- the HBM-bound regime of the 10
// action-expert Euler steps and of greedy decode: every weight byte is read exactly once per call.
//
// Geometry.  Block = 8 waves; a "unit" = TPU x 16 consecutive (packed) weight rows (TPU = 2 or 6 MFMA tiles); the 8
// waves split the block's K range (in-block split-K, reduced through LDS); blockIdx.y adds cross-block split-K for
// narrow outputs (o_proj / down_proj): fp32 partial slabs go to a workspace and are summed by the CONSUMER's
// prologue -- deterministic, no atomics, no extra launch.  The grid is sized to <= one block per CU: a block owns a
// balanced run of consecutive units and executes the (per-block redundant) prologue ONCE.
//
// Memory layout.  Weights are PRE-PACKED in HBM in MFMA-fragment order (vlaser_amd.ops.pack_skinny):
//     [k_split][unit][wave][k_step][tile][lane][8 bf16]
// so every wave-level load is one contiguous 1 KiB and a wave's whole stream is contiguous.  Measured on MI355X
// (tools/micro/pattern.hip): 27.5 MB stream, 256 blocks: contiguous 4.8 TB/s vs 3.3 TB/s for the row-major
// fragment pattern (16 rows x 64 B per instruction).  Fragments go global -> VGPR directly (guide: "GEMV / M <= 16
// decode weights: load straight to VGPRs"); activations (a few KB) sit in LDS.
//
// Latency engineering (a launch lasts only a few microseconds at M = 4, so every serialized round trip shows):
//   * NS, the number of 32-wide K-steps per wave, is a TEMPLATE parameter: every load of a unit is unconditional and
//     straight-line, so hipcc can count vmcnt exactly (a load behind a runtime condition makes it fall back to
//     vmcnt(0) at every use and serialises the whole stream -- guide 5 "three .s-level traps" (c));
//   * the (small, L2-resident) prologue requests -- residual chunk, norm weight, ALL split-K slabs of the chunk,
//     epilogue operands -- are issued first and the unit's weight fragments right behind them; vmcnt retires in
//     issue order, so the prologue's LDS phases run while the weights are still in flight;
//   * the next unit's fragments are requested before the current unit is consumed (last iteration peeled);
//   * one copy of K-loop / reduce / epilogue per peel: small code, a cold instruction cache is paid on every launch.
// MFMA operands are swapped (W as A, x as B) so each lane owns 4 consecutive outputs n of one row m, sharing the
// fused epilogues of the big GEMM (bias / SiLU / SwiGLU / RoPE + KV-cache scatter).
#include <type_traits>

#include "common.h"
#include "../../include/vlaser_hip.h"

#define SKW 8        // waves per block
#define SKT (SKW * 64)

struct SkinnyP {
  VlaserSkinnyArgs a;
  int xs_stride;  // bytes per LDS activation row
  int kb;         // K per block
  float inv_cpr;  // 8 / kb: chunk -> row without an integer division
  int ulo, urem;  // units per block = ulo (+1 for the first urem blocks): host-side, integer division is slow on the device
  int attn_nkv;   // ATTN prologue: K / (128 * attn_group)
};

// x / d for small non-negative ints through a float reciprocal (an integer division is ~40 VALU ops on gfx950)
__device__ __forceinline__ int fdiv(int x, float inv_d) { return (int)(((float)x + 0.5f) * inv_d); }

struct EpiOps {  // epilogue operands of one tile pair
  u32x2 b0, b1;  // 4 bf16 bias values of tile 0 / tile 1
  f32x4 cs, sn;  // RoPE cos / sin
  int pos;       // position id of the row (cache slot when slot_base < 0)
};

template <int EPI>
__device__ __forceinline__ void load_epi(const VlaserSkinnyArgs& a, int pair, int m, int g, EpiOps& e, int pos) {
  // unconditional (clamped) vector loads; pair = index of the 32-row group
  const int n0 = min(pair, (a.N >> 5) - 1) * 32 + g * 4;
  if constexpr (EPI == VL_SK_BIAS || EPI == VL_SK_BIAS_SILU || EPI == VL_SK_QKV_ROPE || EPI == VL_SK_F32) {
    const bf16_t* bias = reinterpret_cast<const bf16_t*>(a.bias);   // padded to N by the host wrapper (or null for F32)
    if (bias) {
      e.b0 = *reinterpret_cast<const u32x2*>(bias + n0);
      e.b1 = *reinterpret_cast<const u32x2*>(bias + n0 + 16);
    } else {
      e.b0 = u32x2{0, 0};
      e.b1 = u32x2{0, 0};
    }
  }
  if constexpr (EPI == VL_SK_QKV_ROPE) {
    const int pp = n0 & 127, d = ((pp >> 5) << 4) + (pp & 15);
    e.pos = pos;            // position id of row m, loaded at kernel start (see skinny_kernel): cos / sin no longer hang off a load issued behind the weights
    e.cs = *reinterpret_cast<const f32x4*>(a.rope_cos + (size_t)pos * 64 + d);
    e.sn = *reinterpret_cast<const f32x4*>(a.rope_sin + (size_t)pos * 64 + d);
  }
}

template <int EPI>
__device__ __forceinline__ void skinny_epilogue(const VlaserSkinnyArgs& a, int ks, int pair, int m, int g, f32x4 acc0, f32x4 acc1,
                                                const EpiOps& e) {
  // lane -> row m, columns n = pair*32 + t*16 + g*4 + j  (acc0: t = 0, acc1: t = 1); n_valid = un-padded N
  const int n0 = pair * 32 + g * 4;
  float b0[4] = {0, 0, 0, 0}, b1[4] = {0, 0, 0, 0};
  if constexpr (EPI == VL_SK_BIAS || EPI == VL_SK_BIAS_SILU || EPI == VL_SK_QKV_ROPE || EPI == VL_SK_F32) {
    b0[0] = bf16lo_to_f32(e.b0[0]); b0[1] = bf16hi_to_f32(e.b0[0]); b0[2] = bf16lo_to_f32(e.b0[1]); b0[3] = bf16hi_to_f32(e.b0[1]);
    b1[0] = bf16lo_to_f32(e.b1[0]); b1[1] = bf16hi_to_f32(e.b1[0]); b1[2] = bf16lo_to_f32(e.b1[1]); b1[3] = bf16hi_to_f32(e.b1[1]);
  }
  if constexpr (EPI == VL_SK_PARTIAL) {
    float* o = a.out_f32 + ((size_t)ks * a.M + m) * a.n_valid;
    if (n0 + 3 < a.n_valid) *reinterpret_cast<f32x4*>(o + n0) = acc0;
    if (n0 + 19 < a.n_valid) *reinterpret_cast<f32x4*>(o + n0 + 16) = acc1;
  } else if constexpr (EPI == VL_SK_F32) {
    float* o = a.out_f32 + (size_t)m * a.n_valid;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (n0 + j < a.n_valid) o[n0 + j] = acc0[j] + b0[j];
      if (n0 + 16 + j < a.n_valid) o[n0 + 16 + j] = acc1[j] + b1[j];
    }
  } else if constexpr (EPI == VL_SK_BIAS || EPI == VL_SK_BIAS_SILU) {
    bf16_t* o = reinterpret_cast<bf16_t*>(a.out) + (size_t)m * a.ldo;
    float r0[4], r1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      r0[j] = acc0[j] + b0[j];
      r1[j] = acc1[j] + b1[j];
      if constexpr (EPI == VL_SK_BIAS_SILU) { r0[j] = silu(round_bf16(r0[j])); r1[j] = silu(round_bf16(r1[j])); }
    }
    *reinterpret_cast<u32x2*>(o + n0) = u32x2{pack_bf16x2(r0[0], r0[1]), pack_bf16x2(r0[2], r0[3])};
    *reinterpret_cast<u32x2*>(o + n0 + 16) = u32x2{pack_bf16x2(r1[0], r1[1]), pack_bf16x2(r1[2], r1[3])};
  } else if constexpr (EPI == VL_SK_SWIGLU) {
    // pair = [gate16 | up16]; output columns pair*16 + g*4 + j
    if (pair * 32 >= a.n_valid) return;      // zero-padded tail
    bf16_t* o = reinterpret_cast<bf16_t*>(a.out) + (size_t)m * a.ldo + pair * 16 + g * 4;
    float r[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = round_bf16(silu(round_bf16(acc0[j]))) * round_bf16(acc1[j]);
    *reinterpret_cast<u32x2*>(o) = u32x2{pack_bf16x2(r[0], r[1]), pack_bf16x2(r[2], r[3])};
  } else if constexpr (EPI == VL_SK_QKV_ROPE) {
    // packed rows: n = head*128 + 32*j + 16*half + r ; acc0 = half 0 (d = 16*j + r), acc1 = half 1 (d + 64)
    const int head = n0 >> 7, pp = n0 & 127;
    const int d = ((pp >> 5) << 4) + (pp & 15);
    float x1[4], x2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      x1[j] = round_bf16(acc0[j] + b0[j]);
      x2[j] = round_bf16(acc1[j] + b1[j]);
    }
    const int b = fdiv(m, __builtin_amdgcn_rcpf((float)a.tok_per_batch));
    // slot_base < 0: the cache slot is the row's position id (device-resident), so a captured decode step can be replayed while the
    // sequence grows (uniform decode: one new token per sequence, slot == position)
    const int slot = a.slot_base >= 0 ? a.slot_base + (m - b * a.tok_per_batch) : e.pos;
    const int nq = a.n_q_heads, nkv = a.n_kv_heads;
    if (head < nq + nkv) {
      float o1[4], o2[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        o1[j] = x1[j] * e.cs[j] - x2[j] * e.sn[j];
        o2[j] = x2[j] * e.cs[j] + x1[j] * e.sn[j];
      }
      bf16_t* dst = head < nq ? reinterpret_cast<bf16_t*>(a.q_out) + (size_t)m * nq * 128 + head * 128
                              : reinterpret_cast<bf16_t*>(a.k_cache) + (((size_t)b * nkv + (head - nq)) * a.s_max + slot) * 128;
      *reinterpret_cast<u32x2*>(dst + d) = u32x2{pack_bf16x2(o1[0], o1[1]), pack_bf16x2(o1[2], o1[3])};
      *reinterpret_cast<u32x2*>(dst + d + 64) = u32x2{pack_bf16x2(o2[0], o2[1]), pack_bf16x2(o2[2], o2[3])};
    } else {
      bf16_t* vt = reinterpret_cast<bf16_t*>(a.vt_cache) + ((size_t)b * nkv + (head - nq - nkv)) * 128 * a.s_max + slot;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        vt[(size_t)(d + j) * a.s_max] = f32_to_bf16(x1[j]);
        vt[(size_t)(d + 64 + j) * a.s_max] = f32_to_bf16(x2[j]);
      }
    }
  }
}

// ---- 16-row units (TPU = 1) with lane-local epilogues (r03): a unit = ONE MFMA tile whose rows are packed so that lane group g (rows 4g .. 4g+3) holds
//   SWIGLU:   [gate 2g, gate 2g+1, up 2g, up 2g+1] of the unit's 8 activation columns  (ops.pack_gate_up8),
//   QKV_ROPE: [d, d+1, d+64, d+65], d = 8*(unit % 8) + 2g, of head unit / 8            (ops.head_perm16),
// i.e. the SwiGLU partner / the RoPE partner of every value sits in the same lane, exactly as with the 32-row units, at half the unit
// size: 1120 instead of 560 gate/up units over 256 workgroups (longest workgroup 80 instead of 96 rows), 128 instead of 64 q/k/v units.
struct EpiOps16 {
  u32x2 b;            // 4 bf16 bias values of the lane's rows
  float cs[2], sn[2]; // RoPE cos / sin of d, d+1
};

template <int EPI>
__device__ __forceinline__ void load_epi16(const VlaserSkinnyArgs& a, int unit, int g, EpiOps16& e, int pos) {
  const int u = min(unit, (a.N >> 4) - 1);
  if constexpr (EPI == VL_SK_QKV_ROPE) {
    e.b = *reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16_t*>(a.bias) + u * 16 + g * 4);
    const int d = ((u & 7) << 3) + 2 * g;
    const f32x2_t c = *reinterpret_cast<const f32x2_t*>(a.rope_cos + (size_t)pos * 64 + d);
    const f32x2_t s = *reinterpret_cast<const f32x2_t*>(a.rope_sin + (size_t)pos * 64 + d);
    e.cs[0] = c[0]; e.cs[1] = c[1]; e.sn[0] = s[0]; e.sn[1] = s[1];
  }
}

template <int EPI>
__device__ __forceinline__ void skinny_epilogue16(const VlaserSkinnyArgs& a, int ks, int unit, int m, int g, f32x4 acc, const EpiOps16& e, int pos) {
  if constexpr (EPI == VL_SK_PARTIAL) {
    const int n0 = unit * 16 + g * 4;
    if (n0 + 3 < a.n_valid) *reinterpret_cast<f32x4*>(a.out_f32 + ((size_t)ks * a.M + m) * a.n_valid + n0) = acc;
  } else if constexpr (EPI == VL_SK_SWIGLU) {
    if (unit * 16 >= a.n_valid) return;      // zero-padded tail
    bf16_t* o = reinterpret_cast<bf16_t*>(a.out) + (size_t)m * a.ldo + unit * 8 + g * 2;
    const float r0 = round_bf16(silu(round_bf16(acc[0]))) * round_bf16(acc[2]);
    const float r1 = round_bf16(silu(round_bf16(acc[1]))) * round_bf16(acc[3]);
    *reinterpret_cast<uint32_t*>(o) = pack_bf16x2(r0, r1);
  } else if constexpr (EPI == VL_SK_QKV_ROPE) {
    const int head = unit >> 3, d = ((unit & 7) << 3) + 2 * g;
    const float b0 = bf16lo_to_f32(e.b[0]), b1 = bf16hi_to_f32(e.b[0]), b2 = bf16lo_to_f32(e.b[1]), b3 = bf16hi_to_f32(e.b[1]);
    const float x1[2] = {round_bf16(acc[0] + b0), round_bf16(acc[1] + b1)}, x2[2] = {round_bf16(acc[2] + b2), round_bf16(acc[3] + b3)};
    const int b = fdiv(m, __builtin_amdgcn_rcpf((float)a.tok_per_batch));
    const int slot = a.slot_base >= 0 ? a.slot_base + (m - b * a.tok_per_batch) : pos;
    const int nq = a.n_q_heads, nkv = a.n_kv_heads;
    if (head < nq + nkv) {
      float o1[2], o2[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        o1[j] = x1[j] * e.cs[j] - x2[j] * e.sn[j];
        o2[j] = x2[j] * e.cs[j] + x1[j] * e.sn[j];
      }
      bf16_t* dst = head < nq ? reinterpret_cast<bf16_t*>(a.q_out) + (size_t)m * nq * 128 + head * 128
                              : reinterpret_cast<bf16_t*>(a.k_cache) + (((size_t)b * nkv + (head - nq)) * a.s_max + slot) * 128;
      *reinterpret_cast<uint32_t*>(dst + d) = pack_bf16x2(o1[0], o1[1]);
      *reinterpret_cast<uint32_t*>(dst + d + 64) = pack_bf16x2(o2[0], o2[1]);
    } else {
      bf16_t* vt = reinterpret_cast<bf16_t*>(a.vt_cache) + ((size_t)b * nkv + (head - nq - nkv)) * 128 * a.s_max + slot;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        vt[(size_t)(d + j) * a.s_max] = f32_to_bf16(x1[j]);
        vt[(size_t)(d + 64 + j) * a.s_max] = f32_to_bf16(x2[j]);
      }
    }
  }
}

// SP > 0 (ATTN prologue): exact attention-split count.  SP >= 0 (NORM prologue): the split-K slab count is a compile-time constant -> exactly 2 + 2*SP loads per chunk and no
// clamped dummy loads / selects.  The prologue is instruction-issue bound (8 waves share 4 SIMDs, ~8 cycles per VALU op
// per wave), so the generic runtime-count path (SP = -1) costs ~2 us more per launch at SP = 5.
// NCH > 1 (chunked K): a wave's K range is NCH chunks of NS steps; the pipelined item is a (unit, chunk) pair and the accumulators
// live across the chunks of a unit -- hidden sizes whose K / (k_splits * 256) exceeds the 8 steps a wave can hold in registers at
// once (Vlaser-8B: 3584 = 14 x 256 -> 2 chunks of 7; its MLP width 18944 is zero-padded to 20480 = 5 splits x 2 chunks of 8).
template <int PRO, int EPI, int TPU, int NS, int SP = -1, int NCH = 1>
__global__ __launch_bounds__(SKT) void skinny_kernel(SkinnyP p) {
  constexpr int RPU = TPU * 16;                 // rows per unit
  constexpr int NF = TPU * NS;                  // fragments (16-byte loads) per lane per pipelined item (unit, or chunk of a unit)
  constexpr int NST = NS * NCH;                 // K-steps per wave per unit
  constexpr bool NEED_EPI = (EPI == VL_SK_BIAS || EPI == VL_SK_BIAS_SILU || EPI == VL_SK_QKV_ROPE || EPI == VL_SK_F32);
  // EARLY (gate/up: 2-3 units per block): units 1 and 2 are requested as soon as the prologue's own loads are back
  // (first barrier) instead of one unit ahead of the MFMAs, so the HBM stream does not restart after the ~3 us prologue
  constexpr bool EARLY = (PRO == VL_PRO_NORM && EPI == VL_SK_SWIGLU && NCH == 1);
  constexpr int UE = TPU == 1 ? 5 : 3;          // EARLY: units held in registers at once (16-row units: 1120 / 256 -> 4 or 5 per workgroup)
  constexpr bool EPI16 = (TPU == 1 && EPI == VL_SK_QKV_ROPE);
  extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef VL_KERNARG_UP_FRONT
  vl_kernargs_up_front(p);
#endif
  const VlaserSkinnyArgs& a = p.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, g = lane >> 4;
  const int ks = blockIdx.y;
  unsigned long long* dbg = a.dbg ? a.dbg + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 : nullptr;
#define STAMP(i) do { if (dbg && tid == 0) dbg[i] = wall_clock64(); } while (0)
  STAMP(0);
  // balanced run of units for this block
  const int n_units = a.N / RPU;
  const int ulo = p.ulo, urem = p.urem;
  const int ucount = ulo + ((int)blockIdx.x < urem ? 1 : 0);
  const int ustart = (int)blockIdx.x * ulo + min((int)blockIdx.x, urem);
  const int kb0 = ks * p.kb;                  // first k of this block
  char* xs = smem;                            // [M][xs_stride] activations (bf16)
  const int xs_bytes = (a.M * p.xs_stride + 15) & ~15;
  float* red = reinterpret_cast<float*>(smem + xs_bytes);                 // [2][SKW-1][TPU][64][4] fp32 (double-buffered per unit)
  char* wn_lds = smem + xs_bytes + 2 * (SKW - 1) * TPU * 64 * 16;         // [K] bf16 norm weight (NORM prologue)

  // ------------------------------------------------------------------ weight stream (fragment-major packed weights)
  const int kl0 = wave * (NST * 32);          // block-local k start of this wave
  // 16-byte index of (ks, unit u, wave, step s, tile t, lane) = ((((ks*n_units + u)*SKW + wave)*NST + s)*TPU + t)*64 + lane
  const u32x4* wp = reinterpret_cast<const u32x4*>(a.W) + ((size_t)ks * n_units * SKW + wave) * (NF * NCH * 64) + lane;
  constexpr size_t unit_stride = (size_t)SKW * NF * NCH * 64;
  const int m = fr;
  // RoPE position of this lane's row, requested before anything else: the cos / sin loads depend on it, and a dependent load issued
  // behind the unit's weight fragments made hipcc drain the whole weight stream (vmcnt(0)) in the middle of the prologue
  int pos_m = 0;
  if constexpr (EPI == VL_SK_QKV_ROPE) pos_m = a.pos_ids[min(m, a.M - 1)];
  u32x4 cw[NF], nw[NF], ew[EARLY ? UE - 2 : 1][EARLY ? NF : 1];  // current / next unit (/ units 2 .. UE-1, EARLY)
  EpiOps ce[NEED_EPI && TPU > 1 ? TPU / 2 : 1], ne[NEED_EPI && TPU > 1 ? TPU / 2 : 1], te[1];
  EpiOps16 ce16, ne16;
  auto load_unit = [&](int ui, u32x4* dst, EpiOps* e, EpiOps16* e16 = nullptr) {
    const u32x4* src = wp + (size_t)(ustart + ui) * unit_stride;
#pragma unroll
    for (int f = 0; f < NF; ++f) dst[f] = __builtin_nontemporal_load(src + f * 64);
    if constexpr (EPI16) {
      load_epi16<EPI>(a, ustart + ui, g, *e16, pos_m);
    } else if constexpr (NEED_EPI) {
#pragma unroll
      for (int pr = 0; pr < TPU / 2; ++pr) load_epi<EPI>(a, (ustart + ui) * (TPU / 2) + pr, m, g, e[pr], pos_m);
    }
  };

  // ------------------------------------------------------------------ prologue: activations -> LDS (bf16)
  // fast path: one 16-byte chunk per thread, all loads unconditional (index clamped); extra chunks (M*K/8 > 512) loop
  if constexpr (PRO == VL_PRO_PLAIN) {
    const bf16_t* X = reinterpret_cast<const bf16_t*>(a.x);
    const int cpr = p.kb >> 3, nch = a.M * cpr;
    const int c0 = min(tid, nch - 1);
    const int r0 = fdiv(c0, p.inv_cpr), j0 = c0 - r0 * cpr;
    const u32x4 x0 = ld_global_16(X + (size_t)r0 * a.K + kb0 + j0 * 8);
    // second chunk (M*K/8 in (512, 1024]: the expert's down projection has 640-800) requested up front as well, clamped instead of
    // conditional: behind the weight fragments a conditional load is waited for with vmcnt(0), i.e. the prologue drained unit 0's weights
    const int c1 = min(tid + SKT, nch - 1);
    const int r1 = fdiv(c1, p.inv_cpr), j1 = c1 - r1 * cpr;
    const u32x4 x1 = ld_global_16(X + (size_t)r1 * a.K + kb0 + j1 * 8);
    load_unit(0, cw, ce, &ce16);
    if (tid < nch) *reinterpret_cast<u32x4*>(xs + r0 * p.xs_stride + j0 * 16) = x0;
    if (tid + SKT < nch) *reinterpret_cast<u32x4*>(xs + r1 * p.xs_stride + j1 * 16) = x1;
    for (int c = tid + 2 * SKT; c < nch; c += SKT)
      *reinterpret_cast<u32x4*>(xs + (c / cpr) * p.xs_stride + (c % cpr) * 16) = ld_global_16(X + (size_t)(c / cpr) * a.K + kb0 + (c % cpr) * 8);
  } else if constexpr (PRO == VL_PRO_ATTN) {
    // x[m][k] (k = h*128 + d) = flash-decoding merge of the attention partials of vlaser_attn_skinny
    const int cpr = p.kb >> 3, nch = a.M * cpr, G = a.attn_group, nq = a.attn_nq, S = a.attn_splits;
    const int nkv = p.attn_nkv;
    auto attn_pass = [&](int cbase, auto issue_tag) {
      const int c = min(cbase + tid, nch - 1);
      const int mm = fdiv(c, p.inv_cpr), j = c - mm * cpr;
      const int k = kb0 + j * 8, h = k >> 7, d = k & 127;
      const int b = fdiv(mm, __builtin_amdgcn_rcpf((float)nq)), tok = mm - b * nq, kvh = fdiv(h, __builtin_amdgcn_rcpf((float)G)), hg = h - kvh * G,
                r = hg * nq + tok;
      const size_t pbase = ((size_t)b * nkv + kvh) * S;
      constexpr int NSPL = SP > 0 ? SP : 8;      // SP > 0: exact split count (no clamped dummy loads)
      float ms[NSPL], ls[NSPL];
      f32x4 o0[NSPL], o1[NSPL];
#pragma unroll
      for (int sp = 0; sp < NSPL; ++sp) {
        const int sc = SP > 0 ? sp : min(sp, S - 1);
        ms[sp] = a.attn_m[(pbase + sc) * 32 + r];
        ls[sp] = a.attn_l[(pbase + sc) * 32 + r];
        const float* po = a.attn_o + ((pbase + sc) * 32 + r) * 128 + d;
        o0[sp] = *reinterpret_cast<const f32x4*>(po);
        o1[sp] = *reinterpret_cast<const f32x4*>(po + 4);
      }
      if constexpr (decltype(issue_tag)::value) {
        // vmcnt retires in issue order: the weight fragments must be requested BEHIND the (L2-resident) attention partials, or every
        // counted wait on a partial also waits for HBM.  hipcc hoisted these two loads to the top of the kernel (their addresses are
        // ready first) until the order was pinned.
        __builtin_amdgcn_sched_barrier(0);
        load_unit(0, cw, ce, &ce16);
        __builtin_amdgcn_sched_barrier(0);
      }
      float Mx = -1.0e30f;
#pragma unroll
      for (int sp = 0; sp < NSPL; ++sp) Mx = fmaxf(Mx, ms[sp]);
      float Ls = 0.f, v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int sp = 0; sp < NSPL; ++sp) {
        const float f = (SP > 0 || sp < S) ? __builtin_amdgcn_exp2f(ms[sp] - Mx) : 0.f;
        Ls += ls[sp] * f;
#pragma unroll
        for (int q = 0; q < 4; ++q) { v[q] += o0[sp][q] * f; v[4 + q] += o1[sp][q] * f; }
      }
      const float inv = Ls > 0.f ? 1.0f / Ls : 0.f;
      u32x4 xr;
#pragma unroll
      for (int q = 0; q < 4; ++q) xr[q] = pack_bf16x2(v[2 * q] * inv, v[2 * q + 1] * inv);
      if (cbase + tid < nch) *reinterpret_cast<u32x4*>(xs + mm * p.xs_stride + j * 16) = xr;
    };
    attn_pass(0, std::true_type{});
    for (int cbase = SKT; cbase < nch; cbase += SKT) attn_pass(cbase, std::false_type{});
  } else {
    // h = bf16(h_in + sum partials); xn = bf16(w * bf16(h * rsqrt(mean(h^2) + eps)))
    // Phase 1 (all threads, one 16-byte chunk each per pass): residual chunk + norm-weight chunk + the first 8 slabs of
    // the chunk are requested together, the weight stream right behind them; the rounded residual goes to LDS (and to
    // h_out from block 0).   Phase 2 (one wave per row, LDS only): sum of squares, normalise in place.
    const bf16_t* Hin = reinterpret_cast<const bf16_t*>(a.x);
    const bf16_t* Wn = reinterpret_cast<const bf16_t*>(a.norm_w);
    const bool write_h = (a.h_out != nullptr) && blockIdx.x == 0 && ks == 0;
    const int cpr = a.K >> 3, nch = a.M * cpr;
    const size_t slab = (size_t)a.M * a.K;
    const int S = a.n_partials;
    // One-pass rows (M * K / 8 <= 512 chunks, rows aligned to 16-lane groups): each thread keeps its 8 rounded residual values and its
    // norm-weight chunk in registers, the sum of squares goes 16-lane group -> LDS -> fixed-order row total, and the thread normalises
    // its own values -- phase 2 then costs one short LDS round trip for all 8 waves instead of "one wave per row" walking the row
    // through LDS twice (1.2 us of the 3.4 us prologue in the in-kernel timeline, tools/micro/skinny_timeline.py).
    const bool fast2 = SP >= 0 && nch <= SKT && (cpr & 15) == 0 && (cpr >> 4) <= 12;
    u32x4 hr0 = {0, 0, 0, 0}, wv0 = {0, 0, 0, 0};
    int mm0 = 0, c0 = 0;
    float* gs = red;                              // group sums [M][gstride] (the unit-reduce buffer is idle during the prologue)
    const int ngr = cpr >> 4, gstride = (ngr + 3) & ~3;
    if constexpr (SP >= 0) {
      auto norm_pass_exact = [&](int cbase, auto issue_tag) {
        const int ch = min(cbase + tid, nch - 1);
        const int mm = fdiv(ch, p.inv_cpr), c = (ch - mm * cpr) << 3;
        const int off = mm * a.K + c, slab32 = a.M * a.K;
        const u32x4 hv = ld_global_16(Hin + off);
        const u32x4 wv = ld_global_16(Wn + c);
        f32x4 q[SP > 0 ? 2 * SP : 1];
#pragma unroll
        for (int u = 0; u < SP; ++u) {
          const float* pp = a.partials + (off + u * slab32);
          q[2 * u] = *reinterpret_cast<const f32x4*>(pp);
          q[2 * u + 1] = *reinterpret_cast<const f32x4*>(pp + 4);
        }
        if constexpr (decltype(issue_tag)::value) load_unit(0, cw, ce, &ce16);       // weight stream right behind the prologue requests
        float v[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[2 * j] = bf16lo_to_f32(hv[j]); v[2 * j + 1] = bf16hi_to_f32(hv[j]); }
        float sl[8] = {0, 0, 0, 0, 0, 0, 0, 0};                               // slabs first, residual last: same order as the generic path
#pragma unroll
        for (int u = 0; u < SP; ++u) {
#pragma unroll
          for (int j = 0; j < 4; ++j) { sl[j] += q[2 * u][j]; sl[4 + j] += q[2 * u + 1][j]; }
        }
        u32x4 hr;
#pragma unroll
        for (int j = 0; j < 4; ++j) hr[j] = pack_bf16x2(sl[2 * j] + v[2 * j], sl[2 * j + 1] + v[2 * j + 1]);
        if (fast2) {
          hr0 = hr; wv0 = wv; mm0 = mm; c0 = c;
          float ssq = 0.f;
#pragma unroll
          for (int j = 0; j < 4; ++j) { const float lo = bf16lo_to_f32(hr[j]), hi = bf16hi_to_f32(hr[j]); ssq += lo * lo + hi * hi; }
          if (tid >= nch) ssq = 0.f;               // clamped duplicate chunk
          ssq = group16_sum(ssq);
          if ((lane & 15) == 0 && tid < nch) gs[mm * gstride + ((tid - mm * cpr) >> 4)] = ssq;
          if (tid < nch && write_h) st_global_16(reinterpret_cast<bf16_t*>(a.h_out) + off, hr);
          // pin the norm-weight chunk's vmcnt wait HERE: its first real use is after the (conditional) early weight requests below, where
          // hipcc can no longer count and would drain the whole weight stream (vmcnt(0)) before phase 2
          asm volatile("" ::"v"(wv0[0]), "v"(wv0[1]), "v"(wv0[2]), "v"(wv0[3]));
        } else if (cbase + tid < nch) {
          *reinterpret_cast<u32x4*>(xs + mm * p.xs_stride + c * 2) = hr;
          if (mm == 0) *reinterpret_cast<u32x4*>(wn_lds + c * 2) = wv;
          if (write_h) st_global_16(reinterpret_cast<bf16_t*>(a.h_out) + off, hr);
        }
      };
      norm_pass_exact(0, std::true_type{});
      for (int cbase = SKT; cbase < nch; cbase += SKT) norm_pass_exact(cbase, std::false_type{});
    } else {
    // one pass = one 16-byte chunk per thread: residual + norm weight + slabs requested together (clamped, unconditional);
    // the rounded residual goes to LDS (and to h_out from block 0)
    auto norm_pass = [&](int cbase, auto issue_tag) {
      const int ch = min(cbase + tid, nch - 1);
      const int mm = ch / cpr, c = (ch - mm * cpr) << 3;
      const size_t off = (size_t)mm * a.K + c;
      const u32x4 hv = ld_global_16(Hin + off);
      const u32x4 wv = ld_global_16(Wn + c);
      float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      if (S <= 4) {
        f32x4 q[8];
        const float* pbase = S > 0 ? a.partials + off : reinterpret_cast<const float*>(Hin);   // any valid address when S == 0
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float* pp = pbase + (size_t)(S > 0 ? min(u, S - 1) : 0) * slab;
          q[2 * u] = *reinterpret_cast<const f32x4*>(pp);
          q[2 * u + 1] = *reinterpret_cast<const f32x4*>(pp + 4);
        }
        if constexpr (decltype(issue_tag)::value) load_unit(0, cw, ce, &ce16);       // weight stream right behind the prologue requests
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const bool on = u < S;     // select, not multiply: the clamped dummy loads may hold non-finite bit patterns
#pragma unroll
          for (int j = 0; j < 4; ++j) { v[j] += on ? q[2 * u][j] : 0.f; v[4 + j] += on ? q[2 * u + 1][j] : 0.f; }
        }
      } else {
        f32x4 q[16];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const float* pp = a.partials + off + (size_t)min(u, S - 1) * slab;
          q[2 * u] = *reinterpret_cast<const f32x4*>(pp);
          q[2 * u + 1] = *reinterpret_cast<const f32x4*>(pp + 4);
        }
        if constexpr (decltype(issue_tag)::value) load_unit(0, cw, ce, &ce16);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const bool on = u < S;
#pragma unroll
          for (int j = 0; j < 4; ++j) { v[j] += on ? q[2 * u][j] : 0.f; v[4 + j] += on ? q[2 * u + 1][j] : 0.f; }
        }
        for (int Sr = S - 8; Sr > 0; Sr -= 8) add_slabs_clamped<8>(v, a.partials + off + (size_t)(S - Sr) * slab, slab, min(Sr, 8));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[2 * j] += bf16lo_to_f32(hv[j]); v[2 * j + 1] += bf16hi_to_f32(hv[j]); }
      u32x4 hr;
#pragma unroll
      for (int j = 0; j < 4; ++j) hr[j] = pack_bf16x2(v[2 * j], v[2 * j + 1]);
      if (cbase + tid < nch) {
        *reinterpret_cast<u32x4*>(xs + mm * p.xs_stride + c * 2) = hr;
        if (mm == 0) *reinterpret_cast<u32x4*>(wn_lds + c * 2) = wv;
        if (write_h) st_global_16(reinterpret_cast<bf16_t*>(a.h_out) + off, hr);
      }
    };
    norm_pass(0, std::true_type{});
    for (int cbase = SKT; cbase < nch; cbase += SKT) norm_pass(cbase, std::false_type{});
    }  // SP < 0
    __syncthreads();
    STAMP(1);
    if constexpr (EARLY) {                     // block-uniform conditions; hipcc then waits for all of them at the first use -- and, read in the ISA in r05, puts an
      if (ucount > 1) load_unit(1, nw, ne);    // `s_waitcnt vmcnt(0)` in FRONT of each of these blocks: the units stream one after the other.  csrc/chain.hip's
#pragma unroll                                 // chain_gu requests everything up front and replaces this path wherever it has a variant
      for (int u = 2; u < UE; ++u)
        if (ucount > u) load_unit(u, ew[u - 2], te);
    }
    if (fast2) {
      float tot = 0.f;
      const float* gr = gs + mm0 * gstride;
      for (int i = 0; i < ngr; i += 4) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(gr + i);
#pragma unroll
        for (int j = 0; j < 4; ++j) tot += (i + j < ngr) ? t[j] : 0.f;      // fixed order: deterministic
      }
      const float rs = rsqrtf(tot / (float)a.K + a.eps);
      u32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float lo = round_bf16(bf16lo_to_f32(hr0[j]) * rs) * bf16lo_to_f32(wv0[j]);
        const float hi = round_bf16(bf16hi_to_f32(hr0[j]) * rs) * bf16hi_to_f32(wv0[j]);
        o[j] = pack_bf16x2(lo, hi);
      }
      if (tid < nch) *reinterpret_cast<u32x4*>(xs + mm0 * p.xs_stride + c0 * 2) = o;
    } else
    // phase 2 (one wave per row, LDS only): sum of squares, normalise in place
    for (int mm = wave; mm < a.M; mm += SKW) {
      float ssq = 0.f;
      for (int c = lane * 8; c < a.K; c += 512) {
        const u32x4 hr = *reinterpret_cast<const u32x4*>(xs + mm * p.xs_stride + c * 2);
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float lo = bf16lo_to_f32(hr[j]), hi = bf16hi_to_f32(hr[j]); ssq += lo * lo + hi * hi; }
      }
      ssq = wave_sum(ssq);
      const float rs = rsqrtf(ssq / (float)a.K + a.eps);
      for (int c = lane * 8; c < a.K; c += 512) {
        const u32x4 hr = *reinterpret_cast<const u32x4*>(xs + mm * p.xs_stride + c * 2);
        const u32x4 wv = *reinterpret_cast<const u32x4*>(wn_lds + c * 2);
        u32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float lo = round_bf16(bf16lo_to_f32(hr[j]) * rs) * bf16lo_to_f32(wv[j]);
          const float hi = round_bf16(bf16hi_to_f32(hr[j]) * rs) * bf16hi_to_f32(wv[j]);
          o[j] = pack_bf16x2(lo, hi);
        }
        *reinterpret_cast<u32x4*>(xs + mm * p.xs_stride + c * 2) = o;
      }
    }
  }
  __syncthreads();
  STAMP(2);

  // ------------------------------------------------------------------ main loop (last unit peeled: no dangling prefetch)
  const char* xrow = xs + (fr < a.M ? fr : 0) * p.xs_stride + (kl0 + g * 8) * 2;
  const bool mok = fr < a.M;
  int par = 0;
  if constexpr (NCH > 1) {
    f32x4 acc[TPU];
#pragma unroll
    for (int t = 0; t < TPU; ++t) acc[t] = f32x4{0, 0, 0, 0};
    auto load_item = [&](int ui, int c, u32x4* dst, EpiOps* e) {
      const u32x4* src = wp + (size_t)(ustart + ui) * unit_stride + (size_t)c * (NF * 64);
#pragma unroll
      for (int f = 0; f < NF; ++f) dst[f] = __builtin_nontemporal_load(src + f * 64);
      if constexpr (NEED_EPI) {
#pragma unroll
        for (int pr = 0; pr < TPU / 2; ++pr) load_epi<EPI>(a, (ustart + ui) * (TPU / 2) + pr, m, g, e[pr], pos_m);
      }
    };
    auto mma = [&](int c, const u32x4* w) {
#pragma unroll
      for (int u = 0; u < NS; ++u) {
        u32x4 xv = {0, 0, 0, 0};
        if (mok) xv = *reinterpret_cast<const u32x4*>(xrow + (c * NS + u) * 64);
        const bf16x8 xf = as_bf16x8(xv);
#pragma unroll
        for (int t = 0; t < TPU; ++t) acc[t] = mfma16(as_bf16x8(w[TPU * u + t]), xf, acc[t]);
      }
    };
    auto finish = [&](int ui, const EpiOps* e) {          // in-block split-K reduce + epilogue of one unit (as consume_finish below)
      float* rb = red + par * (SKW - 1) * TPU * 64 * 4;
      if (wave != 0) {
        float* r = rb + ((wave - 1) * TPU * 64 + lane) * 4;
#pragma unroll
        for (int t = 0; t < TPU; ++t) *reinterpret_cast<f32x4*>(r + t * 64 * 4) = acc[t];
      }
      __syncthreads();
      par ^= 1;
      if (wave == 0) {
#pragma unroll
        for (int w2 = 0; w2 < SKW - 1; ++w2) {
          const float* r = rb + (w2 * TPU * 64 + lane) * 4;
#pragma unroll
          for (int t = 0; t < TPU; ++t) acc[t] += *reinterpret_cast<const f32x4*>(r + t * 64 * 4);
        }
        if (m < a.M) {
#pragma unroll
          for (int pr = 0; pr < TPU / 2; ++pr)
            skinny_epilogue<EPI>(a, ks, (ustart + ui) * (TPU / 2) + pr, m, g, acc[2 * pr], acc[2 * pr + 1], e[NEED_EPI ? pr : 0]);
        }
      }
#pragma unroll
      for (int t = 0; t < TPU; ++t) acc[t] = f32x4{0, 0, 0, 0};
    };
    // item 0 was requested by the prologue (load_unit(0) = chunk 0 of unit 0: the first NF fragments of the unit)
    const int n_items = ucount * NCH;
    int ui = 0, c = 0;
    for (int it = 0; it + 1 < n_items; ++it) {             // item it+1 in flight while item it is consumed; last item peeled
      int c2 = c + 1, ui2 = ui;
      if (c2 == NCH) { c2 = 0; ++ui2; }
      load_item(ui2, c2, nw, ne);
      mma(c, cw);
      if (c == NCH - 1) finish(ui, ce);
#pragma unroll
      for (int f = 0; f < NF; ++f) cw[f] = nw[f];
      if constexpr (NEED_EPI) {
#pragma unroll
        for (int pr = 0; pr < TPU / 2; ++pr) ce[pr] = ne[pr];
      }
      ui = ui2; c = c2;
    }
    mma(c, cw);
    finish(ui, ce);
    return;
  }
  auto consume_finish = [&](int ui, const u32x4* w, const EpiOps* e, const EpiOps16* e16 = nullptr) {
    f32x4 acc[TPU];
#pragma unroll
    for (int t = 0; t < TPU; ++t) acc[t] = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      u32x4 xv = {0, 0, 0, 0};
      if (mok) xv = *reinterpret_cast<const u32x4*>(xrow + u * 64);
      const bf16x8 xf = as_bf16x8(xv);
#pragma unroll
      for (int t = 0; t < TPU; ++t) acc[t] = mfma16(as_bf16x8(w[TPU * u + t]), xf, acc[t]);
    }
    if (ui == 0) { asm volatile("" ::"v"(acc[0][0])); STAMP(3); }
    // in-block split-K reduce + epilogue; `red` is double-buffered by unit parity
    float* rb = red + par * (SKW - 1) * TPU * 64 * 4;
    if (wave != 0) {
      float* r = rb + ((wave - 1) * TPU * 64 + lane) * 4;
#pragma unroll
      for (int t = 0; t < TPU; ++t) *reinterpret_cast<f32x4*>(r + t * 64 * 4) = acc[t];
    }
    __syncthreads();
    if (ui == 0) STAMP(4);
    par ^= 1;
    if (wave == 0) {
#pragma unroll
      for (int w2 = 0; w2 < SKW - 1; ++w2) {
        const float* r = rb + (w2 * TPU * 64 + lane) * 4;
#pragma unroll
        for (int t = 0; t < TPU; ++t) acc[t] += *reinterpret_cast<const f32x4*>(r + t * 64 * 4);
      }
      if (m < a.M) {
        if constexpr (TPU == 1) {          // 16-row units: one tile, lane-local epilogue (partial slab / SwiGLU / bias + RoPE + cache scatter)
          EpiOps16 e0 = {};
          skinny_epilogue16<EPI>(a, ks, ustart + ui, m, g, acc[0], EPI16 ? *e16 : e0, pos_m);
        } else {
#pragma unroll
          for (int pr = 0; pr < TPU / 2; ++pr)
            skinny_epilogue<EPI>(a, ks, (ustart + ui) * (TPU / 2) + pr, m, g, acc[2 * pr], acc[2 * pr + 1], e[NEED_EPI ? pr : 0]);
        }
      }
    }
  };
  int ui0 = 0;
  if constexpr (EARLY) {
    consume_finish(0, cw, ce);
    if (ucount > 1) consume_finish(1, nw, ne);
#pragma unroll
    for (int u = 2; u < UE; ++u)
      if (ucount > u) consume_finish(u, ew[u - 2], te);
    if (ucount <= UE) {
      STAMP(5);
      return;
    }
    load_unit(UE, cw, ce);                            // more than UE units (grids far below the unit count): continue one unit ahead
    ui0 = UE;
  }
  // while unit ui is consumed, unit ui+1 is in flight; last unit peeled (no dangling prefetch)
  for (int ui = ui0; ui + 1 < ucount; ++ui) {
    load_unit(ui + 1, nw, ne, &ne16);
    consume_finish(ui, cw, ce, &ce16);
#pragma unroll
    for (int f = 0; f < NF; ++f) cw[f] = nw[f];
    if constexpr (EPI16) {
      ce16 = ne16;
    } else if constexpr (NEED_EPI) {
#pragma unroll
      for (int pr = 0; pr < TPU / 2; ++pr) ce[pr] = ne[pr];
    }
  }
  consume_finish(ucount - 1, cw, ce, &ce16);
  STAMP(5);
#undef STAMP
}

template <int PRO, int EPI, int TPU, int NS, int SP, int NCH = 1>
static int launch_sp(const VlaserSkinnyArgs* a, hipStream_t stream, SkinnyP& p, int gx, int lds) {
  if (int rc = set_max_lds_once(skinny_kernel<PRO, EPI, TPU, NS, SP, NCH>, lds)) return rc;
  hipLaunchKernelGGL((skinny_kernel<PRO, EPI, TPU, NS, SP, NCH>), dim3(gx, a->k_splits), dim3(SKT), lds, stream, p);
  VL_LAUNCH_CHECK();
  return 0;
}

template <int PRO, int EPI, int TPU, int NS, int NCH = 1>
static int launch_ns(const VlaserSkinnyArgs* a, hipStream_t stream) {
  SkinnyP p;
  p.a = *a;
  if (p.a.n_valid <= 0) p.a.n_valid = a->N;
  p.kb = a->K / a->k_splits;
  p.xs_stride = p.kb * 2 + 16;
  p.inv_cpr = 8.0f / (float)p.kb;
  const int lds = ((a->M * p.xs_stride + 15) & ~15) + 2 * (SKW - 1) * TPU * 64 * 16 + (PRO == VL_PRO_NORM ? a->K * 2 : 0);
  const int n_units = a->N / (16 * TPU);
  int gx = 256 / a->k_splits;            // <= one block per CU (256 CUs)
  if (gx > n_units) gx = n_units;
  if (gx < 1) gx = 1;
  // r04: the FEWEST workgroups that keep the longest run as short as 256 would: the launch lasts as long as its longest workgroup, and the workgroups that
  // would have finished a unit early only add start spread and contention for the ones still streaming (expert gate/up: 560 units -> 187 x 3 instead of
  // 48 x 3 + 208 x 2: 12.87 -> 12.77 ms per chunk; 170 x 3.3 and 280 x 2 on 256 CUs are both slower).  Same units, same arithmetic: bit-identical outputs.
  static const int gx_full = getenv("VLASER_SKINNY_GX_FULL") ? atoi(getenv("VLASER_SKINNY_GX_FULL")) : 0;                   // A/B: 1 = the r03 grid
  if (!gx_full) { const int longest = (n_units + gx - 1) / gx; gx = (n_units + longest - 1) / longest; }
  p.ulo = n_units / gx;
  p.urem = n_units % gx;
  p.attn_nkv = PRO == VL_PRO_ATTN ? a->K / (128 * a->attn_group) : 0;
  if constexpr (NCH > 1) return launch_sp<PRO, EPI, TPU, NS, -1, NCH>(a, stream, p, gx, lds);
  // exact slab-count variants for the hidden sizes / split factors this path produces (engine.py: ks_o, ks_down)
  if constexpr (PRO == VL_PRO_ATTN && (TPU == 2 || TPU == 1) && (NS == 2 || NS == 3)) {   // exact attention-split count (o_proj of the expert / LLM)
    switch (a->attn_splits) {
      case 1: return launch_sp<PRO, EPI, TPU, NS, 1>(a, stream, p, gx, lds);
      case 2: return launch_sp<PRO, EPI, TPU, NS, 2>(a, stream, p, gx, lds);
      case 3: return launch_sp<PRO, EPI, TPU, NS, 3>(a, stream, p, gx, lds);
      case 4: return launch_sp<PRO, EPI, TPU, NS, 4>(a, stream, p, gx, lds);
      case 5: return launch_sp<PRO, EPI, TPU, NS, 5>(a, stream, p, gx, lds);
      case 6: return launch_sp<PRO, EPI, TPU, NS, 6>(a, stream, p, gx, lds);
      case 7: return launch_sp<PRO, EPI, TPU, NS, 7>(a, stream, p, gx, lds);
      case 8: return launch_sp<PRO, EPI, TPU, NS, 8>(a, stream, p, gx, lds);
      default: break;
    }
  }
  if constexpr (PRO == VL_PRO_NORM && (TPU == 2 || TPU == 1) && (NS == 3 || NS == 6)) {
    switch (a->n_partials) {
      case 0: return launch_sp<PRO, EPI, TPU, NS, 0>(a, stream, p, gx, lds);
      case 2: return launch_sp<PRO, EPI, TPU, NS, 2>(a, stream, p, gx, lds);
      case 3: return launch_sp<PRO, EPI, TPU, NS, 3>(a, stream, p, gx, lds);
      case 5: return launch_sp<PRO, EPI, TPU, NS, 5>(a, stream, p, gx, lds);
      case 6: return launch_sp<PRO, EPI, TPU, NS, 6>(a, stream, p, gx, lds);
      case 7: return launch_sp<PRO, EPI, TPU, NS, 7>(a, stream, p, gx, lds);
      default: break;
    }
  }
  return launch_sp<PRO, EPI, TPU, NS, -1>(a, stream, p, gx, lds);
}

template <int PRO, int EPI>
static int launch(const VlaserSkinnyArgs* a, hipStream_t stream) {
  const int ns = a->K / a->k_splits / (32 * SKW);
  if (a->tiles_per_unit == 1) {          // 16-row units: twice the workgroups of a narrow split-K GEMV (down_proj: 48 units x 5 splits)
    if constexpr (PRO == VL_PRO_PLAIN && EPI == VL_SK_PARTIAL) {
      switch (ns) {
        case 5: return launch_ns<PRO, EPI, 1, 5>(a, stream);
        case 7: return launch_ns<PRO, EPI, 1, 7>(a, stream);
        default: break;
      }
    }
    if constexpr (PRO == VL_PRO_ATTN && EPI == VL_SK_PARTIAL) {
      switch (ns) {
        case 2: return launch_ns<PRO, EPI, 1, 2>(a, stream);
        case 3: return launch_ns<PRO, EPI, 1, 3>(a, stream);
        default: break;
      }
    }
    if constexpr (PRO == VL_PRO_NORM && (EPI == VL_SK_SWIGLU || EPI == VL_SK_QKV_ROPE)) {      // lane-local 16-row units (r03), hidden 768 / 1536
      switch (ns) {
        case 3: return launch_ns<PRO, EPI, 1, 3>(a, stream);
        case 6: return launch_ns<PRO, EPI, 1, 6>(a, stream);
        default: break;
      }
    }
    vlaser_set_error("vlaser_skinny: tiles_per_unit = 1 is built for PLAIN + PARTIAL (5 / 7 K-steps per wave), ATTN + PARTIAL (2 / 3) and NORM + SWIGLU / QKV_ROPE (3 / 6)");
    return -1;
  }
  if (a->tiles_per_unit == 6) {      // 96-row units (r01: 12.3 vs 11.2 us for the expert's gate/up, and 380 B of scratch): removed in r05
    vlaser_set_error("vlaser_skinny: tiles_per_unit = 6 is no longer built (slower than 32-row units and the only variant that spilled)");
    return -1;
  }
  switch (ns) {
    case 1: return launch_ns<PRO, EPI, 2, 1>(a, stream);
    case 2: return launch_ns<PRO, EPI, 2, 2>(a, stream);
    case 3: return launch_ns<PRO, EPI, 2, 3>(a, stream);
    case 4: return launch_ns<PRO, EPI, 2, 4>(a, stream);
    case 5: return launch_ns<PRO, EPI, 2, 5>(a, stream);
    case 6: return launch_ns<PRO, EPI, 2, 6>(a, stream);
    case 7: return launch_ns<PRO, EPI, 2, 7>(a, stream);
    case 8: return launch_ns<PRO, EPI, 2, 8>(a, stream);
    case 14: return launch_ns<PRO, EPI, 2, 7, 2>(a, stream);        // chunked K: 2 x 7 steps (hidden 3584)
    case 16: return launch_ns<PRO, EPI, 2, 8, 2>(a, stream);        // 2 x 8 steps (MLP 20480 / 5 splits)
    default:
      vlaser_set_error("vlaser_skinny: K/(k_splits*256) = %d K-steps per wave unsupported (1..8, 14, 16): raise k_splits or pad K", ns);
      return -1;
  }
}

extern "C" int vlaser_skinny(int pro, int epi, const VlaserSkinnyArgs* a, vl_stream_t s) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(s);
  VL_CHECK(a && a->W && (a->x || pro == VL_PRO_ATTN), "vlaser_skinny: null operand");
  if (pro == VL_PRO_ATTN)
    VL_CHECK(a->attn_m && a->attn_l && a->attn_o && a->attn_splits >= 1 && a->attn_splits <= 8 && a->attn_group >= 1 && a->attn_nq >= 1 &&
                 a->K % (128 * a->attn_group) == 0,
             "vlaser_skinny: bad attention-merge arguments");
  VL_CHECK(a->M >= 1 && a->M <= 16, "vlaser_skinny: M=%d must be in 1..16", a->M);
  VL_CHECK(a->k_splits >= 1 && a->K % (a->k_splits * 32 * SKW) == 0, "vlaser_skinny: K=%d not divisible by k_splits*%d (k_splits=%d)", a->K,
           32 * SKW, a->k_splits);
  VL_CHECK(a->tiles_per_unit == 0 || a->tiles_per_unit == 1 || a->tiles_per_unit == 2, "vlaser_skinny: tiles_per_unit must be 1 or 2");
  VL_CHECK(a->N % (a->tiles_per_unit == 6 ? 96 : (a->tiles_per_unit == 1 ? 16 : 32)) == 0,
           "vlaser_skinny: N=%d must be a multiple of the unit height (pack_skinny pads the weight rows; pass n_valid)", a->N);
  VL_CHECK(a->n_valid <= a->N && (a->n_valid <= 0 || a->n_valid > a->N - (a->tiles_per_unit == 6 ? 96 : (a->tiles_per_unit == 1 ? 16 : 32))),
           "vlaser_skinny: n_valid must lie in the last unit");
  VL_CHECK(((uintptr_t)a->W & 15) == 0 && ((uintptr_t)a->x & 15) == 0, "vlaser_skinny: alignment");
  VL_CHECK((size_t)a->M * (a->K / a->k_splits * 2 + 16) <= 120000, "vlaser_skinny: activation tile does not fit in LDS");
  if (pro == VL_PRO_NORM) {
    VL_CHECK(a->norm_w && a->k_splits == 1, "vlaser_skinny: NORM prologue needs norm_w and k_splits == 1");
    VL_CHECK(a->n_partials == 0 || a->partials, "vlaser_skinny: partials null");
  }
  if (epi == VL_SK_PARTIAL) VL_CHECK(a->out_f32 && (a->n_valid <= 0 || a->n_valid % 4 == 0), "vlaser_skinny: partial needs out_f32 and n_valid %% 4 == 0");
  if (epi != VL_SK_PARTIAL) VL_CHECK(a->k_splits == 1, "vlaser_skinny: only VL_SK_PARTIAL may split K across blocks");
  if (epi != VL_SK_F32 && epi != VL_SK_PARTIAL && epi != VL_SK_SWIGLU)
    VL_CHECK(a->n_valid <= 0 || a->n_valid == a->N, "vlaser_skinny: only F32/PARTIAL/SWIGLU epilogues support a padded N");
  if (epi == VL_SK_SWIGLU) VL_CHECK(a->n_valid <= 0 || a->n_valid % (a->tiles_per_unit == 1 ? 16 : 32) == 0, "vlaser_skinny: SWIGLU needs whole [gate|up] groups");
  if (epi == VL_SK_QKV_ROPE && a->tiles_per_unit == 1) VL_CHECK(a->N % 128 == 0, "vlaser_skinny: QKV_ROPE on 16-row units needs whole heads");
  if (epi == VL_SK_BIAS || epi == VL_SK_BIAS_SILU || epi == VL_SK_QKV_ROPE) VL_CHECK(a->bias, "vlaser_skinny: bias null");
  if (epi == VL_SK_F32) VL_CHECK(a->bias == nullptr || a->n_valid <= 0 || a->n_valid == a->N, "vlaser_skinny: F32 bias needs an un-padded N");
#define SK_CASE(P, E)                                   \
  if (pro == P && epi == E) return launch<P, E>(a, stream);
  SK_CASE(VL_PRO_PLAIN, VL_SK_PARTIAL)
  SK_CASE(VL_PRO_PLAIN, VL_SK_BIAS)
  SK_CASE(VL_PRO_PLAIN, VL_SK_BIAS_SILU)
  SK_CASE(VL_PRO_PLAIN, VL_SK_F32)
  SK_CASE(VL_PRO_ATTN, VL_SK_PARTIAL)
  SK_CASE(VL_PRO_NORM, VL_SK_QKV_ROPE)
  SK_CASE(VL_PRO_NORM, VL_SK_SWIGLU)
  SK_CASE(VL_PRO_NORM, VL_SK_F32)
#undef SK_CASE
  vlaser_set_error("vlaser_skinny: unsupported prologue/epilogue combination %d/%d", pro, epi);
  return -1;
}
