// Flash-style attention for gfx950 (bf16 in, fp32 softmax/accumulate), two kernels:
//
//  attn_prefill<HD>  : many query rows (ViT S=1025, LLM prefill).  Block = 4 waves x 16 query rows; K tile
//                      [64 keys][HD] and V^T tile [HD][64 keys] staged through XOR-swizzled LDS.
//  attn_skinny       : <=16 query tokens per (q-head, batch) over a KV cache (10 Euler steps of the action
//                      expert, greedy decode).  Block = 16 waves, one 32-key chunk per wave straight from
//                      global memory (no LDS staging), flash-decoding merge through LDS.
//
// Both compute S^T = K Q^T with the K fragment as MFMA-A (rows = keys) and Q as MFMA-B (cols = queries) so that
// every lane owns ONE query column: softmax statistics are lane-local (2 xor-shuffles per tile for the max).
// The K rows fed to tile t of a 32-key chunk are permuted (key = (i>>2)*8 + t*4 + (i&3)) so that the two
// 16-key S^T tiles of a lane concatenate into 8 CONSECUTIVE keys = exactly the MFMA-B fragment of P^T for
// O^T = V^T P^T.  V is kept TRANSPOSED in memory ([.., HD, S]) so its MFMA-A fragment is one 16-byte load.
//
// Visibility of key j for query row i (replaces the dense additive masks of the reference,
// pizero_internvl.py:517-603 and HF causal masks):   j < lim1(i)  ||  lo2 <= j < hi2
//   FULL   : lim1 = kv_len                         (ViT, modeling_intern_vit.py:220-224: no mask)
//   CAUSAL : lim1 = min(kv_len, i + 1 + causal_off)
//   PREFIX : lim1 = valid_len[b]; rows >= blk_start additionally see [blk_start, kv_len)
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "../../include/vlaser_hip.h"

#define NEG_BIG (-1.0e30f)

struct AttnP {
  VlaserAttnArgs a;
  // (r04, FULL mode only) InternViT's 1025th token: 1025 keys = 8 tiles of 128 + ONE key, 1025 queries = 16 workgroups of 64 rows + ONE row.  r03 paid a ninth
  // key tile through LDS for the one key and a 17th workgroup walking every tile for the one row: 22.5 us against 14.65 us for 1024 tokens (DESIGN lesson 29).
  int tail_key0;      // >= 0: keys [tail_key0, kv_len) (<= 32) are NOT a tile: every wave takes them as one 32-key chunk straight from global memory
  int tail_qb;        // >= 0: workgroup tail_qb holds <= 16 query rows: its 4 waves take the same rows and split the key TILES between them
  int tail_host;      // 1: there is NO workgroup tail_qb -- workgroup tail_qb - 1 of each head carries those rows as a fifth 16-row tile dealt over its 4 waves
};

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// LDS images of the K tile [64 keys][HD] and the V^T tile [HD][64 keys], 16-byte slots XOR-swizzled per row.  A ds_read_b128 is served
// 16 lanes per cycle over 64 banks, so the 16 rows one lane group touches must land on 16 different 16-byte columns of the 256-byte
// bank row: the S^T fragment's 16 lanes read key rows {0-3, 8-11, 16-19, 24-27} (+4 t) at one slot, the O^T fragment's read 16
// consecutive d rows.  With 128-byte rows two rows share a bank row (row parity picks the half), so the swizzle key must separate
// the 8 rows of equal parity; with 256-byte rows (head_dim 128) all 16.  (r01/r02 used row & 7 / row & 15: 4-way conflicts on K at
// head_dim 64, 2-way on V^T and on K at head_dim 128.)
template <int HD>
__device__ __forceinline__ int k_lds_off(int row, int slot) {
  if constexpr (HD == 128) return row * 256 + ((slot ^ ((row & 3) | (((row >> 3) & 3) << 2))) << 4);
  else return row * 128 + ((slot ^ (((row >> 1) & 1) | (((row >> 3) & 3) << 1))) << 4);
}
template <int TK>
__device__ __forceinline__ int vt_lds_off(int row, int slot) {
  if constexpr (TK == 128) return row * 256 + ((slot ^ (row & 15)) << 4);
  else return row * 128 + ((slot ^ ((row >> 1) & 7)) << 4);
}

// One 32-key chunk straight from global memory against a wave's 16 query rows (the layout of attn_skinny's chunks: K-tile t row r <-> key
// key0 + (r >> 2) * 8 + t * 4 + (r & 3)), online softmax state (m, l, o) updated in place.  Keys >= kv_len are masked; loads are clamped.
template <int HD>
struct DirectChunk {
  static constexpr int DC = HD / 32, DT = HD / 16;
  u32x4 kf[2][DC], vf[DT];
  __device__ __forceinline__ void load(const bf16_t* K, const bf16_t* VT, int key0, int kv_len, int ld_vt, int fr, int g) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int key = key0 + (fr >> 2) * 8 + t * 4 + (fr & 3);
#pragma unroll
      for (int dc = 0; dc < DC; ++dc) kf[t][dc] = ld_global_16(K + (size_t)min(key, kv_len - 1) * HD + dc * 32 + g * 8);
    }
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) vf[dt] = ld_global_16(VT + (size_t)(dt * 16 + fr) * ld_vt + min(key0 + g * 8, ld_vt - 8));
  }
  __device__ __forceinline__ void process(const bf16x8 (&qf)[DC], int key0, int kv_len, float sc, int g, float& m_run, float& l_run, f32x4 (&o)[DT]) {
    f32x4 s[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      f32x4 acc = {0, 0, 0, 0};
#pragma unroll
      for (int dc = 0; dc < DC; ++dc) acc = mfma16(as_bf16x8(kf[t][dc]), qf[dc], acc);
      s[t] = acc;
    }
    float mx = NEG_BIG;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s[t][r] = key0 + g * 8 + t * 4 + r < kv_len ? s[t][r] * sc : NEG_BIG;
        mx = fmaxf(mx, s[t][r]);
      }
    mx = vl_xor32_max(vl_xor16_max(mx));          // (r05) permlane swaps instead of two ds_bpermute round trips in the softmax chain of every tile: same operands
    const float m_new = fmaxf(m_run, mx);
    const float alpha = fast_exp2(m_run - m_new);
    m_run = m_new;
    const float meff = fmaxf(m_new, -1.0e20f);             // nothing visible yet: exp2(-1e30 + 1e20) = 0 without a select
    float pv[8], psum = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pe = fast_exp2(s[t][r] - meff);
        psum += pe;
        pv[t * 4 + r] = pe;
      }
    l_run = l_run * alpha + psum;
    const u32x4 pk = {pack_bf16x2(pv[0], pv[1]), pack_bf16x2(pv[2], pv[3]), pack_bf16x2(pv[4], pv[5]), pack_bf16x2(pv[6], pv[7])};
    // V^T columns past the cache row's padding were clamped (duplicates of real keys): their P is exactly 0
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      f32x4 acc = o[dt];
      acc[0] *= alpha; acc[1] *= alpha; acc[2] *= alpha; acc[3] *= alpha;
      o[dt] = mfma16(as_bf16x8(vf[dt]), as_bf16x8(pk), acc);
    }
  }
};

// KS = key splits INSIDE the workgroup: KS groups of 4 waves share the 64 query rows, group j walks key tiles j, j+KS, ... through
// its own K / V^T staging buffers, and group 0 merges the (m, l, o) triples through LDS at the end (fixed order: deterministic).
// The path's prefill grids are about one 4-wave workgroup per CU (ViT: 17 x 16 = 272, joint prefill 6 x 12 = 72), i.e. one wave
// per SIMD walking a serial chain of key tiles with nothing to hide a tile's load latency behind; the split shortens the chain KS
// times and puts KS waves on every SIMD.
// TK = keys per tile (64 / 128): with about one workgroup per CU a wave is alone on its SIMD and every tile is a serial chain
// barrier -> LDS store -> barrier -> S^T -> max (two cross-lane hops) -> exp -> P V; 128-key tiles halve the number of chains
// per key and give each one twice the independent MFMA / exp work to overlap.
// DENSE (ABI 8): VL_ATTN_DENSE -- every key tile is walked and the visibility comes from a.mask (fp32, additive): the mask values of a tile ride in registers with the tile's
// K / V^T (requested one tile ahead, 4 consecutive keys = 16 bytes per score quad); its own instantiation, so the descriptor modes do not carry its registers
template <int HD, int KS, int TK, bool TAIL = false, bool HOST = false, bool DENSE = false>      // HOST: TAIL with the tail rows hosted by the last full workgroup (p.tail_host; its own instantiation: +2 x 16 registers); TAIL: the FULL-mode tail handling (p.tail_key0 / p.tail_qb); its own instantiation, so the other shapes do not carry its registers
__global__ __launch_bounds__(256 * KS) void attn_prefill_kernel(AttnP p) {
  constexpr int DC = HD / 32;   // d-chunks of 32 for S^T
  constexpr int DT = HD / 16;   // d-tiles of 16 for O^T
  constexpr int NC = TK / 32;   // 32-key chunks per tile
  constexpr int KCH = TK * HD / 2048;  // 16-byte chunks per thread for the K tile (TK*HD*2/16/256)
  constexpr int VCH = TK * HD / 2048;  // same for V^T tile
  constexpr int TILE_BYTES = 2 * TK * HD * 2;
  constexpr int MERGE_WAVE = (DT * 4 + 2) * 256;                 // one wave's (o[DT][4], m, l) as [slot][lane] fp32
  constexpr int MERGE_BYTES = (KS - 1) * 4 * MERGE_WAVE;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // max(KS * TILE_BYTES, MERGE_BYTES)
  const VlaserAttnArgs& a = p.a;
  const int lane = threadIdx.x & 63;
  const int grp = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8);      // key-split group of this wave
  const int tid = threadIdx.x & 255, wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // thread / wave index inside the group
  char* Ks = smem + grp * TILE_BYTES;
  char* Vs = Ks + TK * HD * 2;
  const int fr = lane & 15, g = lane >> 4;
  // TAIL launches are 1-D per batch element with the workgroups of the <= 16 tail rows LAST (ids >= n_full * heads): found with a lab switch that made those 16
  // workgroups exit at once -- the launch still took 22.5 us against 14.8 us for 256 workgroups, i.e. the dispatcher hands out CUs round-robin and, with the
  // light workgroups sprinkled through a 17 x 16 grid, doubles full workgroups up on some CUs while others idle.  All full workgroups first = one per CU
  // (empty tail workgroups dispatched last: 17.2 us; with their work 21.7 us -- a workgroup that walks a head's 256 KB of K / V beside a full one takes as long
  // as the full one, however little it computes: profiles/r04m_vit_attention_tail.md).
  int qb = blockIdx.x, h = blockIdx.y;
  if constexpr (TAIL) {
    if (p.tail_qb >= 0) {
      const int nfull = p.tail_qb * a.n_q_heads;
      if ((int)blockIdx.x < nfull) { h = blockIdx.x / p.tail_qb; qb = blockIdx.x - h * p.tail_qb; }
      else { h = blockIdx.x - nfull; qb = p.tail_qb; }
    }
  }
  const int b = blockIdx.z;
  const int kvh = h / (a.n_q_heads / a.n_kv_heads);
  // TAIL: workgroup p.tail_qb holds the LAST <= 16 query rows (FULL mode).  All four of its waves take those same rows and split the key TILES between them
  // (wave w computes tiles w, w + 4, ...; the staging and its barriers stay cooperative), flash-decoding merge through LDS at the end: a quarter of a full
  // workgroup's issue slots, so the full workgroup it shares a CU with (272 workgroups on 256 CUs at S = 1025) is slowed far less than by a second full one.
  // (First built as a workgroup pulling its chunks straight from global memory: latency-bound, as slow as the full workgroups -- profiles/r04l.)
  const bool tailwg = TAIL && !HOST && qb == p.tail_qb;
  // r04 (tail_host): the <= 16 tail rows ride in the LAST FULL workgroup of their head instead -- a second softmax state per wave, advanced over a quarter of the
  // keys of every tile (wave w: the w-th 32-key chunk), merged like the tail workgroup's.  The grid is exactly tail_qb x heads workgroups (ViT: 256 = one per CU, one
  // round) and 16 of them carry +25 % work, instead of 16 extra workgroups that each walk a head's K / V beside a full one.
  const bool hostwg = HOST && qb == p.tail_qb - 1;
  const int q_row = qb * 64 + (tailwg ? 0 : wave * 16) + fr;  // this lane's query row (within the batch element)

  const bf16_t* Q = reinterpret_cast<const bf16_t*>(a.q) + (size_t)b * a.q_bs + (size_t)h * a.q_hs;
  const bf16_t* K = reinterpret_cast<const bf16_t*>(a.k) + (size_t)b * a.k_bs + (size_t)kvh * a.k_hs;
  const bf16_t* VT = reinterpret_cast<const bf16_t*>(a.vt) + (size_t)b * a.vt_bs + (size_t)kvh * a.vt_hs;

  // visibility
  int lim1, lo2 = 0x7fffffff, hi2 = 0;
  int blk_lim1, blk_has2 = 0;
  {
    const int q_last = min(a.sq, qb * 64 + 64) - 1;
    if (a.mode == VL_ATTN_FULL || DENSE) {
      lim1 = blk_lim1 = a.kv_len;
    } else if (a.mode == VL_ATTN_CAUSAL) {
      lim1 = min(a.kv_len, q_row + 1 + a.causal_off);
      blk_lim1 = min(a.kv_len, q_last + 1 + a.causal_off);
    } else {
      const int vl = a.valid_len ? a.valid_len[b] : a.kv_len;
      lim1 = blk_lim1 = min(vl, a.kv_len);
      if (q_row + a.q_row_off >= a.blk_start) { lo2 = a.blk_start; hi2 = a.kv_len; }
      blk_has2 = (q_last + a.q_row_off >= a.blk_start);
    }
  }

  // Q fragments (MFMA-B: col = query fr, k = d)
  bf16x8 qf[DC];
#pragma unroll
  for (int dc = 0; dc < DC; ++dc) {
    u32x4 v = {0, 0, 0, 0};
    if (q_row < a.sq) v = ld_global_16(Q + (size_t)q_row * a.q_ss + dc * 32 + g * 8);
    qf[dc] = as_bf16x8(v);
  }

  f32x4 o[DT];
#pragma unroll
  for (int i = 0; i < DT; ++i) o[i] = f32x4{0, 0, 0, 0};
  float m_run = NEG_BIG, l_run = 0.f;
  // the host workgroup's second state: rows tail_qb * 64 + fr (rows >= sq: zero queries, never written)
  bf16x8 qf2[HOST ? DC : 1];
  f32x4 o2[HOST ? DT : 1];
  float m2 = NEG_BIG, l2 = 0.f;
  if constexpr (HOST) {
#pragma unroll
    for (int i = 0; i < DT; ++i) o2[i] = f32x4{0, 0, 0, 0};
    const int q_row2 = p.tail_qb * 64 + fr;
#pragma unroll
    for (int dc = 0; dc < DC; ++dc) {
      u32x4 v = {0, 0, 0, 0};
      if (hostwg && q_row2 < a.sq) v = ld_global_16(Q + (size_t)q_row2 * a.q_ss + dc * 32 + g * 8);
      qf2[dc] = as_bf16x8(v);
    }
  }
  const float sc = a.scale * 1.4426950408889634f;  // softmax in base 2

  // the <= 32 keys behind the last full tile (FULL mode): requested now as one chunk straight from global memory, folded in after the tile loop
  DirectChunk<TAIL ? HD : 32> tailc;
  if constexpr (TAIL) { if (p.tail_key0 >= 0) tailc.load(K, VT, p.tail_key0, a.kv_len, a.ld_vt, fr, g); }

  // key-tile schedule: [0, n1) then tiles overlapping [blk_start, kv_len)
  const int n1 = (TAIL && p.tail_key0 >= 0) ? p.tail_key0 / TK : (blk_lim1 + TK - 1) / TK;
  int t2_lo = 0, t2_hi = 0;
  if (blk_has2) { t2_lo = max(n1, a.blk_start / TK); t2_hi = (a.kv_len + TK - 1) / TK; }
  const int n_tiles = n1 + max(0, t2_hi - t2_lo);

  u32x4 rk[KCH], rv[VCH];
  // DENSE: this lane's mask values of the tile in flight (rmk) and of the tile being computed (cmk): row = its query row, 4 consecutive keys per (chunk, t)
  f32x4 rmk[DENSE ? NC : 1][2], cmk[DENSE ? NC : 1][2];
  const float* mrow = nullptr;
  const float inv_scale = DENSE ? 1.0f / a.scale : 0.f;
  if constexpr (DENSE) mrow = a.mask + (size_t)b * a.mask_bs + (size_t)min(q_row, a.sq - 1) * a.mask_rs;
  auto tile_key0 = [&](int it) { return (it < n1 ? it : t2_lo + (it - n1)) * TK; };
  auto load_tile = [&](int it) {
    const int key0 = tile_key0(it);
    if constexpr (DENSE) {
#pragma unroll
      for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int t = 0; t < 2; ++t) rmk[c][t] = *reinterpret_cast<const f32x4*>(mrow + min(key0 + c * 32 + g * 8 + t * 4, (int)a.mask_rs - 4));
    }
#pragma unroll
    for (int i = 0; i < KCH; ++i) {
      const int c = tid + i * 256;              // chunk id: row = c / (HD/8), slot = c % (HD/8)
      const int row = c / (HD / 8), slot = c % (HD / 8);
      const int key = min(key0 + row, a.kv_len - 1);      // clamped, unconditional: keys >= kv_len are never visible
      rk[i] = ld_global_16(K + (size_t)key * HD + slot * 8);
    }
#pragma unroll
    for (int i = 0; i < VCH; ++i) {
      const int c = tid + i * 256;              // row = d = c / (TK/8), slot = c % (TK/8) (8 keys each)
      const int row = c / (TK / 8), slot = c % (TK / 8);
      rv[i] = ld_global_16(VT + (size_t)row * a.ld_vt + min(key0 + slot * 8, a.ld_vt - 8));  // rows are padded to a multiple of 64 keys; clamped beyond
    }
  };
  auto store_tile = [&]() {
    if constexpr (DENSE) {
#pragma unroll
      for (int c = 0; c < NC; ++c) { cmk[c][0] = rmk[c][0]; cmk[c][1] = rmk[c][1]; }
    }
#pragma unroll
    for (int i = 0; i < KCH; ++i) {
      const int c = tid + i * 256;
      *reinterpret_cast<u32x4*>(Ks + k_lds_off<HD>(c / (HD / 8), c % (HD / 8))) = rk[i];
    }
#pragma unroll
    for (int i = 0; i < VCH; ++i) {
      const int c = tid + i * 256;
      *reinterpret_cast<u32x4*>(Vs + vt_lds_off<TK>(c / (TK / 8), c % (TK / 8))) = rv[i];
    }
  };

  // one TK-key tile against this wave's 16 query rows; S^T tiles s[c][t], key(c,t,reg) = key0 + c*32 + g*8 + t*4 + reg
  // (nc_c chunks of 32 keys starting at chunk cb of the tile: the whole tile for a wave's own rows, ONE chunk for the hosted tail rows)
  auto tile_body = [&](int key0, auto masked_c, auto nc_c, int cb, const auto& qf, float& m_run, float& l_run, auto& o) __attribute__((always_inline)) {
    constexpr bool MASKED = decltype(masked_c)::value;
    constexpr int NC = decltype(nc_c)::value;
    f32x4 s[NC][2];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        f32x4 acc = {0, 0, 0, 0};
        const int krow = (cb + c) * 32 + (fr >> 2) * 8 + t * 4 + (fr & 3);
#pragma unroll
        for (int dc = 0; dc < DC; ++dc) {
          bf16x8 kf = as_bf16x8(*reinterpret_cast<const u32x4*>(Ks + k_lds_off<HD>(krow, dc * 4 + g)));
          acc = mfma16(kf, qf[dc], acc);
        }
        s[c][t] = acc;
      }
    // online softmax in base 2 (lane-local per query; max all-reduced over the 4 lane groups); the scale is applied inside the
    // exponent's FMA: max(s) * sc == max(s * sc) for sc > 0
    float mx = NEG_BIG;
    bool vis[NC][2][4];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if constexpr (MASKED && DENSE) {
            // score * scale + mask == (score + mask / scale) * scale: the mask enters before the scale that the exponent's FMA applies
            const int key = key0 + (cb + c) * 32 + g * 8 + t * 4 + r;
            const float mv = cmk[cb + c][t][r];
            const bool v = (key < lim1) && (mv > -1.0e30f);
            vis[c][t][r] = v;
            s[c][t][r] = v ? __builtin_fmaf(mv, inv_scale, s[c][t][r]) : s[c][t][r];
            if (v) mx = fmaxf(mx, s[c][t][r]);
          } else if constexpr (MASKED) {
            const int key = key0 + (cb + c) * 32 + g * 8 + t * 4 + r;
            const bool v = (key < lim1) || (key >= lo2 && key < hi2);
            vis[c][t][r] = v;
            if (v) mx = fmaxf(mx, s[c][t][r]);
          } else {
            vis[c][t][r] = true;
            mx = fmaxf(mx, s[c][t][r]);
          }
        }
    mx = vl_xor32_max(vl_xor16_max(mx));          // (r05) permlane swaps instead of two ds_bpermute round trips in the softmax chain of every tile: same operands
    const float m_new = fmaxf(m_run, mx * sc);
    const float alpha = fast_exp2(m_run - m_new);
    m_run = m_new;
    float psum = 0.f;
    bf16x8 pf[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      float pv[8];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float pe = fast_exp2(__builtin_fmaf(s[c][t][r], sc, -m_new));
          if constexpr (MASKED) pe = vis[c][t][r] ? pe : 0.f;
          psum += pe;
          pv[t * 4 + r] = pe;
        }
      u32x4 pk = {pack_bf16x2(pv[0], pv[1]), pack_bf16x2(pv[2], pv[3]), pack_bf16x2(pv[4], pv[5]), pack_bf16x2(pv[6], pv[7])};
      pf[c] = as_bf16x8(pk);
    }
    l_run = l_run * alpha + psum;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      f32x4 acc = o[dt];
      acc[0] *= alpha; acc[1] *= alpha; acc[2] *= alpha; acc[3] *= alpha;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        bf16x8 vf = as_bf16x8(*reinterpret_cast<const u32x4*>(Vs + vt_lds_off<TK>(dt * 16 + fr, (cb + c) * 4 + g)));
        acc = mfma16(vf, pf[c], acc);
      }
      o[dt] = acc;
    }
  };
  // smallest prefix limit among this wave's 16 rows (wave-uniform): a tile ending at or below it needs no masks
  const int wave_lim1 = DENSE ? 0 : (a.mode == VL_ATTN_CAUSAL ? min(a.kv_len, qb * 64 + wave * 16 + 1 + a.causal_off) : blk_lim1);      // (DENSE: every tile takes the masked body)

  const int n_it = (n_tiles + KS - 1) / KS;                   // block-uniform trip count; a group past its last tile idles at the barriers
  if (n_tiles > 0) load_tile(min(grp, n_tiles - 1));
  for (int itg = 0; itg < n_it; ++itg) {
    const int it = itg * KS + grp;
    __syncthreads();  // previous tile's LDS reads are done
    store_tile();
    __syncthreads();
    load_tile(min(it + KS, n_tiles - 1));                      // unconditional (clamped): a conditional load costs a full vmcnt drain
    if (it >= n_tiles) continue;
    if (tailwg && (it & 3) != wave) continue;                  // (wave-uniform) the tail workgroup's waves take every fourth tile each
    const int key0 = tile_key0(it);

    // a tile every row of this wave sees in full (all interior tiles of FULL / PREFIX, the tiles left of the diagonal of CAUSAL)
    // skips the visibility arithmetic: the loop is VALU-issue bound (about 230 VALU + 17 transcendental instructions per tile
    // against 16 MFMAs at head_dim 64), and the masks were 40 % of it
    using all_c = std::integral_constant<int, NC>;
    if (key0 + TK <= wave_lim1) tile_body(key0, std::false_type{}, all_c{}, 0, qf, m_run, l_run, o);
    else tile_body(key0, std::true_type{}, all_c{}, 0, qf, m_run, l_run, o);
    if constexpr (HOST) {
      // the hosted rows: wave w takes the w-th 32-key chunk of EVERY tile (+25 % per wave and tile; a whole tile every fourth trip would double the time
      // of every trip -- the tiles are barrier-stepped: measured 24.6 us against 21.5 for the tail workgroups).  FULL mode: the limits do not depend on the row
      static_assert(!HOST || NC == 4, "HOST deals one 32-key chunk of a 128-key tile to each of the 4 waves");
      if (hostwg) {
        using one_c = std::integral_constant<int, 1>;
        if (key0 + TK <= wave_lim1) tile_body(key0, std::false_type{}, one_c{}, wave, qf2, m2, l2, o2);
        else tile_body(key0, std::true_type{}, one_c{}, wave, qf2, m2, l2, o2);
      }
    }
  }

  if constexpr (TAIL) {
    if (p.tail_key0 >= 0 && grp == 0 && (!tailwg || wave == 0)) tailc.process(qf, p.tail_key0, a.kv_len, sc, g, m_run, l_run, o);
    if constexpr (HOST) { if (p.tail_key0 >= 0 && hostwg && wave == 0) tailc.process(qf2, p.tail_key0, a.kv_len, sc, g, m2, l2, o2); }
  }
  // merge of the four waves' partial softmax states (same rows, disjoint keys) through LDS, fixed order; writes rows out_qb * 64 ...
  auto merge4 = [&](float m_run, float l_run, const auto& o, int out_qb) __attribute__((always_inline)) {
    constexpr int WS = 32 + 16 * HD;
    __syncthreads();                                           // every wave is done with the staged tiles: the area becomes the merge buffer
    float* wm = reinterpret_cast<float*>(smem) + wave * WS;
    {
      float l_tot = vl_xor32_sum(vl_xor16_sum(l_run));
      if (g == 0) { wm[fr] = m_run; wm[16 + fr] = l_tot; }
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) *reinterpret_cast<f32x4*>(wm + 32 + fr * HD + dt * 16 + g * 4) = o[dt];
    }
    __syncthreads();
    constexpr int DPT = HD / 16;                               // head-dim columns per thread: 16 rows x HD over 256 threads
    const int row = threadIdx.x >> 4, d0 = (threadIdx.x & 15) * DPT;
    const float* base = reinterpret_cast<const float*>(smem);
    float M = NEG_BIG;
#pragma unroll
    for (int w = 0; w < 4; ++w) M = fmaxf(M, base[w * WS + row]);
    float L = 0.f, acc[DPT];
#pragma unroll
    for (int j = 0; j < DPT; ++j) acc[j] = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float f = fast_exp2(base[w * WS + row] - M);
      L += base[w * WS + 16 + row] * f;
#pragma unroll
      for (int j = 0; j < DPT; ++j) acc[j] += base[w * WS + 32 + row * HD + d0 + j] * f;
    }
    const int q_out = out_qb * 64 + row;
    if (q_out < a.sq) {
      const float inv = L > 0.f ? 1.0f / L : 0.f;
      if (a.lse_out && d0 == 0) a.lse_out[((size_t)b * a.n_q_heads + h) * a.sq + q_out] = M + __builtin_amdgcn_logf(L);
      bf16_t* O = reinterpret_cast<bf16_t*>(a.out) + (size_t)b * a.o_bs + (size_t)q_out * a.o_ss + h * HD + d0;
#pragma unroll
      for (int j = 0; j < DPT; j += 2) *reinterpret_cast<uint32_t*>(O + j) = pack_bf16x2(acc[j] * inv, acc[j + 1] * inv);
    }
  };
  if constexpr (TAIL) {
    if (tailwg) { merge4(m_run, l_run, o, qb); return; }
    if constexpr (HOST) { if (hostwg) merge4(m2, l2, o2, p.tail_qb); }
  }

  if constexpr (KS > 1) {
    // merge the groups' partial softmax states: lane-for-lane (every group holds the same (row, d) elements in the same registers)
    __syncthreads();                                           // all tile reads done: the staging area becomes the merge area
    if (grp > 0) {
      float* mw = reinterpret_cast<float*>(smem + ((grp - 1) * 4 + wave) * MERGE_WAVE) + lane;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) mw[(dt * 4 + r) * 64] = o[dt][r];
      mw[DT * 4 * 64] = m_run;
      mw[(DT * 4 + 1) * 64] = l_run;
    }
    __syncthreads();
    if (grp > 0) return;
#pragma unroll
    for (int j = 1; j < KS; ++j) {
      const float* rw = reinterpret_cast<const float*>(smem + ((j - 1) * 4 + wave) * MERGE_WAVE) + lane;
      const float m_o = rw[DT * 4 * 64], l_o = rw[(DT * 4 + 1) * 64];
      const float m_new = fmaxf(m_run, m_o);
      const float fa = fast_exp2(m_run - m_new), fb = fast_exp2(m_o - m_new);
      m_run = m_new;
      l_run = l_run * fa + l_o * fb;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) o[dt][r] = o[dt][r] * fa + rw[(dt * 4 + r) * 64] * fb;
    }
  }
  float l_tot = vl_xor32_sum(vl_xor16_sum(l_run));
  if (q_row < a.sq) {
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    // base-2 log-sum-exp of the scaled scores, for the fused backward (vlaser_attn_bwd): P = exp2(s * scale * log2 e - lse)
    if (a.lse_out && g == 0) a.lse_out[((size_t)b * a.n_q_heads + h) * a.sq + q_row] = m_run + __builtin_amdgcn_logf(l_tot);
    bf16_t* O = reinterpret_cast<bf16_t*>(a.out) + (size_t)b * a.o_bs + (size_t)q_row * a.o_ss + h * HD;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      u32x2 pk = {pack_bf16x2(o[dt][0] * inv, o[dt][1] * inv), pack_bf16x2(o[dt][2] * inv, o[dt][3] * inv)};
      *reinterpret_cast<u32x2*>(O + dt * 16 + g * 4) = pk;
    }
  }
}

// ------------------------------------------------------------------------------------------------ skinny
// <= 16 query tokens per batch element over a KV cache, GQA-aware and key-split:
//   grid (n_kv_heads, n_splits, B); block = 4 waves; rows = (q-head-in-group hg, token) pairs of ONE kv head
//   (r = hg*nq + tok, G*nq <= 32 -> two 16-row MFMA tiles), so K / V^T of a kv head are read ONCE for all its
//   q heads; split s owns a contiguous run of 32-key chunks, one chunk per wave per pass, every K and V^T
//   fragment of the chunk requested up front.  Per-CU traffic is ~1/n_splits of the cache instead of all of it.
// Output = flash-decoding partials per (b, kv head, split): m, l [32] and unnormalised o [32][128] (fp32); they are
// merged by the CONSUMER (o_proj skinny kernel, VL_PRO_ATTN prologue) -- no second launch, no atomics.
// Visibility is uniform over the rows of a batch element: keys [0, lim1) U [lo2, hi2).
#define SKA_WAVES 4
__global__ __launch_bounds__(256) void attn_skinny_kernel(AttnP p) {
#ifdef VL_KERNARG_UP_FRONT
  vl_kernargs_up_front(p);
#endif
  constexpr int HD = 128, DC = 4, DT = 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [4 waves][ m[32] l[32] o[32][128] ] fp32
  const VlaserAttnArgs& a = p.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, g = lane >> 4;
  const int kvh = blockIdx.x, split = blockIdx.y, b = blockIdx.z;
  // optional in-kernel timeline (tools/micro/chain_timeline.py): [0] start, [1] first chunk processed, [2] merge barrier passed, [3] end
  unsigned long long* dbg = a.dbg ? a.dbg + (((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 : nullptr;
#define STAMP(i) do { if (dbg && tid == 0) dbg[i] = wall_clock64(); } while (0)
  STAMP(0);
  const int G = a.n_q_heads / a.n_kv_heads, nq = a.sq, nrows = G * nq;
  const bf16_t* K = reinterpret_cast<const bf16_t*>(a.k) + (size_t)b * a.k_bs + (size_t)kvh * a.k_hs;
  const bf16_t* VT = reinterpret_cast<const bf16_t*>(a.vt) + (size_t)b * a.vt_bs + (size_t)kvh * a.vt_hs;

  // chunk schedule depends on kernel arguments only (not on valid_len[b]): loads never wait for that value
  const int n_chunks = (a.kv_len + 31) >> 5;
  const int cps = (n_chunks + a.n_splits - 1) / a.n_splits;
  const int c_begin = split * cps, c_end = min(n_chunks, c_begin + cps);

  bf16x8 qf[2][DC];
  int row_hi2[2];                                      // per query row: end of the visible part of the [blk_start, kv_len) block
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    // rows past nrows shadow the last real row: their loads are unconditional (a load behind `if (r < nrows)` is its own basic block
    // and made hipcc drain ALL of Q with vmcnt(0) before the first K / V^T request: one extra memory round trip per launch) and
    // their results are never stored
    const int r = min(qt * 16 + fr, nrows - 1);
    const int hg = (int)(((float)r + 0.5f) * __builtin_amdgcn_rcpf((float)nq)), tok = r - hg * nq;     // r / nq without an integer division
    row_hi2[qt] = (tok == 0 && a.first_tok_kv_len > 0) ? a.first_tok_kv_len : 0x7fffffff;
    const bf16_t* Q = reinterpret_cast<const bf16_t*>(a.q) + (size_t)b * a.q_bs + (size_t)(kvh * G + hg) * a.q_hs + (size_t)tok * a.q_ss;
#pragma unroll
    for (int dc = 0; dc < DC; ++dc) qf[qt][dc] = as_bf16x8(ld_global_16(Q + dc * 32 + g * 8));      // unconditional (row clamped above)
  }
  int lim1 = a.kv_len, lo2 = 0x7fffffff, hi2 = 0;
  if (a.mode == VL_ATTN_PREFIX) {
    lim1 = min(a.valid_len ? vl_sload_i32(a.valid_len + b) : a.kv_len, a.kv_len);      // scalar load: not in the vmcnt queue (see vl_sload_i32)
    lo2 = a.blk_start; hi2 = a.kv_len;
  }
  f32x4 o[2][DT];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int i = 0; i < DT; ++i) o[qt][i] = f32x4{0, 0, 0, 0};
  float m_run[2] = {NEG_BIG, NEG_BIG}, l_run[2] = {0.f, 0.f};
  const float sc = a.scale * 1.4426950408889634f;

  // The wave's FIRST chunk is requested unconditionally (clamped for waves without one) in straight-line code right behind the Q
  // loads: a load inside the chunk loop sits behind hipcc's loop-header `vmcnt(0)`, i.e. K / V^T used to wait for Q's round trip.
  auto load_chunk = [&](int key0, u32x4 (&kf)[2][DC], u32x4 (&vf)[DT]) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int key = key0 + (fr >> 2) * 8 + t * 4 + (fr & 3);
#pragma unroll
      for (int dc = 0; dc < DC; ++dc)
        kf[t][dc] = ld_global_16(K + (size_t)min(key, a.kv_len - 1) * HD + dc * 32 + g * 8);   // clamped, unconditional: keys >= kv_len are never visible
    }
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) vf[dt] = ld_global_16(VT + (size_t)(dt * 16 + fr) * a.ld_vt + key0 + g * 8);
  };
  auto process_chunk = [&](int key0, u32x4 (&kf)[2][DC], u32x4 (&vf)[DT]) __attribute__((always_inline)) {
    bool visk[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = key0 + g * 8 + t * 4 + r;
        visk[t][r] = (key < lim1) || (key >= lo2 && key < hi2);
      }
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      if (qt * 16 >= nrows) continue;
      bool vis[2][4];                                  // this lane's query row (fr of tile qt) x its 8 keys
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = key0 + g * 8 + t * 4 + r;
          vis[t][r] = visk[t][r] && (key < lim1 || key < row_hi2[qt]);
        }
      f32x4 s[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        f32x4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int dc = 0; dc < DC; ++dc) acc = mfma16(as_bf16x8(kf[t][dc]), qf[qt][dc], acc);
        s[t] = acc;
      }
      float mx = NEG_BIG;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          s[t][r] *= sc;
          if (vis[t][r]) mx = fmaxf(mx, s[t][r]);
        }
      mx = vl_xor32_max(vl_xor16_max(mx));
      const float m_new = fmaxf(m_run[qt], mx);
      const float alpha = fast_exp2(m_run[qt] - m_new);
      m_run[qt] = m_new;
      float pv[8], psum = 0.f;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pe = vis[t][r] ? fast_exp2(s[t][r] - m_new) : 0.f;
          psum += pe;
          pv[t * 4 + r] = pe;
        }
      l_run[qt] = l_run[qt] * alpha + psum;
      u32x4 pk = {pack_bf16x2(pv[0], pv[1]), pack_bf16x2(pv[2], pv[3]), pack_bf16x2(pv[4], pv[5]), pack_bf16x2(pv[6], pv[7])};
      const bf16x8 pf = as_bf16x8(pk);
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        f32x4 acc = o[qt][dt];
        acc[0] *= alpha; acc[1] *= alpha; acc[2] *= alpha; acc[3] *= alpha;
        o[qt][dt] = mfma16(as_bf16x8(vf[dt]), pf, acc);
      }
    }
  };
  {
    u32x4 kf[2][DC], vf[DT];
    const int ci0 = c_begin + wave;
    load_chunk(min(ci0, n_chunks - 1) << 5, kf, vf);          // n_chunks >= 1; the chunk's keys lie inside the padded cache row
    if (ci0 < c_end) process_chunk(ci0 << 5, kf, vf);
    if (dbg) { asm volatile("" ::"v"(o[0][0][0])); STAMP(1); }
    for (int ci = ci0 + SKA_WAVES; ci < c_end; ci += SKA_WAVES) {
      load_chunk(ci << 5, kf, vf);
      process_chunk(ci << 5, kf, vf);
    }
  }

  // in-block merge of the 4 waves through LDS, then one partial per (b, kvh, split)
  constexpr int WS = 64 + 32 * 128;
  float* wm = reinterpret_cast<float*>(smem) + wave * WS;
  const int nwa = min(SKA_WAVES, c_end - c_begin);     // waves that owned at least one chunk (block-uniform)
  if (wave < nwa) {
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    float l_tot = vl_xor32_sum(vl_xor16_sum(l_run[qt]));
    if (g == 0) { wm[qt * 16 + fr] = m_run[qt]; wm[32 + qt * 16 + fr] = l_tot; }
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) *reinterpret_cast<f32x4*>(wm + 64 + (qt * 16 + fr) * 128 + dt * 16 + g * 4) = o[qt][dt];
  }
  }
  __syncthreads();
  STAMP(2);
  {
    const size_t pidx = ((size_t)b * a.n_kv_heads + kvh) * a.n_splits + split;
    float* PM = a.part_m + pidx * 32;
    float* PL = a.part_l + pidx * 32;
    float* PO = a.part_o + pidx * 32 * 128;
    const float* base = reinterpret_cast<const float*>(smem);
    // thread -> (row = tid / 8, 16 consecutive d = (tid % 8) * 16)
    const int row = tid >> 3, d0 = (tid & 7) * 16;
    if (row < nrows) {
      float M = NEG_BIG;
#pragma unroll
      for (int w = 0; w < SKA_WAVES; ++w)
        if (w < nwa) M = fmaxf(M, base[w * WS + row]);
      float Lsum = 0.f;
      f32x4 acc[4] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
#pragma unroll
      for (int w = 0; w < SKA_WAVES; ++w) {
        if (w >= nwa) break;
        const float* bw = base + w * WS;
        const float f = fast_exp2(bw[row] - M);
        Lsum += bw[32 + row] * f;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const f32x4 ov = *reinterpret_cast<const f32x4*>(bw + 64 + row * 128 + d0 + q4 * 4);
          acc[q4][0] += ov[0] * f; acc[q4][1] += ov[1] * f; acc[q4][2] += ov[2] * f; acc[q4][3] += ov[3] * f;
        }
      }
      if ((tid & 7) == 0) { PM[row] = M; PL[row] = Lsum; }
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) *reinterpret_cast<f32x4*>(PO + row * 128 + d0 + q4 * 4) = acc[q4];
    }
  }
  STAMP(3);
#undef STAMP
}

extern "C" int vlaser_attn_prefill(const VlaserAttnArgs* a, vl_stream_t s) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(s);
  VL_CHECK(a && a->q && a->k && a->vt && a->out, "vlaser_attn_prefill: null pointer");
  VL_CHECK(a->head_dim == 64 || a->head_dim == 128, "vlaser_attn_prefill: head_dim %d unsupported", a->head_dim);
  VL_CHECK(a->n_q_heads % a->n_kv_heads == 0, "vlaser_attn_prefill: GQA group mismatch");
  VL_CHECK(a->ld_vt % 64 == 0, "vlaser_attn_prefill: V^T row length must be padded to a multiple of 64 keys");
  VL_CHECK(a->kv_len <= a->ld_vt, "vlaser_attn_prefill: kv_len exceeds cache");
  VL_CHECK(a->sq > 0 && a->batch > 0, "vlaser_attn_prefill: empty");
  AttnP p; p.a = *a; p.tail_key0 = -1; p.tail_qb = -1; p.tail_host = 0;
  dim3 grid((a->sq + 63) / 64, a->n_q_heads, a->batch);
  if (a->mode == VL_ATTN_DENSE) {
    VL_CHECK(a->head_dim == 128 && a->mask && a->mask_rs % 4 == 0 && a->mask_rs >= ((a->kv_len + 63) & ~63) && (((uintptr_t)a->mask) & 15) == 0 && a->mask_bs % 4 == 0,
             "vlaser_attn_prefill: VL_ATTN_DENSE needs head_dim 128 and a 16-byte aligned fp32 mask whose row stride is a multiple of 4 >= kv_len rounded up to 64");
    constexpr int lds = 2 * 64 * 128 * 2;
    if (int rc = set_max_lds_once(attn_prefill_kernel<128, 1, 64, false, false, true>, lds)) return rc;
    hipLaunchKernelGGL((attn_prefill_kernel<128, 1, 64, false, false, true>), grid, dim3(256), lds, stream, p);
    VL_LAUNCH_CHECK();
    return 0;
  }
  // tile / split choice, measured (tools/micro/attn_lab.py, profiles/r02k_attn.md): a 2-way in-workgroup key split pays only when
  // the grid does not even reach one workgroup per CU (joint prefill: 72 workgroups, 11.4 -> 10.8 us); 128-key tiles only for the
  // unmasked head_dim-64 case around one workgroup per CU (ViT, 1 tile: 23.9 -> 22.8 us); more registers per wave (a forced
  // 4 waves / SIMD) and a 4-way split both lose everywhere
  const long blocks = (long)grid.x * grid.y * grid.z;
  static const int force_ks = getenv("VLASER_ATTN_KS") ? atoi(getenv("VLASER_ATTN_KS")) : 0;      // tuning / A-B
  // r03: the split is 8 % faster alone at 108 workgroups (S = 560: 14.7 vs 15.9 us) but stretches x2.5 instead of x1.5 when the SFT step's AdamW streams
  // beside it (tools/micro/contention_lab.py; whole step 23.4 -> 22.8 ms with the split off): only grids below 96 workgroups split
  int ks = force_ks == 1 || force_ks == 2 ? force_ks : (blocks <= 96 ? 2 : 1);
  const int tk = (a->head_dim == 64 && a->mode == VL_ATTN_FULL && blocks <= 512 && ks == 1) ? 128 : 64;
  if (a->kv_len <= tk) ks = 1;
  static const int no_tail = getenv("VLASER_ATTN_NO_TAIL") ? atoi(getenv("VLASER_ATTN_NO_TAIL")) : 0;                       // A/B: the r03 schedule
  if (a->mode == VL_ATTN_FULL && !no_tail && ks == 1) {      // (the 2-way key-split variant at head_dim 128 has no registers to spare for the tail chunk)
    const int tail = a->kv_len % tk;
    if (tail >= 1 && tail <= 32 && a->kv_len > tk) p.tail_key0 = a->kv_len - tail;
    if (a->sq % 64 >= 1 && a->sq % 64 <= 16 && a->sq > 64) p.tail_qb = a->sq / 64;
    static const int tail_wg = getenv("VLASER_ATTN_TAIL_WG") ? atoi(getenv("VLASER_ATTN_TAIL_WG")) : 0;                     // A/B: the tail rows in workgroups of their own (first r04 form)
    if (p.tail_qb >= 0 && !tail_wg && tk == 128) p.tail_host = 1;          // the one-round grids (tk == 128 <=> <= 512 workgroups); larger grids keep the tail workgroups
  }
  static const int dbg_attn = getenv("VLASER_ATTN_DEBUG") ? 1 : 0;
  if (dbg_attn) fprintf(stderr, "attn_prefill: sq %d kv %d hd %d mode %d blocks %ld ks %d tk %d tail_key0 %d tail_qb %d\n", a->sq, a->kv_len, a->head_dim, a->mode, blocks, ks, tk, p.tail_key0, p.tail_qb);
#define VL_ATTN_LAUNCH(HD_, KS_, TK_)                                                                                       \
  do {                                                                                                                      \
    constexpr int tile = 2 * TK_ * HD_ * 2, merge = (KS_ - 1) * 4 * (HD_ / 16 * 4 + 2) * 256;                                \
    constexpr int tailm = 4 * KS_ * (32 + 16 * HD_) * 4;                                                                    \
    constexpr int lds0 = KS_ * tile > merge ? KS_ * tile : merge;                                                           \
    constexpr int lds = lds0 > tailm ? lds0 : tailm;                                                                        \
    if (p.tail_host) {                                                                                                      \
      if constexpr (HD_ == 64 && KS_ == 1 && TK_ == 128) {                                                                  \
        if (int rc = set_max_lds_once(attn_prefill_kernel<HD_, KS_, TK_, true, true>, lds)) return rc;                      \
        hipLaunchKernelGGL((attn_prefill_kernel<HD_, KS_, TK_, true, true>), dim3((grid.x - 1) * grid.y, 1, grid.z), dim3(256 * KS_), lds, stream, p); \
      }                                                                                                                     \
    } else if (p.tail_key0 >= 0 || p.tail_qb >= 0) {                                                                        \
      if constexpr (KS_ == 1) {      /* tails are only set with ks == 1 (above): the 2-way key-split variant would spill (196 B of scratch) and is not built */ \
        if (int rc = set_max_lds_once(attn_prefill_kernel<HD_, KS_, TK_, true>, lds)) return rc;                            \
        const dim3 tgrid = p.tail_qb >= 0 ? dim3(grid.x * grid.y, 1, grid.z) : grid;                                        \
        hipLaunchKernelGGL((attn_prefill_kernel<HD_, KS_, TK_, true>), tgrid, dim3(256 * KS_), lds, stream, p);             \
      }                                                                                                                     \
    } else {                                                                                                                \
      if (int rc = set_max_lds_once(attn_prefill_kernel<HD_, KS_, TK_>, lds)) return rc;                                    \
      hipLaunchKernelGGL((attn_prefill_kernel<HD_, KS_, TK_>), grid, dim3(256 * KS_), lds, stream, p);                      \
    }                                                                                                                       \
  } while (0)
#define VL_ATTN_PICK(HD_)                                                                                                   \
  do {                                                                                                                      \
    if (ks == 2) VL_ATTN_LAUNCH(HD_, 2, 64);                                                                           \
    else if (tk == 128) VL_ATTN_LAUNCH(HD_, 1, 128);                                                                        \
    else VL_ATTN_LAUNCH(HD_, 1, 64);                                                                                        \
  } while (0)
  if (a->head_dim == 128) VL_ATTN_PICK(128); else VL_ATTN_PICK(64);
#undef VL_ATTN_PICK
#undef VL_ATTN_LAUNCH
  VL_LAUNCH_CHECK();
  return 0;
}

extern "C" int vlaser_attn_skinny(const VlaserAttnArgs* a, vl_stream_t s) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(s);
  VL_CHECK(a && a->q && a->k && a->vt, "vlaser_attn_skinny: null pointer");
  VL_CHECK(a->part_m && a->part_l && a->part_o, "vlaser_attn_skinny: partial buffers null");
  VL_CHECK(a->head_dim == 128, "vlaser_attn_skinny: head_dim must be 128");
  VL_CHECK(a->n_q_heads % a->n_kv_heads == 0, "vlaser_attn_skinny: GQA group mismatch");
  VL_CHECK(a->sq >= 1 && a->sq * (a->n_q_heads / a->n_kv_heads) <= 32, "vlaser_attn_skinny: group*tokens must be <= 32 (tokens=%d)", a->sq);
  VL_CHECK(a->ld_vt % 32 == 0 && a->kv_len <= a->ld_vt, "vlaser_attn_skinny: bad cache geometry");
  VL_CHECK(a->n_splits >= 1 && a->n_splits <= 8, "vlaser_attn_skinny: n_splits must be 1..8");
  VL_CHECK(a->mode == VL_ATTN_FULL || a->mode == VL_ATTN_PREFIX, "vlaser_attn_skinny: mode must be FULL or PREFIX");
  AttnP p; p.a = *a;
  const int lds = SKA_WAVES * (64 + 32 * 128) * 4;
  if (int rc = set_max_lds_once(attn_skinny_kernel, lds)) return rc;
  hipLaunchKernelGGL(attn_skinny_kernel, dim3(a->n_kv_heads, a->n_splits, a->batch), dim3(64 * SKA_WAVES), lds, stream, p);
  VL_LAUNCH_CHECK();
  return 0;
}
