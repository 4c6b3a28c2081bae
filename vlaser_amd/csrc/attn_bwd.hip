// Fused backward of the prefill attention for gfx950 (r03): no materialised score matrices, no atomics, two kernels.
//
// Forward (attn.hip): O = softmax(scale Q K^T + mask) V, base-2 log-sum-exp per (head, query) kept in `lse`.  With P = exp2(s - lse):
//   D_q  = <dO_q, O_q>,   dP = dO V^T,   dS = P o (dP - D) * scale,   dQ = dS K,   dK = dS^T Q,   dV = P^T dO.
// dQ is a sum over keys, dK / dV are sums over queries: instead of atomics each gets its own kernel that owns its output rows --
//   attn_bwd_dq_kernel   workgroup = (64 queries, q head), loop over the visible 64-key tiles;   also writes D for the second kernel
//   attn_bwd_dkv_kernel  workgroup = (64 keys, q head),    loop over the 64-query tiles that see them
// (7 tile products instead of the minimal 5; everything deterministic).  dK / dV are written per Q head; vlaser_rope_bwd_pack sums the
// heads of a kv group in a fixed order, exactly as with the per-head TN GEMMs this replaces (7 launches per layer through [heads, S, S]
// fp32 / bf16 matrices in r02).
//
// MFMA operand plumbing (16x16x32 bf16; A: lane -> row lane&15, 8 k values; B: lane -> col lane&15, 8 k values; C: col lane&15, rows (lane>>4)*4 + r):
//   * products that contract over the head dimension take both operands as rows of row-major tiles (k = 8 consecutive d);
//   * products that contract over keys (dQ) or queries (dK, dV) need a TRANSPOSED operand: it is read out of the row-major LDS tile with gfx950's
//     transposing `ds_read_tr16_b64` (gemm.hip's tr_frag): k slot e = 4h + j of lane group g <-> tile row 16h + 4g + j;
//   * the other operand of those products is the P / dS tile just computed: its 16-row MFMA tiles are assigned to rows c*32 + 16t + (0..15), so lane
//     group g holds rows 16t + 4g + r -- e = 4t + r is then exactly the transposing read's k order, and the accumulator registers convert to an operand
//     fragment with a pack, no LDS round trip.
#include "common.h"
#include "../../include/vlaser_hip.h"

typedef __attribute__((ext_vector_type(4))) short ab_s16x4;
#define AB_PQ 288      // bytes per row of a [64][128] tile (conflict-free transposing reads: gemm.hip)
#define AB_PV 160      // bytes per row of a [128][64] tile

__device__ __forceinline__ bf16x8 ab_tr_frag(const char* tile, int pitch, int col0, int krow0, int fq, int fr) {
  const char* p0 = tile + (krow0 + 4 * fq + (fr >> 2)) * pitch + (col0 + 4 * (fr & 3)) * 2;
  typedef __attribute__((address_space(3))) ab_s16x4* lds_p;
  union { ab_s16x4 h[2]; bf16x8 b; } u;
  u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(uint32_t)(uintptr_t)p0);
  u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(uint32_t)(uintptr_t)(p0 + 16 * pitch));
  return u.b;
}

// lab build only (-DAB_TIMELINE, tools/micro/attn_bwd_timeline.py): cycle stamps of wave 0 of the workgroup with the longest tile chain
#ifdef AB_TIMELINE
__device__ long long ab_dbg[2][64];
#define AB_STAMP(k, i, cond)                                                                            \
  {                                                                                                     \
    const int _i = (i);                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
    if ((cond) && threadIdx.x == 0 && blockIdx.y == 0 && _i < 60) ab_dbg[k][_i] = clock64();            \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
  }
extern "C" int vlaser_attn_bwd_debug_read(long long* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(ab_dbg), sizeof(long long) * 128); }
#else
#define AB_STAMP(k, i, cond)
#endif

struct AttnBwdP {
  const bf16_t *q, *k, *vt, *o, *d_o;
  const float* lse;
  float* delta;
  bf16_t *dq, *dk, *dv;
  int S, n_q, n_kv, s_max;
  float scale;
  int causal, kv_valid;
};

// 8 values of one row in the transposing read's k order: columns c0 + 16h + 4g + j  (two 8-byte loads)
__device__ __forceinline__ bf16x8 ab_load_tr_order(const bf16_t* row, int c0, int g) {
  const u32x2 lo = *reinterpret_cast<const u32x2*>(row + c0 + 4 * g);
  const u32x2 hi = *reinterpret_cast<const u32x2*>(row + c0 + 16 + 4 * g);
  return as_bf16x8(u32x4{lo[0], lo[1], hi[0], hi[1]});
}

// ---------------------------------------------------------------------------------------------- dQ (+ D)
// TK = keys per staged tile; KS = wave groups (of 4 waves) that take the key tiles in turn and merge their dQ sums at the end: at SFT lengths the grid
// is 108 workgroups on 256 CUs and one wave per SIMD runs LDS reads, MFMAs and the softmax arithmetic back to back -- a second group per
// workgroup gives every SIMD a second instruction stream to overlap them with and halves the chain of tile iterations.
template <int TK, int KS, bool WRITE_D>
__device__ __forceinline__ void attn_bwd_dq_body(const AttnBwdP& p) {
  constexpr int HD = 128, DC = 4, DT = 8, NCH = TK / 32, NLD = TK / 16, PV = TK == 64 ? AB_PV : AB_PQ;
  constexpr int TILE_B = TK * AB_PQ + 128 * PV, MERGE_WAVE = DT * 4 * 64 * 4;
  extern __shared__ __attribute__((aligned(16))) char ab_smem[];
  const int grp = threadIdx.x >> 8, tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6, fr = lane & 15, g = lane >> 4;
  // two LDS tile buffers per wave group (r03x): tile i+1 is stored while tile i is still being read, so a tile costs ONE workgroup barrier instead of two --
  // the SQ counters show the waves waiting 43 % of their cycles and no unit above 13 % busy (profiles/r03x_attn_bwd_pmc.md)
  char* const tiles = ab_smem + grp * 2 * TILE_B;
  char* Ks = tiles;                        // [TK keys][128 d], pitch 288
  char* Vs = Ks + TK * AB_PQ;              // [128 d][TK keys], pitch 160 / 288
  const int qb = blockIdx.x, h = blockIdx.y, kvh = h / (p.n_q / p.n_kv);
  const int qi = qb * 64 + wave * 16 + fr;                   // this lane's query (operand column)
  const int qc = min(qi, p.S - 1);
  const size_t qrow = (size_t)qc * p.n_q * HD + (size_t)h * HD;
  const bf16_t* K = p.k + (size_t)kvh * p.s_max * HD;
  const bf16_t* VT = p.vt + (size_t)kvh * HD * p.s_max;
  const bool ab_me = blockIdx.x == gridDim.x - 1;
  int ab_i = 0;
  (void)ab_me; (void)ab_i;
  AB_STAMP(0, ab_i++, ab_me)
  bf16x8 qf[DC], dof[DC];
  float dsum = 0.f;
#pragma unroll
  for (int dc = 0; dc < DC; ++dc) {
    qf[dc] = as_bf16x8(ld_global_16(p.q + qrow + dc * 32 + g * 8));
    dof[dc] = ab_load_tr_order(p.d_o + qrow, dc * 32, g);
    const bf16x8 of = ab_load_tr_order(p.o + qrow, dc * 32, g);
#pragma unroll
    for (int e = 0; e < 8; ++e) dsum += (float)dof[dc][e] * (float)of[e];
  }
  dsum = vl_xor32_sum(vl_xor16_sum(dsum));
  const float D = dsum;
  const float lse = p.lse[(size_t)h * p.S + qc];
  if (WRITE_D && g == 0 && grp == 0 && qi < p.S) p.delta[(size_t)h * p.S + qi] = D;      // (the one-launch form gets D from attn_bwd_delta_kernel: same bits)
  const int klim = qi < p.S ? (p.causal ? min(qi + 1, p.kv_valid) : p.kv_valid) : 0;
  const float sc = p.scale * 1.4426950408889634f;
  f32x4 dqT[DT];
#pragma unroll
  for (int i = 0; i < DT; ++i) dqT[i] = f32x4{0, 0, 0, 0};
  AB_STAMP(0, ab_i++, ab_me)          // prologue values arrived (D reduced)
  const int kmax = p.causal ? min(p.kv_valid, min(p.S, qb * 64 + 64)) : p.kv_valid;     // keys any row of this workgroup sees
  const int n_tiles = (kmax + TK - 1) / TK;
  u32x4 rk[NLD], rv[NLD];
  auto load_tile = [&](int it) {
    const int key0 = it * TK;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int c = tid + i * 256, row = c >> 4, slot = c & 15;                     // K tile: TK rows x 16 chunks
      rk[i] = ld_global_16(K + (size_t)min(key0 + row, p.s_max - 1) * HD + slot * 8);
      const int row2 = c / (TK / 8), slot2 = c % (TK / 8);                          // V^T tile: 128 rows x TK/8 chunks (8 keys each)
      rv[i] = ld_global_16(VT + (size_t)row2 * p.s_max + min(key0 + slot2 * 8, p.s_max - 8));
    }
  };
  auto store_tile = [&](int buf) {
    char* ks = tiles + buf * TILE_B;
    char* vs = ks + TK * AB_PQ;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int c = tid + i * 256;
      *reinterpret_cast<u32x4*>(ks + (c >> 4) * AB_PQ + (c & 15) * 16) = rk[i];
      *reinterpret_cast<u32x4*>(vs + (c / (TK / 8)) * PV + (c % (TK / 8)) * 16) = rv[i];
    }
  };
  const int n_it = (n_tiles + KS - 1) / KS;                   // workgroup-uniform trip count: a group past its last tile idles at the barriers
  if (n_tiles > 0) {
    load_tile(min(grp, n_tiles - 1));
    store_tile(0);
    load_tile(min(grp + KS, n_tiles - 1));
  }
  for (int itg = 0; itg < n_it; ++itg) {
    const int it = itg * KS + grp;
    __syncthreads();                   // tile itg is in its buffer for every wave, and every wave is done reading the other buffer (tile itg-1)
    Ks = tiles + (itg & 1) * TILE_B;
    Vs = Ks + TK * AB_PQ;
    store_tile((itg + 1) & 1);         // tile itg+1 (requested one iteration ago) into the other buffer; then request tile itg+2
    load_tile(min(it + 2 * KS, n_tiles - 1));
    AB_STAMP(0, ab_i++, ab_me)
    if (it >= n_tiles) continue;
    const int key0 = it * TK;
    bf16x8 dsf[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      float dsv[8];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int krow = c * 32 + t * 16 + fr;
        f32x4 s = {0, 0, 0, 0}, dp = {0, 0, 0, 0};
#pragma unroll
        for (int dc = 0; dc < DC; ++dc) {
          const bf16x8 kf = as_bf16x8(*reinterpret_cast<const u32x4*>(Ks + krow * AB_PQ + (dc * 32 + g * 8) * 2));
          s = mfma16(kf, qf[dc], s);
          const bf16x8 vf = ab_tr_frag(Vs, PV, c * 32 + t * 16, dc * 32, g, fr);
          dp = mfma16(vf, dof[dc], dp);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = key0 + c * 32 + t * 16 + g * 4 + r;
          const float pe = key < klim ? __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], sc, -lse)) : 0.f;
          dsv[t * 4 + r] = pe * (dp[r] - D) * p.scale;
        }
      }
      dsf[c] = as_bf16x8(u32x4{pack_bf16x2(dsv[0], dsv[1]), pack_bf16x2(dsv[2], dsv[3]), pack_bf16x2(dsv[4], dsv[5]), pack_bf16x2(dsv[6], dsv[7])});
    }
    AB_STAMP(0, ab_i++, ab_me)        // S^T, dP^T, dS done
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      f32x4 acc = dqT[dt];
#pragma unroll
      for (int c = 0; c < NCH; ++c) acc = mfma16(ab_tr_frag(Ks, AB_PQ, dt * 16, c * 32, g, fr), dsf[c], acc);
      dqT[dt] = acc;
    }
  }
  AB_STAMP(0, ab_i++, ab_me)          // loop done
  if constexpr (KS > 1) {          // sum the groups' dQ: lane for lane (every group holds the same elements in the same registers), fixed order
    __syncthreads();
    if (grp > 0) {
      float* mw = reinterpret_cast<float*>(ab_smem + ((grp - 1) * 4 + wave) * MERGE_WAVE) + lane;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) mw[(dt * 4 + r) * 64] = dqT[dt][r];
    }
    __syncthreads();
    if (grp > 0) return;
#pragma unroll
    for (int j = 1; j < KS; ++j) {
      const float* rw = reinterpret_cast<const float*>(ab_smem + ((j - 1) * 4 + wave) * MERGE_WAVE) + lane;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) dqT[dt][r] += rw[(dt * 4 + r) * 64];
    }
  }
  if (qi < p.S) {
    bf16_t* o = p.dq + qrow;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
      *reinterpret_cast<u32x2*>(o + dt * 16 + g * 4) = u32x2{pack_bf16x2(dqT[dt][0], dqT[dt][1]), pack_bf16x2(dqT[dt][2], dqT[dt][3])};
  }
  AB_STAMP(0, ab_i++, ab_me)
#ifdef AB_TIMELINE
  if (ab_me && threadIdx.x == 0 && blockIdx.y == 0) ab_dbg[0][62] = ab_i;
#endif
}

// ---------------------------------------------------------------------------------------------- dK, dV (one partial per Q head)
template <int TK, int KS>      // TK queries per staged tile, KS wave groups taking the query tiles in turn (as above)
__device__ __forceinline__ void attn_bwd_dkv_body(const AttnBwdP& p) {
  constexpr int HD = 128, DC = 4, DT = 8, NCH = TK / 32, NLD = TK / 16;
  constexpr int TILE_B = 2 * TK * AB_PQ + 2 * TK * 4, MERGE_WAVE = 2 * DT * 4 * 64 * 4;
  extern __shared__ __attribute__((aligned(16))) char ab_smem[];
  const int grp = threadIdx.x >> 8, tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6, fr = lane & 15, g = lane >> 4;
  char* const tiles = ab_smem + grp * 2 * TILE_B;          // two tile buffers per wave group: one barrier per tile (see the dQ kernel)
  char* Qs = tiles;                        // [TK queries][128 d], pitch 288
  char* Os = Qs + TK * AB_PQ;              // dO, same shape
  float* lse_s = reinterpret_cast<float*>(Qs + 2 * TK * AB_PQ);
  float* del_s = lse_s + TK;
  const int kb = blockIdx.x, h = blockIdx.y, kvh = h / (p.n_q / p.n_kv);
  const int key = kb * 64 + wave * 16 + fr;                  // this lane's key (operand column)
  const int keyc = min(key, p.s_max - 1);
  const bool key_ok = key < p.kv_valid && key < p.S;
  const bf16_t* K = p.k + (size_t)kvh * p.s_max * HD;
  const bf16_t* VT = p.vt + (size_t)kvh * HD * p.s_max;
  bf16x8 kf[DC], vf[DC];
#pragma unroll
  for (int dc = 0; dc < DC; ++dc) {
    kf[dc] = as_bf16x8(ld_global_16(K + (size_t)keyc * HD + dc * 32 + g * 8));
    union { bf16_t s[8]; bf16x8 b; } u;
#pragma unroll
    for (int e = 0; e < 8; ++e) u.s[e] = VT[(size_t)(dc * 32 + g * 8 + e) * p.s_max + keyc];
    vf[dc] = u.b;
  }
  const float sc = p.scale * 1.4426950408889634f;
  f32x4 dvT[DT], dkT[DT];
#pragma unroll
  for (int i = 0; i < DT; ++i) { dvT[i] = f32x4{0, 0, 0, 0}; dkT[i] = f32x4{0, 0, 0, 0}; }
  const int n_qt = (p.S + TK - 1) / TK;
  const int it0 = p.causal ? (kb * 64) / TK : 0;              // causal: query tiles left of this key tile see none of its keys
  u32x4 rq[NLD], ro[NLD];
  float rl = 0.f, rd = 0.f;
  auto load_tile = [&](int it) {
    const int q0 = it * TK;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int c = tid + i * 256, row = c >> 4, slot = c & 15;
      const size_t off = (size_t)min(q0 + row, p.S - 1) * p.n_q * HD + (size_t)h * HD + slot * 8;
      rq[i] = ld_global_16(p.q + off);
      ro[i] = ld_global_16(p.d_o + off);
    }
    if (tid < TK) {
      const int qq = min(q0 + tid, p.S - 1);
      rl = p.lse[(size_t)h * p.S + qq];
      rd = p.delta[(size_t)h * p.S + qq];
    }
  };
  auto store_tile = [&](int buf) {
    char* qs = tiles + buf * TILE_B;
    char* os = qs + TK * AB_PQ;
    float* ls = reinterpret_cast<float*>(qs + 2 * TK * AB_PQ);
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int c = tid + i * 256;
      *reinterpret_cast<u32x4*>(qs + (c >> 4) * AB_PQ + (c & 15) * 16) = rq[i];
      *reinterpret_cast<u32x4*>(os + (c >> 4) * AB_PQ + (c & 15) * 16) = ro[i];
    }
    if (tid < TK) { ls[tid] = rl; ls[TK + tid] = rd; }
  };
  const int n_it = (max(n_qt - it0, 0) + KS - 1) / KS;
  if (it0 < n_qt) {
    load_tile(min(it0 + grp, n_qt - 1));
    store_tile(0);
    load_tile(min(it0 + grp + KS, n_qt - 1));
  }
  for (int itg = 0; itg < n_it; ++itg) {
    const int it = it0 + itg * KS + grp;
    __syncthreads();
    Qs = tiles + (itg & 1) * TILE_B;
    Os = Qs + TK * AB_PQ;
    lse_s = reinterpret_cast<float*>(Qs + 2 * TK * AB_PQ);
    del_s = lse_s + TK;
    store_tile((itg + 1) & 1);
    load_tile(min(it + 2 * KS, n_qt - 1));
    if (it >= n_qt) continue;
    const int q0 = it * TK;
    bf16x8 pf[NCH], dsf[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      float pv[8], dsv[8];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int qr = c * 32 + t * 16 + fr;
        f32x4 s = {0, 0, 0, 0}, dp = {0, 0, 0, 0};
#pragma unroll
        for (int dc = 0; dc < DC; ++dc) {
          const bf16x8 qa = as_bf16x8(*reinterpret_cast<const u32x4*>(Qs + qr * AB_PQ + (dc * 32 + g * 8) * 2));
          s = mfma16(qa, kf[dc], s);
          const bf16x8 oa = as_bf16x8(*reinterpret_cast<const u32x4*>(Os + qr * AB_PQ + (dc * 32 + g * 8) * 2));
          dp = mfma16(oa, vf[dc], dp);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ql = c * 32 + t * 16 + g * 4 + r, qq = q0 + ql;
          const bool vis = key_ok && qq < p.S && (!p.causal || key <= qq);
          const float pe = vis ? __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], sc, -lse_s[ql])) : 0.f;
          pv[t * 4 + r] = pe;
          dsv[t * 4 + r] = pe * (dp[r] - del_s[ql]) * p.scale;
        }
      }
      pf[c] = as_bf16x8(u32x4{pack_bf16x2(pv[0], pv[1]), pack_bf16x2(pv[2], pv[3]), pack_bf16x2(pv[4], pv[5]), pack_bf16x2(pv[6], pv[7])});
      dsf[c] = as_bf16x8(u32x4{pack_bf16x2(dsv[0], dsv[1]), pack_bf16x2(dsv[2], dsv[3]), pack_bf16x2(dsv[4], dsv[5]), pack_bf16x2(dsv[6], dsv[7])});
    }
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      f32x4 av = dvT[dt], ak = dkT[dt];
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        av = mfma16(ab_tr_frag(Os, AB_PQ, dt * 16, c * 32, g, fr), pf[c], av);
        ak = mfma16(ab_tr_frag(Qs, AB_PQ, dt * 16, c * 32, g, fr), dsf[c], ak);
      }
      dvT[dt] = av; dkT[dt] = ak;
    }
  }
  if constexpr (KS > 1) {
    __syncthreads();
    if (grp > 0) {
      float* mw = reinterpret_cast<float*>(ab_smem + ((grp - 1) * 4 + wave) * MERGE_WAVE) + lane;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { mw[(dt * 4 + r) * 64] = dvT[dt][r]; mw[(DT * 4 + dt * 4 + r) * 64] = dkT[dt][r]; }
    }
    __syncthreads();
    if (grp > 0) return;
#pragma unroll
    for (int j = 1; j < KS; ++j) {
      const float* rw = reinterpret_cast<const float*>(ab_smem + ((j - 1) * 4 + wave) * MERGE_WAVE) + lane;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { dvT[dt][r] += rw[(dt * 4 + r) * 64]; dkT[dt][r] += rw[(DT * 4 + dt * 4 + r) * 64]; }
    }
  }
  if (key < p.S) {
    const size_t off = (size_t)key * p.n_q * HD + (size_t)h * HD;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      *reinterpret_cast<u32x2*>(p.dv + off + dt * 16 + g * 4) = u32x2{pack_bf16x2(dvT[dt][0], dvT[dt][1]), pack_bf16x2(dvT[dt][2], dvT[dt][3])};
      *reinterpret_cast<u32x2*>(p.dk + off + dt * 16 + g * 4) = u32x2{pack_bf16x2(dkT[dt][0], dkT[dt][1]), pack_bf16x2(dkT[dt][2], dkT[dt][3])};
    }
  }
}

template <int TK, int KS>
__global__ __launch_bounds__(256 * KS) void attn_bwd_dq_kernel(AttnBwdP p) { attn_bwd_dq_body<TK, KS, true>(p); }
template <int TK, int KS>
__global__ __launch_bounds__(256 * KS) void attn_bwd_dkv_kernel(AttnBwdP p) { attn_bwd_dkv_body<TK, KS>(p); }

// r04: both halves in ONE launch.  At S = 560 each of the two kernels is 9 x 12 = 108 workgroups on 256 CUs, one after the other (18 + 22 us per layer); the only thing
// the dK / dV half needed from the dQ half was D_q = <dO_q, O_q>.  A small kernel computes D first (the dQ prologue's own summation order, so both halves see the bits the
// two-launch form produces), then blockIdx.z picks the half: 216 workgroups side by side.
__global__ __launch_bounds__(256) void attn_bwd_delta_kernel(AttnBwdP p) {
  constexpr int HD = 128, DC = 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, g = lane >> 4;
  const int qi = blockIdx.x * 64 + wave * 16 + fr, h = blockIdx.y;
  const int qc = min(qi, p.S - 1);
  const size_t qrow = (size_t)qc * p.n_q * HD + (size_t)h * HD;
  float dsum = 0.f;
#pragma unroll
  for (int dc = 0; dc < DC; ++dc) {
    const bf16x8 dof = ab_load_tr_order(p.d_o + qrow, dc * 32, g);
    const bf16x8 of = ab_load_tr_order(p.o + qrow, dc * 32, g);
#pragma unroll
    for (int e = 0; e < 8; ++e) dsum += (float)dof[e] * (float)of[e];
  }
  dsum = vl_xor32_sum(vl_xor16_sum(dsum));
  if (g == 0 && qi < p.S) p.delta[(size_t)h * p.S + qi] = dsum;
}
template <int TK, int KS>
__global__ __launch_bounds__(256 * KS) void attn_bwd_both_kernel(AttnBwdP p) {
  if (blockIdx.z == 0) attn_bwd_dq_body<TK, KS, false>(p);
  else attn_bwd_dkv_body<TK, KS>(p);
}

extern "C" int vlaser_attn_bwd(const void* q, const void* k, const void* vt, const void* o, const void* d_o, const float* lse, float* delta_ws, void* dq, void* dk,
                               void* dv, int S, int n_q, int n_kv, int s_max, float scale, int causal, int kv_valid, int head_dim, vl_stream_t s) {
  VL_CHECK(q && k && vt && o && d_o && lse && delta_ws && dq && dk && dv, "vlaser_attn_bwd: null pointer");
  VL_CHECK(head_dim == 128, "vlaser_attn_bwd: head_dim %d -- the kernels are built for head_dim 128 (Qwen2.5)", head_dim);
  VL_CHECK(S >= 1 && n_q % n_kv == 0 && s_max % 64 == 0 && S <= s_max && kv_valid >= 1 && kv_valid <= s_max, "vlaser_attn_bwd: bad geometry");
  AttnBwdP p;
  p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.vt = (const bf16_t*)vt; p.o = (const bf16_t*)o; p.d_o = (const bf16_t*)d_o;
  p.lse = lse; p.delta = delta_ws; p.dq = (bf16_t*)dq; p.dk = (bf16_t*)dk; p.dv = (bf16_t*)dv;
  p.S = S; p.n_q = n_q; p.n_kv = n_kv; p.s_max = s_max; p.scale = scale; p.causal = causal; p.kv_valid = kv_valid < S ? kv_valid : (causal ? S : kv_valid);
  const dim3 grid((S + 63) / 64, n_q);
  static const int force_ks = getenv("VLASER_ATTN_BWD_KS") ? atoi(getenv("VLASER_ATTN_BWD_KS")) : 0;       // tuning / A-B
  const int ks = force_ks == 1 || force_ks == 2 ? force_ks : (S > 128 ? 2 : 1);
  static const bool two_launches = getenv("VLASER_ATTN_BWD_TWO_LAUNCHES") && atoi(getenv("VLASER_ATTN_BWD_TWO_LAUNCHES")) == 1;      // A/B: the r03 form
#define AB_LAUNCH(KS_)                                                                                               \
  {                                                                                                                  \
    const int lds_q = 2 * KS_ * (64 * AB_PQ + 128 * AB_PV), lds_kv = 2 * KS_ * (2 * 64 * AB_PQ + 2 * 64 * 4);   /* two tile buffers per wave group */                \
    if (two_launches) {                                                                                              \
      if (int rc = set_max_lds_once(attn_bwd_dq_kernel<64, KS_>, lds_q)) return rc;                                 \
      if (int rc = set_max_lds_once(attn_bwd_dkv_kernel<64, KS_>, lds_kv)) return rc;                               \
      hipLaunchKernelGGL((attn_bwd_dq_kernel<64, KS_>), grid, dim3(256 * KS_), lds_q, (hipStream_t)s, p);           \
      hipLaunchKernelGGL((attn_bwd_dkv_kernel<64, KS_>), grid, dim3(256 * KS_), lds_kv, (hipStream_t)s, p);         \
    } else {                                                                                                         \
      const int lds_b = lds_q > lds_kv ? lds_q : lds_kv;                                                            \
      if (int rc = set_max_lds_once(attn_bwd_both_kernel<64, KS_>, lds_b)) return rc;                               \
      hipLaunchKernelGGL(attn_bwd_delta_kernel, grid, dim3(256), 0, (hipStream_t)s, p);                             \
      hipLaunchKernelGGL((attn_bwd_both_kernel<64, KS_>), dim3(grid.x, grid.y, 2), dim3(256 * KS_), lds_b, (hipStream_t)s, p); \
    }                                                                                                                \
  }
  if (ks == 2) AB_LAUNCH(2) else AB_LAUNCH(1)
#undef AB_LAUNCH
  VL_LAUNCH_CHECK();
  return 0;
}
