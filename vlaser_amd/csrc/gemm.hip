// bf16 MFMA GEMM for gfx950: out = epilogue(A[M,K] @ W[N,K]^T), fp32 accumulate.
//
// Tile BM(M) x 128(N) x 64(K) with BM in {128, 64, 32}, 256 threads = 4 waves (2x2 for BM >= 64, 1x4 for BM = 32);
// BM = 128 gives each wave a 64x64 sub-tile (4x4 MFMA 16x16x32 tiles).  Small-M prefill shapes (M = 385 / 1025
// rows) pick the smaller BM and/or split-K (blockIdx.y, fp32 partial slabs reduced by the fused
// residual+norm kernel in misc.hip) so that >= 256 workgroups exist to fill the chip.  Both operands are K-contiguous (activations [M,K], torch Linear weights [N,K]) so MFMA fragments are
// 16-byte LDS reads.  Operands are issued "swapped" (W fragment as MFMA-A, activation fragment as MFMA-B) so
// that each lane ends up with 4 CONSECUTIVE output columns n of one row m -> 8-byte stores and lane-local
// fused epilogues (bias / GELU / layer-scale residual / SwiGLU / RoPE + KV-cache scatter).
// LDS: 2 x (A 16 KiB + W 16 KiB), 16-byte slots XOR-swizzled by (row & 7) (guide T2) -> <=2-way conflicts on
// ds_read_b128.  Global->register prefetch of tile k+1 overlaps the MFMAs of tile k; one barrier per K-step.
// blockIdx is remapped so each XCD owns a contiguous run of tiles that share the same weight panel (guide T1).

#include <cstdlib>
#include <cstring>
#include <cstdio>
#include "common.h"
#include "../../include/vlaser_hip.h"

#define BN 128
#define BK 64
#ifndef GEMM_PD
#define GEMM_PD 2
#endif

struct GemmP {
  VlaserGemmArgs a;
  int tiles_m, tiles_n;
};

__device__ __forceinline__ int lds_off(int row, int slot) { return row * 128 + ((slot ^ (row & 7)) << 4); }

template <int EPI>
__device__ __forceinline__ void epilogue(const VlaserGemmArgs& a, int m, int n0, f32x4 v, f32x4 v2) {
  // v holds acc for output columns n0..n0+3 of row m. (v2: partner accumulator for SWIGLU / ROPE)
  if (m >= a.M) return;
  if constexpr (EPI == VL_EPI_PARTIAL) {
    if (n0 < a.N) *reinterpret_cast<f32x4*>(a.out_f32 + ((size_t)blockIdx.y * a.M + m) * a.N + n0) = v;
    return;
  }
  if constexpr (EPI == VL_EPI_F32) {
    float* o = reinterpret_cast<float*>(a.out) + (size_t)m * a.ldo + n0;
    if (n0 + 3 < a.N && (a.ldo & 1) == 0 && ((uintptr_t)a.out & 7) == 0) {      // 8-byte stores (the lm_head's N = 151674 rows are only 8-byte aligned)
      *reinterpret_cast<f32x2_t*>(o) = f32x2_t{v[0], v[1]};
      *reinterpret_cast<f32x2_t*>(o + 2) = f32x2_t{v[2], v[3]};
      return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (n0 + j < a.N) o[j] = v[j];
    return;
  }
  if constexpr (EPI == VL_EPI_NONE || EPI == VL_EPI_BIAS || EPI == VL_EPI_BIAS_GELU || EPI == VL_EPI_BIAS_LS_RES ||
                EPI == VL_EPI_RES) {
    float r[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float x = v[j];
      int n = n0 + j;
      if (n < a.N) {
        if constexpr (EPI == VL_EPI_BIAS || EPI == VL_EPI_BIAS_GELU || EPI == VL_EPI_BIAS_LS_RES)
          x += bf16_to_f32(reinterpret_cast<const bf16_t*>(a.bias)[n]);
        if constexpr (EPI == VL_EPI_BIAS_GELU) {
          if (a.aux_out) reinterpret_cast<bf16_t*>(a.aux_out)[(size_t)m * a.ld_aux + n] = f32_to_bf16(x);      // pre-activation kept for GELU's backward
          x = gelu_erf(x);
        }
        if constexpr (EPI == VL_EPI_BIAS_LS_RES)
          x = bf16_to_f32(reinterpret_cast<const bf16_t*>(a.res)[(size_t)m * a.ldo + n]) +
              bf16_to_f32(reinterpret_cast<const bf16_t*>(a.ls)[n]) * x;
        if constexpr (EPI == VL_EPI_RES) x += bf16_to_f32(reinterpret_cast<const bf16_t*>(a.res)[(size_t)m * a.ldo + n]);
      }
      r[j] = x;
    }
    bf16_t* o = reinterpret_cast<bf16_t*>(a.out) + (size_t)m * a.ldo + n0;
    if (n0 + 3 < a.N && (a.ldo & 3) == 0) {
      u32x2 pk = {pack_bf16x2(r[0], r[1]), pack_bf16x2(r[2], r[3])};
      *reinterpret_cast<u32x2*>(o) = pk;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (n0 + j < a.N) o[j] = f32_to_bf16(r[j]);
    }
    return;
  }
  if constexpr (EPI == VL_EPI_SWIGLU) {
    // n0 indexes the packed [gate16|up16] row space; v = gate rows, v2 = up rows; output column = (n0/32)*16 + n0%16
    int no = (n0 >> 5) * 16 + (n0 & 15);
    if (no >= (a.N >> 1)) return;
    bf16_t* o = reinterpret_cast<bf16_t*>(a.out) + (size_t)m * a.ldo + no;
    // HF rounds gate and up to bf16 before the activation (separate Linear outputs)
    float r[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = round_bf16(silu(round_bf16(v[j]))) * round_bf16(v2[j]);
    u32x2 pk = {pack_bf16x2(r[0], r[1]), pack_bf16x2(r[2], r[3])};
    *reinterpret_cast<u32x2*>(o) = pk;
    if (a.aux_out) {                             // training forward: the rounded pre-activations, packed [gate16 | up16] like VL_EPI_NONE's output
      bf16_t* g = reinterpret_cast<bf16_t*>(a.aux_out) + (size_t)m * a.ld_aux + n0;
      *reinterpret_cast<u32x2*>(g) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
      *reinterpret_cast<u32x2*>(g + 16) = u32x2{pack_bf16x2(v2[0], v2[1]), pack_bf16x2(v2[2], v2[3])};
    }
    return;
  }
  if constexpr (EPI == VL_EPI_QKV_ROPE) {
    // n0 = head*128 + p where p is the packed in-head row: p = 32*j + 16*half + r  <->  d = 16*j + r + 64*half.
    // v = rows with half 0 (d), v2 = rows with half 1 (d + 64)
    int head = n0 >> 7;
    int p = n0 & 127;
    int d = ((p >> 5) << 4) + (p & 15);  // p&15 is a multiple of 4; j-th element -> d + j
    int pos = a.pos_ids[m];
    const bf16_t* bias = reinterpret_cast<const bf16_t*>(a.bias);
    float x1[4], x2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      // q/k/v projections are bf16 Linear outputs in the reference: round once after bias
      x1[j] = round_bf16(v[j] + bf16_to_f32(bias[n0 + j]));
      x2[j] = round_bf16(v2[j] + bf16_to_f32(bias[n0 + 16 + j]));
    }
    int b = m / a.tok_per_batch;
    int slot = a.slot_base + (m - b * a.tok_per_batch);
    const int nq = a.n_q_heads, nkv = a.n_kv_heads;
    if (head < nq + nkv) {
      float o1[4], o2[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float c = a.rope_cos[(size_t)pos * 64 + d + j], s = a.rope_sin[(size_t)pos * 64 + d + j];
        o1[j] = x1[j] * c - x2[j] * s;
        o2[j] = x2[j] * c + x1[j] * s;
      }
      bf16_t* dst;
      if (head < nq) {
        dst = reinterpret_cast<bf16_t*>(a.q_out) + (size_t)m * nq * 128 + head * 128;
      } else {
        dst = reinterpret_cast<bf16_t*>(a.k_cache) + (((size_t)b * nkv + (head - nq)) * a.s_max + slot) * 128;
      }
      u32x2 pk1 = {pack_bf16x2(o1[0], o1[1]), pack_bf16x2(o1[2], o1[3])};
      u32x2 pk2 = {pack_bf16x2(o2[0], o2[1]), pack_bf16x2(o2[2], o2[3])};
      *reinterpret_cast<u32x2*>(dst + d) = pk1;
      *reinterpret_cast<u32x2*>(dst + d + 64) = pk2;
    } else {
      bf16_t* vt = reinterpret_cast<bf16_t*>(a.vt_cache) + ((size_t)b * nkv + (head - nq - nkv)) * 128 * a.s_max + slot;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        vt[(size_t)(d + j) * a.s_max] = f32_to_bf16(x1[j]);
        vt[(size_t)(d + 64 + j) * a.s_max] = f32_to_bf16(x2[j]);
      }
    }
    return;
  }
  if constexpr (EPI == VL_EPI_VIT_QKV) {
    // columns: [0,C) q, [C,2C) k, [2C,3C) v with C = heads*64 ("three h d", modeling_intern_vit.py:212,231)
    const int C = a.vit_heads * 64;
    int which = n0 / C;
    int c = n0 - which * C;
    int h = c >> 6, d = c & 63;
    int t = m / a.vit_seq, s = m - t * a.vit_seq;
    const bf16_t* bias = reinterpret_cast<const bf16_t*>(a.bias);
    float r[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      r[j] = round_bf16(v[j] + bf16_to_f32(bias[n0 + j]));
      if (which == 0) r[j] *= a.q_scale;
    }
    if (which < 2) {
      bf16_t* dst = reinterpret_cast<bf16_t*>(which == 0 ? a.vq : a.vk) +
                    (((size_t)t * a.vit_heads + h) * a.vit_seq_pad + s) * 64 + d;
      u32x2 pk = {pack_bf16x2(r[0], r[1]), pack_bf16x2(r[2], r[3])};
      *reinterpret_cast<u32x2*>(dst) = pk;
    } else {
      bf16_t* vt = reinterpret_cast<bf16_t*>(a.vvt) + (((size_t)t * a.vit_heads + h) * 64 + d) * a.vit_seq_pad + s;
#pragma unroll
      for (int j = 0; j < 4; ++j) vt[(size_t)j * a.vit_seq_pad] = f32_to_bf16(r[j]);
    }
    return;
  }
}

// QKV_ROPE for a whole wave tile: position ids, bias vectors and the cos / sin rows are requested as straight-line vector loads on
// clamped rows (the per-fragment version pays two dependent L2 round trips per fragment behind the `m < M` branch: +4.5 us on the
// joint prefill's 11 us qkv GEMM); only the stores are predicated.  Fragment pair (nt, nt+1) = RoPE halves d and d + 64.
template <int MT, int NT>
__device__ __forceinline__ void epilogue_tile_rope(const VlaserGemmArgs& a, int m_w, int n_w, int fr, int fq, f32x4 (&acc)[NT][MT]) {
  const bf16_t* bias = reinterpret_cast<const bf16_t*>(a.bias);
  const bool fast = (n_w + NT * 16 <= a.N) && (((uintptr_t)bias & 7) == 0) && (((uintptr_t)a.rope_cos & 15) == 0) && (((uintptr_t)a.rope_sin & 15) == 0);
  if (!fast) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; nt += 2) epilogue<VL_EPI_QKV_ROPE>(a, m_w + mt * 16 + fr, n_w + nt * 16 + fq * 4, acc[nt][mt], acc[nt + 1][mt]);
    return;
  }
  constexpr int NP = NT / 2;
  u32x2 b1[NP], b2[NP];
  int pos[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) pos[mt] = a.pos_ids[min(m_w + mt * 16 + fr, a.M - 1)];
#pragma unroll
  for (int np = 0; np < NP; ++np) {
    const int n0 = n_w + np * 32 + fq * 4;
    b1[np] = *reinterpret_cast<const u32x2*>(bias + n0);
    b2[np] = *reinterpret_cast<const u32x2*>(bias + n0 + 16);
  }
  const int nq = a.n_q_heads, nkv = a.n_kv_heads;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = m_w + mt * 16 + fr;
    const int mc = min(m, a.M - 1);
    const int b = mc / a.tok_per_batch;
    const int slot = a.slot_base + (mc - b * a.tok_per_batch);
    f32x4 cs[NP], sn[NP];
#pragma unroll
    for (int np = 0; np < NP; ++np) {
      const int p = (n_w + np * 32 + fq * 4) & 127, d = ((p >> 5) << 4) + (p & 15);
      cs[np] = *reinterpret_cast<const f32x4*>(a.rope_cos + (size_t)pos[mt] * 64 + d);
      sn[np] = *reinterpret_cast<const f32x4*>(a.rope_sin + (size_t)pos[mt] * 64 + d);
    }
#pragma unroll
    for (int np = 0; np < NP; ++np) {
      const int n0 = n_w + np * 32 + fq * 4;
      const int head = n0 >> 7, p = n0 & 127, d = ((p >> 5) << 4) + (p & 15);
      float x1[4], x2[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        // q/k/v projections are bf16 Linear outputs in the reference: round once after bias
        x1[j] = round_bf16(acc[2 * np][mt][j] + ((j & 1) ? bf16hi_to_f32(b1[np][j >> 1]) : bf16lo_to_f32(b1[np][j >> 1])));
        x2[j] = round_bf16(acc[2 * np + 1][mt][j] + ((j & 1) ? bf16hi_to_f32(b2[np][j >> 1]) : bf16lo_to_f32(b2[np][j >> 1])));
      }
      if (head < nq + nkv) {
        float o1[4], o2[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          o1[j] = x1[j] * cs[np][j] - x2[j] * sn[np][j];
          o2[j] = x2[j] * cs[np][j] + x1[j] * sn[np][j];
        }
        bf16_t* dst = head < nq ? reinterpret_cast<bf16_t*>(a.q_out) + (size_t)mc * nq * 128 + head * 128
                                : reinterpret_cast<bf16_t*>(a.k_cache) + (((size_t)b * nkv + (head - nq)) * a.s_max + slot) * 128;
        if (m < a.M) {
          *reinterpret_cast<u32x2*>(dst + d) = u32x2{pack_bf16x2(o1[0], o1[1]), pack_bf16x2(o1[2], o1[3])};
          *reinterpret_cast<u32x2*>(dst + d + 64) = u32x2{pack_bf16x2(o2[0], o2[1]), pack_bf16x2(o2[2], o2[3])};
        }
      } else if (m < a.M) {
        bf16_t* vt = reinterpret_cast<bf16_t*>(a.vt_cache) + ((size_t)b * nkv + (head - nq - nkv)) * 128 * a.s_max + slot;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          vt[(size_t)(d + j) * a.s_max] = f32_to_bf16(x1[j]);
          vt[(size_t)(d + 64 + j) * a.s_max] = f32_to_bf16(x2[j]);
        }
      }
    }
  }
}

// VIT_QKV for a whole wave tile: bias vectors up front, branch-free math, predicated stores (see epilogue_tile).
template <int MT, int NT>
__device__ __forceinline__ void epilogue_tile_vit_qkv(const VlaserGemmArgs& a, int m_w, int n_w, int fr, int fq, f32x4 (&acc)[NT][MT]) {
  const bf16_t* bias = reinterpret_cast<const bf16_t*>(a.bias);
  const bool fast = (n_w + NT * 16 <= a.N) && (((uintptr_t)bias & 7) == 0);
  if (!fast) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) epilogue<VL_EPI_VIT_QKV>(a, m_w + mt * 16 + fr, n_w + nt * 16 + fq * 4, acc[nt][mt], acc[nt][mt]);
    return;
  }
  u32x2 bv[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bv[nt] = *reinterpret_cast<const u32x2*>(bias + n_w + nt * 16 + fq * 4);
  const int C = a.vit_heads * 64;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = m_w + mt * 16 + fr;
    const int mc = min(m, a.M - 1);
    const int t = mc / a.vit_seq, sq = mc - t * a.vit_seq;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n0 = n_w + nt * 16 + fq * 4;
      const int which = n0 / C, c = n0 - which * C, h = c >> 6, d = c & 63;
      float r[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        r[j] = round_bf16(acc[nt][mt][j] + ((j & 1) ? bf16hi_to_f32(bv[nt][j >> 1]) : bf16lo_to_f32(bv[nt][j >> 1])));
        if (which == 0) r[j] *= a.q_scale;
      }
      if (m >= a.M) continue;
      if (which < 2) {
        bf16_t* dst = reinterpret_cast<bf16_t*>(which == 0 ? a.vq : a.vk) + (((size_t)t * a.vit_heads + h) * a.vit_seq_pad + sq) * 64 + d;
        *reinterpret_cast<u32x2*>(dst) = u32x2{pack_bf16x2(r[0], r[1]), pack_bf16x2(r[2], r[3])};
      } else {
        bf16_t* vt = reinterpret_cast<bf16_t*>(a.vvt) + (((size_t)t * a.vit_heads + h) * 64 + d) * a.vit_seq_pad + sq;
#pragma unroll
        for (int j = 0; j < 4; ++j) vt[(size_t)j * a.vit_seq_pad] = f32_to_bf16(r[j]);
      }
    }
  }
}

// SWIGLU_BWD for a whole wave tile (NN dgrad of down_proj): acc = d(act); the forward's rounded gate / up pre-activations are read from
// `res` (packed [gate16 | up16], row stride ldo) up front as 8-byte vectors; d(gate) = d u sig (1 + g (1 - sig)), d(up) = d g sig with
// d rounded to bf16 first -- the same expressions, in the same order, as the stand-alone swiglu_bwd kernel this epilogue replaces.
template <int MT, int NT>
__device__ __forceinline__ void epilogue_tile_swiglu_bwd(const VlaserGemmArgs& a, int m_w, int n_w, int fr, int fq, f32x4 (&acc)[NT][MT]) {
  const bf16_t* gu = reinterpret_cast<const bf16_t*>(a.res);
  bf16_t* out = reinterpret_cast<bf16_t*>(a.out);
  const bool in_range = n_w + NT * 16 <= a.N;              // wave-uniform; a ragged right edge takes the guarded path per fragment
  const bool wide = (a.ldo & 7) == 0 && (((uintptr_t)out & 15) == 0);      // 16-byte stores
  // r04: ALL of the wave tile's saved pre-activations are requested before the first is used (MT x NT x 2 eight-byte loads in flight, 64 VGPRs for the
  // 64x64 wave tile) -- one L2 / HBM round trip per workgroup instead of one per 16-row tile (the epilogue was 10.7 of the workgroup's 33.5 us:
  // tools/micro/gemm_timeline.py)
  constexpr int MG = MT < 4 ? MT : ((MT % 4 == 0) ? 4 : (MT % 3 == 0 ? 3 : 2));      // 16-row tiles whose loads fly together (<= 64 VGPRs)
  static_assert(MT % MG == 0, "wave tile height");
#pragma unroll
  for (int mg = 0; mg < MT; mg += MG) {
  u32x2 gv[MT][NT], uv[MT][NT];
#pragma unroll
  for (int mt = mg; mt < mg + MG; ++mt) {
    const size_t row = (size_t)min(m_w + mt * 16 + fr, a.M - 1) * a.ldo;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = min(n_w + nt * 16 + fq * 4, a.N - 4), pg = (n >> 4) * 32 + (n & 15);
      gv[mt][nt] = *reinterpret_cast<const u32x2*>(gu + row + pg);
      uv[mt][nt] = *reinterpret_cast<const u32x2*>(gu + row + pg + 16);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int mt = mg; mt < mg + MG; ++mt) {
    const int m = m_w + mt * 16 + fr;
    const size_t row = (size_t)min(m, a.M - 1) * a.ldo;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = n_w + nt * 16 + fq * 4, pg = (n >> 4) * 32 + (n & 15);
      float dg[4], du[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float g = (j & 1) ? bf16hi_to_f32(gv[mt][nt][j >> 1]) : bf16lo_to_f32(gv[mt][nt][j >> 1]);
        const float u = (j & 1) ? bf16hi_to_f32(uv[mt][nt][j >> 1]) : bf16lo_to_f32(uv[mt][nt][j >> 1]);
        const float d = round_bf16(acc[nt][mt][j]);
        const float sig = 1.0f / (1.0f + __expf(-g));
        dg[j] = d * u * sig * (1.0f + g * (1.0f - sig));
        du[j] = d * g * sig;
      }
      const u32x2 pg_ = u32x2{pack_bf16x2(dg[0], dg[1]), pack_bf16x2(dg[2], dg[3])}, pu_ = u32x2{pack_bf16x2(du[0], du[1]), pack_bf16x2(du[2], du[3])};
      if (in_range && wide) {
        // r04: d(gate) and d(up) of one column tile are neighbours in the packed row ([gate16 | up16]): lane pairs 16 apart trade halves
        // (v_permlane16_swap; both lanes of a pair hold the same row, so the row mask cannot split them) and every lane stores 16 bytes -- one
        // instruction per tile writing 64 contiguous bytes of 16 rows instead of two writing 32
        const u32x2 s0 = __builtin_amdgcn_permlane16_swap(pg_[0], pu_[0], false, false);
        const u32x2 s1 = __builtin_amdgcn_permlane16_swap(pg_[1], pu_[1], false, false);
        if (m < a.M) *reinterpret_cast<u32x4*>(out + row + (n_w + nt * 16) * 2 + (fq & 1) * 16 + (fq >> 1) * 8) = u32x4{s0[0], s1[0], s0[1], s1[1]};
      } else if (m < a.M && (in_range || n + 3 < a.N)) {
        *reinterpret_cast<u32x2*>(out + row + pg) = pg_;
        *reinterpret_cast<u32x2*>(out + row + pg + 16) = pu_;
      }
    }
  }
  }
}

// SwiGLU for a whole wave tile (r04): the same values as epilogue<VL_EPI_SWIGLU> per fragment (HF rounds gate and up to bf16 before the activation), with
// 16-byte stores -- the aux image's [gate16 | up16] neighbours through a lane-pair swap like the backward above, the activation tiles of two neighbouring
// (gate, up) pairs through another (the epilogue was 5.3 of the training forward's 45 us per workgroup: tools/micro/gemm_timeline.py).  Returns false
// (nothing done) when the tile is ragged or an operand unaligned: the caller falls back to the per-fragment path.
template <int MT, int NT>
__device__ __forceinline__ bool epilogue_tile_swiglu(const VlaserGemmArgs& a, int m_w, int n_w, int fr, int fq, f32x4 (&acc)[NT][MT]) {
  bf16_t* out = reinterpret_cast<bf16_t*>(a.out);
  bf16_t* aux = reinterpret_cast<bf16_t*>(a.aux_out);
  const bool ok = n_w + NT * 16 <= a.N && (a.ldo & 7) == 0 && (((uintptr_t)out & 15) == 0) && (!aux || ((a.ld_aux & 7) == 0 && (((uintptr_t)aux & 15) == 0)));
  if (!ok) return false;                                     // wave-uniform
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = m_w + mt * 16 + fr;
    const bool st = m < a.M;
    const size_t mrow = (size_t)min(m, a.M - 1);
    u32x2 actp[NT / 2];
#pragma unroll
    for (int nt = 0; nt < NT; nt += 2) {
      const f32x4 v = acc[nt][mt], v2 = acc[nt + 1][mt];
      float r[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) r[j] = round_bf16(silu(round_bf16(v[j]))) * round_bf16(v2[j]);
      actp[nt / 2] = u32x2{pack_bf16x2(r[0], r[1]), pack_bf16x2(r[2], r[3])};
      if (aux) {
        const u32x2 g_ = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])}, u_ = u32x2{pack_bf16x2(v2[0], v2[1]), pack_bf16x2(v2[2], v2[3])};
        const u32x2 s0 = __builtin_amdgcn_permlane16_swap(g_[0], u_[0], false, false);
        const u32x2 s1 = __builtin_amdgcn_permlane16_swap(g_[1], u_[1], false, false);
        if (st) *reinterpret_cast<u32x4*>(aux + mrow * a.ld_aux + n_w + nt * 16 + (fq & 1) * 16 + (fq >> 1) * 8) = u32x4{s0[0], s1[0], s0[1], s1[1]};
      }
    }
    bf16_t* orow = out + mrow * a.ldo + (n_w >> 1);          // activation column of packed column n: (n / 32) * 16 + n % 16
    if constexpr (NT % 4 == 0) {
#pragma unroll
      for (int pp = 0; pp < NT / 2; pp += 2) {
        const u32x2 s0 = __builtin_amdgcn_permlane16_swap(actp[pp][0], actp[pp + 1][0], false, false);
        const u32x2 s1 = __builtin_amdgcn_permlane16_swap(actp[pp][1], actp[pp + 1][1], false, false);
        if (st) *reinterpret_cast<u32x4*>(orow + (pp + (fq & 1)) * 16 + (fq >> 1) * 8) = u32x4{s0[0], s1[0], s0[1], s1[1]};
      }
    } else {
#pragma unroll
      for (int pp = 0; pp < NT / 2; ++pp)
        if (st) *reinterpret_cast<u32x2*>(orow + pp * 16 + fq * 4) = actp[pp];
    }
  }
  return true;
}

// Whole wave tile at once for the bf16-row epilogues (NONE / BIAS / BIAS_GELU / BIAS_LS_RES / RES).  The per-fragment epilogue
// above sits behind lane-divergent `m < M` / `n < N` branches with its bias / residual loads inside them: hipcc turns every
// fragment into its own basic block with a load and a full `s_waitcnt` (measured on the ViT fc1 shape: the BIAS_GELU epilogue cost
// 7.7 us of a 24.5 us launch, one L2 round trip per fragment).  Here the common case -- the tile's columns all inside N, rows
// clamped instead of branched on -- is straight-line: NT bias / layer-scale vectors and MT x NT residual vectors are requested
// up front as 8-byte loads, the math is branch-free, only the stores are predicated.
// SSQ (r04, the TN weight-gradient form only): also returns this lane's sum of the squares of the bf16 values it STORED -- the gradient norm's
// share of this tile, so that the 3.6 GB gradient buffer is not read back for it (sft.py `_norm_bucket`).
// Row-contiguous stores (r04).  An MFMA output fragment gives a lane 4 consecutive columns of one row: stored directly, a wave instruction writes 16 rows x 32
// bytes -- a quarter of a 128-byte line per row, and MT x NT such instructions per wave.  Priced with the stores predicated off (tools/micro/
// gemm_epilogue_lab.py): 19 of the 54 us of the gate/up weight gradient, 10 of 27 (down), 3 of 15 (ViT fc1).  With `scr` (this wave's private piece of the
// stage ring, free once every wave has left the K loop: the caller barriers first) the wave tile goes through LDS MC x 16 rows at a time and leaves as
// 16 bytes per lane, whole rows of the wave tile per 2 NT lanes: NT x 32 bytes contiguous, a quarter of the instructions for NT = 4.  The row pitch of the
// scratch image (+16 bytes) spreads the 16 rows of a fragment store over distinct banks.
#ifndef GEMM_STORE_MODE
#define GEMM_STORE_MODE 1        // 0 = fragment stores (8 bytes per lane), 1 = rows through LDS (16 bytes per lane, NT x 32 contiguous), 2 = lane-pair swap (16 bytes, 64 contiguous)
#endif
template <int MT, int NT> struct EpiScratch {
  static constexpr int MC = (MT % 4 == 0) ? 4 : (MT % 3 == 0) ? 3 : (MT % 2 == 0) ? 2 : 1;      // 16-row tiles per pass through the scratch image
  static constexpr int ROWB = NT * 32 + 16, LPR = NT * 2, RPI = 64 / LPR;                        // row pitch; lanes per row; rows per 64-lane instruction
  static constexpr int BYTES = MC * 16 * ROWB;
  static_assert(64 % LPR == 0 && (MC * 16) % RPI == 0, "wave tiles of 1 / 2 / 4 / 8 column tiles");
};
template <int EPI, int MT, int NT, bool SSQ = false>
__device__ __forceinline__ float epilogue_tile(const VlaserGemmArgs& a, int m_w, int n_w, int fr, int fq, f32x4 (&acc)[NT][MT], char* scr = nullptr) {
  static_assert(!SSQ || EPI == VL_EPI_NONE, "the sum of squares is taken of the plain bf16 output");
  typedef EpiScratch<MT, NT> ES;
  float ssq = 0.f;
  constexpr bool HAS_BIAS = (EPI == VL_EPI_BIAS || EPI == VL_EPI_BIAS_GELU || EPI == VL_EPI_BIAS_LS_RES);
  constexpr bool HAS_RES = (EPI == VL_EPI_BIAS_LS_RES || EPI == VL_EPI_RES);
  const bf16_t* bias = reinterpret_cast<const bf16_t*>(a.bias);
  const bf16_t* ls = reinterpret_cast<const bf16_t*>(a.ls);
  const bf16_t* res = reinterpret_cast<const bf16_t*>(a.res);
  bf16_t* out = reinterpret_cast<bf16_t*>(a.out);
  bool fast = (n_w + NT * 16 <= a.N) && ((a.ldo & 3) == 0) && (((uintptr_t)out & 7) == 0);
  if constexpr (HAS_BIAS) fast = fast && (((uintptr_t)bias & 7) == 0);
  if constexpr (EPI == VL_EPI_BIAS_LS_RES) fast = fast && (((uintptr_t)ls & 7) == 0);
  if constexpr (HAS_RES) fast = fast && (((uintptr_t)res & 7) == 0);
  if (!fast) {                                             // ragged right edge / unaligned operands: per-fragment path
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        epilogue<EPI>(a, m_w + mt * 16 + fr, n_w + nt * 16 + fq * 4, acc[nt][mt], acc[nt][mt]);
        if constexpr (SSQ) {
          if (m_w + mt * 16 + fr < a.M) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float r = bf16_to_f32(f32_to_bf16(acc[nt][mt][j]));
              if (n_w + nt * 16 + fq * 4 + j < a.N) ssq += r * r;
            }
          }
        }
      }
    return ssq;
  }
  const bool via_lds = GEMM_STORE_MODE == 1 && scr != nullptr && (a.ldo & 7) == 0 && (((uintptr_t)out & 15) == 0);
  u32x2 bv[NT], lv[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = n_w + nt * 16 + fq * 4;
    if constexpr (HAS_BIAS) bv[nt] = *reinterpret_cast<const u32x2*>(bias + n);
    if constexpr (EPI == VL_EPI_BIAS_LS_RES) lv[nt] = *reinterpret_cast<const u32x2*>(ls + n);
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = m_w + mt * 16 + fr;
    const size_t row = (size_t)min(m, a.M - 1) * a.ldo;     // clamped: rows past M compute on the last row and are not stored
    u32x2 rv[NT];
    if constexpr (HAS_RES) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) rv[nt] = *reinterpret_cast<const u32x2*>(res + row + n_w + nt * 16 + fq * 4);
    }
    u32x2 pk[NT], pa[EPI == VL_EPI_BIAS_GELU ? NT : 1];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      float r[4];
      [[maybe_unused]] float z[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float x = acc[nt][mt][j];
        if constexpr (HAS_BIAS) x += (j & 1) ? bf16hi_to_f32(bv[nt][j >> 1]) : bf16lo_to_f32(bv[nt][j >> 1]);
        if constexpr (EPI == VL_EPI_BIAS_GELU) { z[j] = x; x = gelu_erf(x); }
        if constexpr (EPI == VL_EPI_BIAS_LS_RES)
          x = ((j & 1) ? bf16hi_to_f32(rv[nt][j >> 1]) : bf16lo_to_f32(rv[nt][j >> 1])) +
              ((j & 1) ? bf16hi_to_f32(lv[nt][j >> 1]) : bf16lo_to_f32(lv[nt][j >> 1])) * x;
        if constexpr (EPI == VL_EPI_RES) x += (j & 1) ? bf16hi_to_f32(rv[nt][j >> 1]) : bf16lo_to_f32(rv[nt][j >> 1]);
        r[j] = x;
      }
      pk[nt] = u32x2{pack_bf16x2(r[0], r[1]), pack_bf16x2(r[2], r[3])};
      if constexpr (EPI == VL_EPI_BIAS_GELU) pa[nt] = u32x2{pack_bf16x2(z[0], z[1]), pack_bf16x2(z[2], z[3])};
    }
#if GEMM_STORE_MODE == 2
    // lane pairs 16 apart trade halves of two neighbouring column tiles (v_permlane16_swap: odd 16-lane rows of the first operand <-> even rows of the
    // second): afterwards a lane holds 8 consecutive columns (16 bytes) of ONE tile -- tile nt for even fq, nt + 1 for odd fq -- and a wave
    // instruction writes 64 contiguous bytes of each of 16 rows, with no LDS pass and half the store instructions
    if ((a.ldo & 7) == 0 && (((uintptr_t)out & 15) == 0)) {
#ifdef GEMM_LAB_NOSTORE
      const bool st_ok = m < a.M && a.ldo == 12345;
#else
      const bool st_ok = m < a.M;
#endif
#pragma unroll
      for (int nt = 0; nt < NT; nt += 2) {
        const u32x2 s0 = __builtin_amdgcn_permlane16_swap(pk[nt][0], pk[nt + 1][0], false, false);
        const u32x2 s1 = __builtin_amdgcn_permlane16_swap(pk[nt][1], pk[nt + 1][1], false, false);
        if (st_ok) *reinterpret_cast<u32x4*>(out + row + n_w + (nt + (fq & 1)) * 16 + (fq >> 1) * 8) = u32x4{s0[0], s1[0], s0[1], s1[1]};
      }
      if constexpr (SSQ) {
        if (m < a.M) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const float r0 = bf16lo_to_f32(pk[nt][0]), r1 = bf16hi_to_f32(pk[nt][0]), r2 = bf16lo_to_f32(pk[nt][1]), r3 = bf16hi_to_f32(pk[nt][1]);
            ssq += (r0 * r0 + r1 * r1) + (r2 * r2 + r3 * r3);
          }
        }
      }
      if constexpr (EPI == VL_EPI_BIAS_GELU) {
        if (m < a.M && a.aux_out) {
          bf16_t* aux = reinterpret_cast<bf16_t*>(a.aux_out) + (size_t)m * a.ld_aux + n_w + fq * 4;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) *reinterpret_cast<u32x2*>(aux + nt * 16) = pa[nt];
        }
      }
      continue;
    }
#endif
    if (via_lds) {
      char* wrow = scr + ((mt % ES::MC) * 16 + fr) * ES::ROWB + fq * 8;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) *reinterpret_cast<u32x2*>(wrow + nt * 32) = pk[nt];
      if ((mt % ES::MC) == ES::MC - 1) {                   // the pass is complete: whole rows back out, 16 bytes per lane
        const int lane = fq * 16 + fr;
#pragma unroll
        for (int i = 0; i < ES::MC * 16 / ES::RPI; ++i) {
          const int r = i * ES::RPI + lane / ES::LPR, slot = lane % ES::LPR;
          const u32x4 v = *reinterpret_cast<const u32x4*>(scr + r * ES::ROWB + slot * 16);
          const int mr = m_w + (mt - (ES::MC - 1)) * 16 + r;
#ifdef GEMM_LAB_NOSTORE          // lab build only (tools/micro/gemm_epilogue_lab.py): the epilogue's global stores predicated off, to price them
          if (mr < a.M && a.ldo == 12345)
#else
          if (mr < a.M)
#endif
            *reinterpret_cast<u32x4*>(out + (size_t)mr * a.ldo + n_w + slot * 8) = v;
        }
      }
    }
#ifdef GEMM_LAB_NOSTORE
    if (m < a.M && (a.ldo == 12345 || via_lds)) {
#else
    if (m < a.M) {
#endif
      if (!via_lds) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) *reinterpret_cast<u32x2*>(out + row + n_w + nt * 16 + fq * 4) = pk[nt];
      }
      if constexpr (SSQ) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const float r0 = bf16lo_to_f32(pk[nt][0]), r1 = bf16hi_to_f32(pk[nt][0]), r2 = bf16lo_to_f32(pk[nt][1]), r3 = bf16hi_to_f32(pk[nt][1]);
          ssq += (r0 * r0 + r1 * r1) + (r2 * r2 + r3 * r3);
        }
      }
      if constexpr (EPI == VL_EPI_BIAS_GELU) {
        if (a.aux_out) {                                   // SFT forward of the projector: the rounded pre-activations for GELU's backward (aux rows 8-byte aligned)
          bf16_t* aux = reinterpret_cast<bf16_t*>(a.aux_out) + (size_t)m * a.ld_aux + n_w + fq * 4;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) *reinterpret_cast<u32x2*>(aux + nt * 16) = pa[nt];
        }
      }
    }
  }
  return ssq;
}

template <int EPI, int BM>
__global__ __launch_bounds__(256) void gemm_kernel(GemmP p) {
  constexpr int PD = GEMM_PD;                           // K-tiles in flight (register stages)
  constexpr int WR = BM >= 64 ? 2 : 1, WC = 4 / WR;     // wave grid
  constexpr int WTM = BM / WR, WTN = BN / WC;           // wave tile
  constexpr int MT = WTM / 16, NT = WTN / 16;
  constexpr int ACH = BM / 32;                          // 16-byte A chunks per thread per K-step
  constexpr int BUF = BM * 128 + 16384;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][A BM*128 | W 16K]
  const VlaserGemmArgs& a = p.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave / WC, wc = wave % WC;

  // XCD-aware remap: block b runs on XCD b%8; give each XCD a contiguous chunk of the tile list (bijective form)
  const int nwg = gridDim.x;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_m = bid % p.tiles_m, tile_n = bid / p.tiles_m;  // consecutive tiles share the weight panel
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int kc = a.K / (int)gridDim.y;                           // split-K slice of this block
  const int kbase = blockIdx.y * kc;

  const int bz = blockIdx.z;
  const bf16_t* A = reinterpret_cast<const bf16_t*>(a.A) + (size_t)bz * a.a_bs;
  const bf16_t* W = reinterpret_cast<const bf16_t*>(a.W) + (size_t)(a.w_group > 1 ? bz / a.w_group : bz) * a.w_bs;

  // staging map: thread -> (row = tid/8 + 32*i, slot = tid%8).  PD K-tiles are kept in flight in registers: with one
  // workgroup per CU (the small-M shapes of this path) a single tile of look-ahead leaves every K-step waiting a
  // full L2/HBM round trip.  Loads are unconditional (row indices clamped; rows >= M / N only feed outputs the
  // epilogue drops) so hipcc can count them with vmcnt(n) instead of draining to 0 (guide: lesson 1 in DESIGN.md).
  const int srow = tid >> 3, sslot = tid & 7;
  u32x4 ra[PD][ACH], rw[PD][4];
  const bf16_t* pa[ACH];
  const bf16_t* pw[4];
#pragma unroll
  for (int i = 0; i < ACH; ++i) pa[i] = A + (size_t)min(m0 + srow + 32 * i, a.M - 1) * a.lda + kbase + sslot * 8;
#pragma unroll
  for (int i = 0; i < 4; ++i) pw[i] = W + (size_t)min(n0 + srow + 32 * i, a.N - 1) * a.ldw + kbase + sslot * 8;
  const int nk = kc / BK;
  auto load_tile = [&](int kt, int st) {
    const int ko = min(kt, nk - 1) * BK;
#pragma unroll
    for (int i = 0; i < ACH; ++i) ra[st][i] = ld_global_16(pa[i] + ko);
#pragma unroll
    for (int i = 0; i < 4; ++i) rw[st][i] = ld_global_16(pw[i] + ko);
  };
  auto store_tile = [&](int buf, int st) {
    char* base = smem + buf * BUF;
#pragma unroll
    for (int i = 0; i < ACH; ++i) *reinterpret_cast<u32x4*>(base + lds_off(srow + 32 * i, sslot)) = ra[st][i];
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(base + BM * 128 + lds_off(srow + 32 * i, sslot)) = rw[st][i];
  };

  f32x4 acc[NT][MT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0, 0, 0, 0};

#pragma unroll
  for (int st = 0; st < PD; ++st) {          // issue order = consumption order (vmcnt retires in order); pinned so the
    load_tile(st, st);                       // loop-header wait state matches the steady state of the back edge
    __builtin_amdgcn_sched_barrier(0);
  }
  store_tile(0, 0);
  __builtin_amdgcn_sched_barrier(0);
  load_tile(PD, 0);
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();
  const int fr = lane & 15, fq = lane >> 4;
  auto kstep = [&](int kt, int nst) __attribute__((always_inline)) {
    const int buf = kt & 1;
    const char* As = smem + buf * BUF;
    const char* Ws = As + BM * 128;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 fa[MT], fw[NT];
#pragma unroll
      for (int t = 0; t < MT; ++t)
        fa[t] = as_bf16x8(*reinterpret_cast<const u32x4*>(As + lds_off(wr * WTM + t * 16 + fr, ks * 4 + fq)));
#pragma unroll
      for (int t = 0; t < NT; ++t)
        fw[t] = as_bf16x8(*reinterpret_cast<const u32x4*>(Ws + lds_off(wc * WTN + t * 16 + fr, ks * 4 + fq)));
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = mfma16(fw[nt], fa[mt], acc[nt][mt]);
    }
    store_tile(buf ^ 1, nst);              // register stage nst holds tile kt+1 (clamped duplicate past the end)
    __builtin_amdgcn_sched_barrier(0);     // keep each refill where it is written: hipcc otherwise clusters the PD
    load_tile(kt + 1 + PD, nst);           // refills of an unrolled round and the in-flight depth decays 3,2,1
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
  };
  // full rounds carry no conditionals (a branch inside the round makes the waitcnt pass drain vmcnt to 0 at the loop
  // header); the < PD leftover K-steps run after the loop
  int kt0 = 0;
  for (; kt0 + PD <= nk; kt0 += PD) {
#pragma unroll
    for (int u = 0; u < PD; ++u) kstep(kt0 + u, (u + 1) % PD);
  }
#pragma unroll
  for (int u = 0; u < PD - 1; ++u)
    if (kt0 + u < nk) kstep(kt0 + u, (u + 1) % PD);

  // epilogue: lane -> m = ... + (lane&15), n = ... + (lane>>4)*4 + reg
  VlaserGemmArgs ea = a;      // batched: shift the output (bf16 / fp32 / partial) of this batch element
  if (bz > 0) {
    if (ea.out) ea.out = reinterpret_cast<char*>(ea.out) + (size_t)bz * a.o_bs * (EPI == VL_EPI_F32 ? 4 : 2);
    if (ea.out_f32) ea.out_f32 += (size_t)bz * a.o_bs;
    if (ea.res) ea.res = reinterpret_cast<const char*>(ea.res) + (size_t)bz * a.o_bs * 2;
  }
  if constexpr (EPI == VL_EPI_NONE || EPI == VL_EPI_BIAS || EPI == VL_EPI_BIAS_GELU || EPI == VL_EPI_BIAS_LS_RES || EPI == VL_EPI_RES) {
    epilogue_tile<EPI, MT, NT>(ea, m0 + wr * WTM, n0 + wc * WTN, fr, fq, acc);
    return;
  }
  if constexpr (EPI == VL_EPI_QKV_ROPE) {
    epilogue_tile_rope<MT, NT>(ea, m0 + wr * WTM, n0 + wc * WTN, fr, fq, acc);
    return;
  }
  if constexpr (EPI == VL_EPI_VIT_QKV) {
    epilogue_tile_vit_qkv<MT, NT>(ea, m0 + wr * WTM, n0 + wc * WTN, fr, fq, acc);
    return;
  }
  if constexpr (EPI == VL_EPI_SWIGLU_BWD) {
    epilogue_tile_swiglu_bwd<MT, NT>(ea, m0 + wr * WTM, n0 + wc * WTN, fr, fq, acc);
    return;
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = m0 + wr * WTM + mt * 16 + fr;
    if constexpr (EPI == VL_EPI_SWIGLU || EPI == VL_EPI_QKV_ROPE) {
#pragma unroll
      for (int nt = 0; nt < NT; nt += 2) epilogue<EPI>(ea, m, n0 + wc * WTN + nt * 16 + fq * 4, acc[nt][mt], acc[nt + 1][mt]);
    } else {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) epilogue<EPI>(ea, m, n0 + wc * WTN + nt * 16 + fq * 4, acc[nt][mt], acc[nt][mt]);
    }
  }
}

// ---------------------------------------------------------------------------------------------- LDS-DMA pipeline
// Same fragments and epilogues as gemm_kernel, but the operands go global -> LDS directly (global_load_lds_dwordx4: no
// staging VGPRs, no ds_write pass) into a ring of NST stages with NST-1 K-tiles in flight, ONE raw s_barrier per K-step and
// a counted vmcnt (never 0 inside the loop): the L2/HBM latency of a K-tile is covered by the MFMAs of the NST-2 tiles
// before it instead of by a second co-resident workgroup -- the path's GEMMs are single-round grids (M = 384..1025), so
// there is no second workgroup to hide behind.  WM x WN waves (4 or 8) own a BM x BN tile; BN = 256 halves the activation
// re-reads of the wide GEMMs (gate/up, fc1).
// An LDS-DMA wave instruction writes 64 lanes x 16 B to CONSECUTIVE LDS addresses (M0 base + lane*16), i.e. 8 rows x 128 B
// of the row-major tile image; the XOR swizzle of the image is therefore applied on the SOURCE side: lane l of a piece
// fetches logical slot (l&7) ^ (row&7) of row 8*piece + (l>>3).  The DMA is issued from inline asm so that hipcc neither
// counts it nor drains it at its own waits (guide 5.7); ordering is ours: vmcnt(N) -> s_barrier -> ds_read.
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
// M0 is written and read inside ONE statement and NOT saved / restored (r03): hipcc reserves m0 but emits no use of it anywhere in this file (checked
// in the ISA: every `m0` of gemm.s sits between #ASMSTART / #ASMEND), and the save + restore pair was 2 of the 5 issue slots of every 1-KiB piece
// (3.1 SALU per MFMA on the 64x128 tile, profiles/r02t_pmc_gemm.md).
__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_dst) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gsrc), "s"(lds_dst) : "memory");
}
// Lab build (-DGEMM_TIMELINE, tools/micro/gemm_timeline.py): s_memtime stamps of wave 0 of every workgroup of gemm_glds_kernel
#ifdef GEMM_TIMELINE
__device__ long long gemm_dbg[1024 * 40];
#define GEMM_STAMP(i) { __builtin_amdgcn_sched_barrier(0); if (threadIdx.x == 0 && blockIdx.x < 1024 && (i) < 40) gemm_dbg[blockIdx.x * 40 + (i)] = clock64(); __builtin_amdgcn_sched_barrier(0); }
extern "C" int vlaser_gemm_debug_read(long long* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(gemm_dbg), sizeof(long long) * 1024 * 40); }
#else
#define GEMM_STAMP(i)
#endif
#ifndef GLDS_ISSUE_FIRST
#define GLDS_ISSUE_FIRST 0        // 1 = r02 order (refill issued right behind the barrier, in front of the fragment reads)
#endif
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// WKM ("W k-major", the NN form out = A @ B): the second operand is stored [K][N] row-major -- a weight matrix exactly as the
// forward keeps it, seen from its dgrad (dX = dY @ W).  Its K-tile is staged as [64 k][BNT n] (1 KiB pieces = 1024 / (2 BNT)
// whole k-rows; XOR swizzle of the 16-byte slots on the source side again) and the MFMA fragments come out of LDS through the
// transposing `ds_read_tr16_b64`: lane (fr, fq) reads 8 bytes of k-row 8 fq + (fr >> 2) (+4 for the upper half) at columns
// 4 (fr & 3) .. +3 and receives column fr's four k-values, i.e. operand element e <-> k = 8 fq + e, the same order the
// row-major A fragment has.  A 32-lane phase of that read touches k-rows {0-3, 8-11} (+4), 32 bytes each, at the same two
// logical slots: the swizzle key 2 ((k & 3) | ((k >> 3) & 1) << 2) sends them to eight different slot pairs of the 256-byte bank row.
// AKM ("A k-major", r03): the FIRST operand is stored [K][M] row-major as well -- together with WKM the TN form out = At^T @ Wt of the weight gradients
// (dW = dY^T X, contraction over the sequence), both operands staged and read exactly like the WKM operand.  K must be a whole number of 64-row tiles:
// the caller pads the sequence axis (rows K_true..K of At zero, of Wt finite).
// IFIRST: the refill of the stage read in the previous step is requested right behind the barrier, in FRONT of this step's fragment reads (r02 order) instead
// of behind the first half's (r03).  With only two stages and HBM-cold weights (192x256, the training forward's gate/up) the extra half step of lead is
// worth more than the reads it delays: 43.7 -> 38.7 us; every ring of three or more stages loses 5-13 % with it (tools/micro/gemm_epilogue_lab.py).
// PRD (r05, LAB ONLY -- measured slower, see launch<>'s case 2100): PRD extra PRODUCER waves (one per SIMD) issue every LDS-DMA piece; the WM x WN compute waves only pass
// the barrier, read fragments and issue MFMAs.  The idea: with no DMA at all the loop's compute side (fragment reads -> MFMAs, one barrier per K-step, 192x256 tile) runs at
// 0.90 us per K-step = 1.8 PFLOP/s whatever the MFMA shape (tools/micro/mfma_shape_lab.hip, profiles/r05r_mfma_shape_lab.md), the product loop at 1.6.  A producer wave's
// K-step: vmcnt for ITS pieces of tile kt -> barrier -> refill of the stage read in step kt-1.  What it showed: issuing pieces is serial PER WAVE, so fewer issuing waves
// lengthen the step -- the 8 compute waves issuing 6 pieces each in parallel are the better arrangement.
// ASYM (r05): the W operand's ring is ONE STAGE DEEPER than the A operand's (NST + 1 against NST).  A two-stage ring has one tile in flight -- requested at step kt, needed at
// kt + 1 -- and its step lasts as long as that round trip (192x256: 1.6 us against 0.9 us of fragment reads + MFMAs, profiles/r05r_mfma_shape_lab.md); a third full stage does
// not fit 160 KB at these tile sizes, a third stage of the HBM-cold operand alone does (192x256: 2 x 24 + 3 x 32 = 144 KB): the weights get two steps of lead, the activations
// (L2-resident) one.  Issue order per step: A(kt + NST - 1), then W(kt + NST); everything that must have landed at step kt is older than everything that may still fly.
// SPREAD (r05 lab): the refill's pieces are not issued as one burst (behind the barrier or behind the first half's reads -- both waves of a SIMD are in that burst together and
// the MFMA pipe idles for its 6-7 x ~100 issue cycles) but ONE AT A TIME between groups of MFMAs, evenly over the K-step's 2 x MT x NT of them.
// PIPE (r06, tools/micro/kstep_pipe_lab.hip, profiles/r06d_kstep_pipe_lab.md): the fragment reads are software-pipelined across the K-step's barrier.  In the plain loop both waves
// of a SIMD read their fragments together and issue their MFMAs together, so the LDS reads (0.28 us of a 1.08 us resident 192x256 K-step on random data) ADD to the MFMAs.  Here a
// step is [barrier | reads(kt, h0) -> set A | MFMAs(kt-1, h1) from set B | reads(kt, h1) -> set B | MFMAs(kt, h0) from set A]: every read block is issued in front of an MFMA block
// that does not depend on it.  Same accumulation order per accumulator (bit-identical).  The stage read in step kt-1 is still refilled in step kt: its pending reads (set B) are
// retired by an explicit lgkmcnt(0) in front of the barrier -- long complete by then -- so no prefetch depth is lost (r02's version of the idea moved the barrier and lost a stage).
template <int EPI, int BM, int BNT, int WM, int WN, int NST, bool WKM = false, bool AKM = false, bool IFIRST = GLDS_ISSUE_FIRST != 0, int PRD = 0, bool ASYM = false, bool SPREAD = false,
          bool PIPE = false>
__global__ __launch_bounds__((WM * WN + PRD) * 64) void gemm_glds_kernel(GemmP p) {
  static_assert(!(ASYM && PRD) && !(SPREAD && PRD), "one lab at a time");
  static_assert(!PIPE || (PRD == 0 && !IFIRST), "PIPE: compute waves issue the pieces; the refill is spread or issued behind the first read block");
  constexpr int NW = WM * WN;
  constexpr int NWI = PRD ? PRD : NW;                    // waves that issue LDS-DMA pieces
  constexpr int WTM = BM / WM, WTN = BNT / WN;
  constexpr int MT = WTM / 16, NT = WTN / 16;
  constexpr int STAGE = (BM + BNT) * 128;               // bytes: A tile | W tile (row-major, 128 B = 64 k per row)
  constexpr int NPA = BM / 8, NPW = BNT / 8;             // 1 KiB (8-row) pieces of the A / W tile (WKM: 64 k-rows x 2 BNT bytes = the same count)
  constexpr int WROWB = BNT * 2, WRPP = 1024 / WROWB, WSPR = BNT / 8;      // WKM: bytes per k-row, k-rows per piece, 16-byte slots per k-row
  constexpr int AROWB = BM * 2, ARPP = 1024 / AROWB, ASPR = BM / 8;        // AKM: the same for the A tile
  static_assert(!AKM || (WKM && (BM == 128 || BM == 256)), "AKM needs WKM and a power-of-two tile of >= 16 slots per k-row");
  constexpr int PA = (NPA + NWI - 1) / NWI, PW = (NPW + NWI - 1) / NWI; // pieces per issuing wave per K-tile; an uneven split re-issues the last piece
  constexpr int PIECES = PA + PW;                        // (same bytes to the same LDS address: benign) so every wave's vmcnt arithmetic is identical
  static_assert(NPA * 8 == BM && NPW * 8 == BNT, "tile rows must be multiples of 8");
  static_assert(NT % 2 == 0, "fused pair epilogues need an even number of 16-column tiles per wave");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const VlaserGemmArgs& a = p.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WN, wc = wave % WN;
  const bool producer = PRD && wave >= NW;               // (wave-uniform)
  const int iw = PRD ? wave - NW : wave;                 // index among the issuing waves (compute waves of a PRD kernel never use it)
  const int nwg = gridDim.x;
  GEMM_STAMP(0)
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_m = bid % p.tiles_m, tile_n = bid / p.tiles_m;
  const int m0 = tile_m * BM, n0 = tile_n * BNT;
  const int kc = a.K / (int)gridDim.y;
  const int kbase = blockIdx.y * kc;
  const int bz = blockIdx.z;
  const bf16_t* A = reinterpret_cast<const bf16_t*>(a.A) + (size_t)bz * a.a_bs;
  const bf16_t* W = reinterpret_cast<const bf16_t*>(a.W) + (size_t)(a.w_group > 1 ? bz / a.w_group : bz) * a.w_bs;
  const int nk = kc / BK;

  // per-lane source pointers of this wave's pieces (row clamped; the swizzle lives in the source slot)
  const int prow = lane >> 3, pslot = lane & 7;
  const bf16_t* srcA[PA];
  const bf16_t* srcW[PW];
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    if constexpr (AKM) {
      const int krow = min(iw * PA + i, NPA - 1) * ARPP + lane / ASPR;
      const int slot = (lane % ASPR) ^ (2 * ((krow & 3) | (((krow >> 3) & 1) << 2)));
      srcA[i] = A + (size_t)(kbase + krow) * a.lda + min(m0 + slot * 8, ((a.M + 7) & ~7) - 8);
    } else {
      const int row = min(iw * PA + i, NPA - 1) * 8 + prow;
      srcA[i] = A + (size_t)min(m0 + row, a.M - 1) * a.lda + kbase + ((pslot ^ (row & 7)) << 3);
    }
  }
#pragma unroll
  for (int i = 0; i < PW; ++i) {
    if constexpr (WKM) {
      const int krow = min(iw * PW + i, NPW - 1) * WRPP + lane / WSPR;              // k-row of this lane inside the tile
      const int slot = (lane % WSPR) ^ (2 * ((krow & 3) | (((krow >> 3) & 1) << 2)));    // logical 16-byte slot fetched into physical slot lane % WSPR
      srcW[i] = W + (size_t)(kbase + krow) * a.ldw + min(n0 + slot * 8, ((a.N + 7) & ~7) - 8);   // columns past N (chunk-aligned clamp) only feed dropped outputs
    } else {
      const int row = min(iw * PW + i, NPW - 1) * 8 + prow;
      srcW[i] = W + (size_t)min(n0 + row, a.N - 1) * a.ldw + kbase + ((pslot ^ (row & 7)) << 3);
    }
  }
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;      // LDS byte address of the stage ring (low 32 bits of the generic pointer)
  constexpr int WRING = ASYM ? NST * BM * 128 : 0;      // ASYM: [NST A stages][NST + 1 W stages]; else [NST (A | W) stages]
  auto issue_a = [&](int kt, int sa) {
    const int ko = min(kt, nk - 1) * BK;                 // tiles past the end re-fetch the last one (never read)
    const uint32_t base = lds0 + sa * (ASYM ? BM * 128 : STAGE);
#pragma unroll
    for (int i = 0; i < PA; ++i) glds16(srcA[i] + (AKM ? (size_t)ko * a.lda : (size_t)ko), __builtin_amdgcn_readfirstlane(base + min(iw * PA + i, NPA - 1) * 1024));
  };
  auto issue_w = [&](int kt, int sw) {
    const int ko = min(kt, nk - 1) * BK;
    const uint32_t base = ASYM ? lds0 + WRING + sw * (BNT * 128) : lds0 + sw * STAGE + BM * 128;
#pragma unroll
    for (int i = 0; i < PW; ++i) glds16(srcW[i] + (WKM ? (size_t)ko * a.ldw : (size_t)ko), __builtin_amdgcn_readfirstlane(base + min(iw * PW + i, NPW - 1) * 1024));
  };
  auto issue_tile = [&](int kt, int st) { issue_a(kt, st); issue_w(kt, st); };
  // piece j of a refill (A pieces first, then W: the order the vmcnt arithmetic assumes); kta / ktw and sa / sw as issue_a / issue_w take them
  auto issue_piece = [&](int j, int kta, int sa, int ktw, int sw2) {
    if (j < PA) {
      const int ko = min(kta, nk - 1) * BK;
      const uint32_t base = lds0 + sa * (ASYM ? BM * 128 : STAGE);
      glds16(srcA[j] + (AKM ? (size_t)ko * a.lda : (size_t)ko), __builtin_amdgcn_readfirstlane(base + min(iw * PA + j, NPA - 1) * 1024));
    } else {
      const int i = j - PA, ko = min(ktw, nk - 1) * BK;
      const uint32_t base = ASYM ? lds0 + WRING + sw2 * (BNT * 128) : lds0 + sw2 * STAGE + BM * 128;
      glds16(srcW[i] + (WKM ? (size_t)ko * a.ldw : (size_t)ko), __builtin_amdgcn_readfirstlane(base + min(iw * PW + i, NPW - 1) * 1024));
    }
  };

  f32x4 acc[NT][MT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0, 0, 0, 0};

  if (!PRD || producer) {
#pragma unroll
    for (int st = 0; st < NST - 1; ++st) issue_tile(st, st);
    if constexpr (ASYM) issue_w(NST - 1, NST - 1);
  }
  GEMM_STAMP(1)
  const int fr = lane & 15, fq = lane >> 4;
  int st = 0;                                            // stage of tile kt (ASYM: of its A tile)
  [[maybe_unused]] int sw = 0;                           // ASYM: stage of tile kt's W tile (ring of NST + 1)
  if constexpr (PIPE) {
    auto read_frags = [&](int ks, const char* As, const char* Ws, bf16x8 (&fa)[MT], bf16x8 (&fw)[NT]) __attribute__((always_inline)) {
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        if constexpr (AKM) {
          const int kr = ks * 32 + 8 * fq + (fr >> 2), c0 = wr * WTM + t * 16;
          const int sl = (c0 >> 3) + ((fr & 3) >> 1), hb = (fr & 1) * 8;
          const int key_lo = 2 * ((kr & 3) | (((kr >> 3) & 1) << 2)), key_hi = 2 * (((kr + 4) & 3) | ((((kr + 4) >> 3) & 1) << 2));
          typedef __attribute__((address_space(3))) s16x4_t* lds_p;
          const char* plo = As + kr * AROWB + (((sl & ~15) | ((sl & 15) ^ key_lo)) << 4) + hb;
          const char* phi = As + (kr + 4) * AROWB + (((sl & ~15) | ((sl & 15) ^ key_hi)) << 4) + hb;
          union { s16x4_t h[2]; bf16x8 b; } u;
          u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(uint32_t)(uintptr_t)plo);
          u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(uint32_t)(uintptr_t)phi);
          fa[t] = u.b;
        } else {
          fa[t] = as_bf16x8(*reinterpret_cast<const u32x4*>(As + lds_off(wr * WTM + t * 16 + fr, ks * 4 + fq)));
        }
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if constexpr (WKM) {
          const int kr = ks * 32 + 8 * fq + (fr >> 2), c0 = wc * WTN + t * 16;
          const int sl = (c0 >> 3) + ((fr & 3) >> 1), hb = (fr & 1) * 8;
          const int key_lo = 2 * ((kr & 3) | (((kr >> 3) & 1) << 2)), key_hi = 2 * (((kr + 4) & 3) | ((((kr + 4) >> 3) & 1) << 2));
          typedef __attribute__((address_space(3))) s16x4_t* lds_p;
          const char* plo = Ws + kr * WROWB + (((sl & ~15) | ((sl & 15) ^ key_lo)) << 4) + hb;
          const char* phi = Ws + (kr + 4) * WROWB + (((sl & ~15) | ((sl & 15) ^ key_hi)) << 4) + hb;
          union { s16x4_t h[2]; bf16x8 b; } u;
          u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(uint32_t)(uintptr_t)plo);
          u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(uint32_t)(uintptr_t)phi);
          fw[t] = u.b;
        } else {
          fw[t] = as_bf16x8(*reinterpret_cast<const u32x4*>(Ws + lds_off(wc * WTN + t * 16 + fr, ks * 4 + fq)));
        }
      }
    };
    // one block of MT x NT MFMAs; `blk` = 0 / 1: first / second block of the step (the spread refill counts MFMAs over both); kt / stn / swn: the step whose refill rides here
    auto mfma_block = [&](const bf16x8 (&fa)[MT], const bf16x8 (&fw)[NT], int blk, bool refill, int kt, int stn, int swn) __attribute__((always_inline)) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          acc[nt][mt] = mfma16(fw[nt], fa[mt], acc[nt][mt]);
          if constexpr (SPREAD) {
            constexpr int GAP = (2 * MT * NT) / (PIECES + 1) > 0 ? (2 * MT * NT) / (PIECES + 1) : 1;
            const int g = blk * (MT * NT) + nt * MT + mt + 1;
            if (refill && g % GAP == 0 && g / GAP - 1 < PIECES) {
              __builtin_amdgcn_sched_barrier(0);
              issue_piece(g / GAP - 1, kt + NST - 1, stn, ASYM ? kt + NST : kt + NST - 1, ASYM ? swn : stn);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
    };
    auto refill_burst = [&](int kt, int stn, int swn) __attribute__((always_inline)) {
      if constexpr (ASYM) { issue_a(kt + NST - 1, stn); issue_w(kt + NST, swn); }
      else issue_tile(kt + NST - 1, stn);
    };
    bf16x8 faA[MT], fwA[NT], faB[MT], fwB[NT];
    for (int kt = 0; kt < nk; ++kt) {
      if constexpr (ASYM) wait_vmcnt<PA * (NST - 2) + PW * (NST - 1)>();
      else wait_vmcnt<PIECES * (NST - 2)>();               // this wave's pieces of tile kt have landed (younger tiles may fly)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // set B's reads of tile kt-1 (issued a whole MFMA block ago) are retired: stage kt-1 may be refilled behind the barrier
      __builtin_amdgcn_s_barrier();
      GEMM_STAMP(2 + kt)
      int stn = st + NST - 1; if (stn >= NST) stn -= NST;
      [[maybe_unused]] int swn = sw + NST; if (swn >= NST + 1) swn -= NST + 1;
      const char* As = smem + st * (ASYM ? BM * 128 : STAGE);
      const char* Ws = ASYM ? smem + WRING + sw * (BNT * 128) : As + BM * 128;
      read_frags(0, As, Ws, faA, fwA);
      // the refill of this step: spread over the step's two MFMA blocks, or -- burst form, and ALWAYS in step 0, which has no first block -- in one piece behind the first read
      // block, in the order the vmcnt arithmetic assumes (A pieces, then W pieces)
      if (!SPREAD || kt == 0) refill_burst(kt, stn, swn);
      if (kt > 0) mfma_block(faB, fwB, 0, SPREAD, kt, stn, swn);
      read_frags(1, As, Ws, faB, fwB);
      mfma_block(faA, fwA, 1, SPREAD && kt > 0, kt, stn, swn);
      if (++st == NST) st = 0;
      if constexpr (ASYM) { if (++sw == NST + 1) sw = 0; }
    }
    mfma_block(faB, fwB, 0, false, 0, 0, 0);               // (nk - 1, h1)
  } else
  {
  for (int kt = 0; kt < nk; ++kt) {
    if constexpr (ASYM) wait_vmcnt<PA * (NST - 2) + PW * (NST - 1)>();
    else if (!PRD || producer) wait_vmcnt<PIECES * (NST - 2)>();      // this wave's pieces of tile kt have landed (younger tiles may fly)
    __builtin_amdgcn_s_barrier();                        // ... and everyone else's; also: all waves are done reading stage kt-1
    GEMM_STAMP(2 + kt)
    int stn = st + NST - 1; if (stn >= NST) stn -= NST;  // = (kt-1) % NST: the stage read in the previous step
    if constexpr (PRD > 0) {
      if (producer) {
        issue_tile(kt + NST - 1, stn);
        if (++st == NST) st = 0;
        continue;
      }
    }
    const char* As = smem + st * (ASYM ? BM * 128 : STAGE);
    const char* Ws = ASYM ? smem + WRING + sw * (BNT * 128) : As + BM * 128;
    [[maybe_unused]] int swn = sw + NST; if (swn >= NST + 1) swn -= NST + 1;      // ASYM: = (kt - 1) % (NST + 1): the W stage read in the previous step
    auto refill = [&]() {
      if constexpr (ASYM) { issue_a(kt + NST - 1, stn); issue_w(kt + NST, swn); }
      else issue_tile(kt + NST - 1, stn);
    };
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 fa[MT], fw[NT];
      // (r03) the refill of stage kt-1 is issued BEHIND the first half's fragment reads: an LDS-DMA piece costs ~100-185 issue cycles
      // (MI355X_MICROARCH.md), and in front of the reads the whole K-step sat behind PIECES of them before its first ds_read went out
      if constexpr (PRD == 0 && !SPREAD) {
        if (ks == 1 && !IFIRST) refill();
        if (ks == 0 && IFIRST) refill();
      }
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        if constexpr (AKM) {
          const int kr = ks * 32 + 8 * fq + (fr >> 2), c0 = wr * WTM + t * 16;
          const int sl = (c0 >> 3) + ((fr & 3) >> 1), hb = (fr & 1) * 8;
          const int key_lo = 2 * ((kr & 3) | (((kr >> 3) & 1) << 2)), key_hi = 2 * (((kr + 4) & 3) | ((((kr + 4) >> 3) & 1) << 2));
          typedef __attribute__((address_space(3))) s16x4_t* lds_p;
          const char* plo = As + kr * AROWB + (((sl & ~15) | ((sl & 15) ^ key_lo)) << 4) + hb;
          const char* phi = As + (kr + 4) * AROWB + (((sl & ~15) | ((sl & 15) ^ key_hi)) << 4) + hb;
          union { s16x4_t h[2]; bf16x8 b; } u;
          u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(uint32_t)(uintptr_t)plo);
          u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(uint32_t)(uintptr_t)phi);
          fa[t] = u.b;
        } else {
          fa[t] = as_bf16x8(*reinterpret_cast<const u32x4*>(As + lds_off(wr * WTM + t * 16 + fr, ks * 4 + fq)));
        }
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if constexpr (WKM) {
          const int kr = ks * 32 + 8 * fq + (fr >> 2), c0 = wc * WTN + t * 16;      // lower half: k-rows kr, upper: kr + 4 (same swizzle bits 0-1, 3)
          const int sl = (c0 >> 3) + ((fr & 3) >> 1), hb = (fr & 1) * 8;
          const int key_lo = 2 * ((kr & 3) | (((kr >> 3) & 1) << 2)), key_hi = 2 * (((kr + 4) & 3) | ((((kr + 4) >> 3) & 1) << 2));
          typedef __attribute__((address_space(3))) s16x4_t* lds_p;
          const char* plo = Ws + kr * WROWB + (((sl & ~15) | ((sl & 15) ^ key_lo)) << 4) + hb;
          const char* phi = Ws + (kr + 4) * WROWB + (((sl & ~15) | ((sl & 15) ^ key_hi)) << 4) + hb;
          union { s16x4_t h[2]; bf16x8 b; } u;
          u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(uint32_t)(uintptr_t)plo);
          u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(uint32_t)(uintptr_t)phi);
          fw[t] = u.b;
        } else {
          fw[t] = as_bf16x8(*reinterpret_cast<const u32x4*>(Ws + lds_off(wc * WTN + t * 16 + fr, ks * 4 + fq)));
        }
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          acc[nt][mt] = mfma16(fw[nt], fa[mt], acc[nt][mt]);
          if constexpr (SPREAD) {
            constexpr int GAP = (2 * MT * NT) / (PIECES + 1) > 0 ? (2 * MT * NT) / (PIECES + 1) : 1;
            const int g = ks * (MT * NT) + nt * MT + mt + 1;          // MFMAs of this K-step issued so far (a compile-time value after unrolling)
            if (g % GAP == 0 && g / GAP - 1 < PIECES) {
              __builtin_amdgcn_sched_barrier(0);
              issue_piece(g / GAP - 1, kt + NST - 1, stn, ASYM ? kt + NST : kt + NST - 1, ASYM ? swn : stn);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
    }
    if (++st == NST) st = 0;
    if constexpr (ASYM) { if (++sw == NST + 1) sw = 0; }
  }
  }
  GEMM_STAMP(36)
  wait_vmcnt<0>();                                       // drain the clamped look-ahead tiles before the block retires
  if constexpr (PRD > 0) { if (producer) return; }       // (a terminated wave leaves the barrier's count: the epilogue's barrier below completes once every producer has drained and left)

  VlaserGemmArgs ea = a;
  if (bz > 0) {
    if (ea.out) ea.out = reinterpret_cast<char*>(ea.out) + (size_t)bz * a.o_bs * (EPI == VL_EPI_F32 ? 4 : 2);
    if (ea.out_f32) ea.out_f32 += (size_t)bz * a.o_bs;
    if (ea.res) ea.res = reinterpret_cast<const char*>(ea.res) + (size_t)bz * a.o_bs * 2;
  }
  [[maybe_unused]] char* scr = nullptr;                // this wave's scratch for the row-contiguous stores: the stage ring, once every wave is out of the K loop
  if constexpr (EPI == VL_EPI_NONE || EPI == VL_EPI_BIAS || EPI == VL_EPI_BIAS_GELU || EPI == VL_EPI_BIAS_LS_RES || EPI == VL_EPI_RES) {
    static_assert(NW * EpiScratch<MT, NT>::BYTES <= NST * STAGE + (ASYM ? BNT * 128 : 0), "epilogue scratch exceeds the stage ring");
    __builtin_amdgcn_s_barrier();                        // (each wave drained its own LDS-DMA pieces above: nothing lands in the ring any more)
    scr = smem + wave * EpiScratch<MT, NT>::BYTES;
  }
  if constexpr (EPI == VL_EPI_NONE && AKM) {           // the weight-gradient form: optionally this wave's share of the gradient norm (one slot per wave,
    GEMM_STAMP(37)
    const float ssq = epilogue_tile<EPI, MT, NT, true>(ea, m0 + wr * WTM, n0 + wc * WTN, fr, fq, acc, scr);      // no atomics: the caller sums the slots in order)
    if (a.sumsq_part) {
      const float w = wave_sum(ssq);
      if (lane == 0) a.sumsq_part[(size_t)(blockIdx.x + gridDim.x * blockIdx.z) * NW + wave] = w;
    }
    GEMM_STAMP(38)
    return;
  }
  if constexpr (EPI == VL_EPI_NONE || EPI == VL_EPI_BIAS || EPI == VL_EPI_BIAS_GELU || EPI == VL_EPI_BIAS_LS_RES || EPI == VL_EPI_RES) {
    GEMM_STAMP(37)
    epilogue_tile<EPI, MT, NT>(ea, m0 + wr * WTM, n0 + wc * WTN, fr, fq, acc, scr);
    GEMM_STAMP(38)
    return;
  }
  if constexpr (EPI == VL_EPI_QKV_ROPE) {
    GEMM_STAMP(37)
    epilogue_tile_rope<MT, NT>(ea, m0 + wr * WTM, n0 + wc * WTN, fr, fq, acc);
    GEMM_STAMP(38)
    return;
  }
  if constexpr (EPI == VL_EPI_VIT_QKV) {
    GEMM_STAMP(37)
    epilogue_tile_vit_qkv<MT, NT>(ea, m0 + wr * WTM, n0 + wc * WTN, fr, fq, acc);
    GEMM_STAMP(38)
    return;
  }
  if constexpr (EPI == VL_EPI_SWIGLU_BWD) {
    GEMM_STAMP(37)
    epilogue_tile_swiglu_bwd<MT, NT>(ea, m0 + wr * WTM, n0 + wc * WTN, fr, fq, acc);
    GEMM_STAMP(38)
    return;
  }
  GEMM_STAMP(37)
  if constexpr (EPI == VL_EPI_SWIGLU) {
    if (epilogue_tile_swiglu<MT, NT>(ea, m0 + wr * WTM, n0 + wc * WTN, fr, fq, acc)) {
      GEMM_STAMP(38)
      return;
    }
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = m0 + wr * WTM + mt * 16 + fr;
    if constexpr (EPI == VL_EPI_SWIGLU || EPI == VL_EPI_QKV_ROPE) {
#pragma unroll
      for (int nt = 0; nt < NT; nt += 2) epilogue<EPI>(ea, m, n0 + wc * WTN + nt * 16 + fq * 4, acc[nt][mt], acc[nt + 1][mt]);
    } else {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) epilogue<EPI>(ea, m, n0 + wc * WTN + nt * 16 + fq * 4, acc[nt][mt], acc[nt][mt]);
    }
  }
  GEMM_STAMP(38)
}

// ---------------------------------------------------------------------------------------------- TN form, two staggered wave groups (r04)
// The eight waves of gemm_glds_kernel read their fragments together and issue their MFMAs together: the MFMA pipe sits idle while both waves of a SIMD
// are in their LDS-read phase, and 36-51 % of the wave time is an issue stall behind the sibling (profiles/r04r_pmc_gemm.md: 24 % of the pipe busy on
// the weight gradients).  Here waves 0-3 and waves 4-7 (one of each per SIMD) run ONE BARRIER APART: a phase is [LDS-DMA issue + fragment reads | barrier |
// MFMAs | barrier], so while one group issues the MFMAs of phase p the other reads the fragments of phase p (or p + 1) -- the CDNA guide's two-group
// schedule.  Both operands are k-major, so a phase is simply 32 k-rows of each; FOUR such buffers (4 x 32 KB for 256x256) form the ring:
//   phase p:  issue tile p+2 into buffer (p+2) % 4  -- last read in phase p-2, by the late group one barrier interval later, retired (lgkmcnt) before
//                                                      this interval began: three buffers would let the DMA land under those reads
//             read tile p (buffer p % 4)            -- made visible by the wait of phase p-1 + one barrier for the early group, + two for the late one
//             vmcnt(pieces of ONE tile)             -- tile p+1 has landed (tile p+2 may still fly): read in the NEXT phase, never in this one
// Every wave executes the same number of barriers: the late group takes one extra in front of the loop, the early group one behind it.
template <int BM, int BNT>
__global__ __launch_bounds__(512) void gemm_tn_stag_kernel(GemmP p) {
  constexpr int WM = 2, WN = 4, NW = 8, KT = 32, NB = 4;
  constexpr int WTM = BM / WM, WTN = BNT / WN;
  constexpr int MT = WTM / 16, NT = WTN / 16;
  constexpr int ABYTES = BM * 2 * KT, BUF = (BM + BNT) * 2 * KT;
  constexpr int NPA = ABYTES / 1024, NPW = BNT * 2 * KT / 1024;
  constexpr int WROWB = BNT * 2, WRPP = 1024 / WROWB, WSPR = BNT / 8;
  constexpr int AROWB = BM * 2, ARPP = 1024 / AROWB, ASPR = BM / 8;
  constexpr int PA = (NPA + NW - 1) / NW, PW = (NPW + NW - 1) / NW, PIECES = PA + PW;
  static_assert((BM == 128 || BM == 256) && (BNT == 128 || BNT == 256), "power-of-two tiles of >= 16 slots per staged k-row");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const VlaserGemmArgs& a = p.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WN, wc = wave % WN;
  const int nwg = gridDim.x;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_m = bid % p.tiles_m, tile_n = bid / p.tiles_m;
  const int m0 = tile_m * BM, n0 = tile_n * BNT;
  const bf16_t* A = reinterpret_cast<const bf16_t*>(a.A);
  const bf16_t* W = reinterpret_cast<const bf16_t*>(a.W);
  const int nk = a.K / KT;
  const bf16_t* srcA[PA];
  const bf16_t* srcW[PW];
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int krow = min(wave * PA + i, NPA - 1) * ARPP + lane / ASPR;
    const int slot = (lane % ASPR) ^ (2 * ((krow & 3) | (((krow >> 3) & 1) << 2)));
    srcA[i] = A + (size_t)krow * a.lda + min(m0 + slot * 8, ((a.M + 7) & ~7) - 8);
  }
#pragma unroll
  for (int i = 0; i < PW; ++i) {
    const int krow = min(wave * PW + i, NPW - 1) * WRPP + lane / WSPR;
    const int slot = (lane % WSPR) ^ (2 * ((krow & 3) | (((krow >> 3) & 1) << 2)));
    srcW[i] = W + (size_t)krow * a.ldw + min(n0 + slot * 8, ((a.N + 7) & ~7) - 8);
  }
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  auto issue_tile = [&](int kt, int b) {
    const int ko = min(kt, nk - 1) * KT;                 // tiles past the end re-fetch the last one (never read)
    const uint32_t base = lds0 + b * BUF;
#pragma unroll
    for (int i = 0; i < PA; ++i) glds16(srcA[i] + (size_t)ko * a.lda, __builtin_amdgcn_readfirstlane(base + min(wave * PA + i, NPA - 1) * 1024));
#pragma unroll
    for (int i = 0; i < PW; ++i) glds16(srcW[i] + (size_t)ko * a.ldw, __builtin_amdgcn_readfirstlane(base + ABYTES + min(wave * PW + i, NPW - 1) * 1024));
  };
  f32x4 acc[NT][MT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
  const int fr = lane & 15, fq = lane >> 4;
  // fragment addresses inside a buffer (k-row kr and kr + 4 of the 32 staged; swizzle as in gemm_glds_kernel's k-major operands)
  const int kr = 8 * fq + (fr >> 2), hb = (fr & 1) * 8;
  const int key_lo = 2 * ((kr & 3) | (((kr >> 3) & 1) << 2)), key_hi = 2 * (((kr + 4) & 3) | ((((kr + 4) >> 3) & 1) << 2));
  typedef __attribute__((address_space(3))) s16x4_t* lds_p;
  issue_tile(0, 0);
  issue_tile(1, 1);
  wait_vmcnt<PIECES>();                                  // tile 0 has landed
  __builtin_amdgcn_s_barrier();
  const bool late = wr == 1;                             // waves 4-7: one barrier interval behind waves 0-3
  if (late) __builtin_amdgcn_s_barrier();
  int b = 0;
  for (int ph = 0; ph < nk; ++ph) {
    int nb = b + 2; if (nb >= NB) nb -= NB;
    issue_tile(ph + 2, nb);
    __builtin_amdgcn_sched_barrier(0);
    const char* As = smem + b * BUF;
    const char* Ws = As + ABYTES;
    bf16x8 fa[MT], fw[NT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const int c0 = wr * WTM + t * 16, sl = (c0 >> 3) + ((fr & 3) >> 1);
      const char* plo = As + kr * AROWB + (((sl & ~15) | ((sl & 15) ^ key_lo)) << 4) + hb;
      const char* phi = As + (kr + 4) * AROWB + (((sl & ~15) | ((sl & 15) ^ key_hi)) << 4) + hb;
      union { s16x4_t h[2]; bf16x8 v; } u;
      u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(uint32_t)(uintptr_t)plo);
      u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(uint32_t)(uintptr_t)phi);
      fa[t] = u.v;
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int c0 = wc * WTN + t * 16, sl = (c0 >> 3) + ((fr & 3) >> 1);
      const char* plo = Ws + kr * WROWB + (((sl & ~15) | ((sl & 15) ^ key_lo)) << 4) + hb;
      const char* phi = Ws + (kr + 4) * WROWB + (((sl & ~15) | ((sl & 15) ^ key_hi)) << 4) + hb;
      union { s16x4_t h[2]; bf16x8 v; } u;
      u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(uint32_t)(uintptr_t)plo);
      u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(uint32_t)(uintptr_t)phi);
      fw[t] = u.v;
    }
    __builtin_amdgcn_sched_barrier(0);
    wait_vmcnt<PIECES>();                                // tile ph+1 has landed (this wave's pieces); tile ph+2 may still fly
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = mfma16(fw[nt], fa[mt], acc[nt][mt]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    if (++b == NB) b = 0;
  }
  if (!late) __builtin_amdgcn_s_barrier();
  wait_vmcnt<0>();                                       // drain the clamped look-ahead tiles before the block retires
  static_assert(NW * EpiScratch<MT, NT>::BYTES <= NB * BUF, "epilogue scratch exceeds the buffer ring");
  __builtin_amdgcn_s_barrier();                          // every wave is out of the loop and its LDS-DMA pieces have landed: the ring is scratch now
  const float ssq = epilogue_tile<VL_EPI_NONE, MT, NT, true>(a, m0 + wr * WTM, n0 + wc * WTN, fr, fq, acc, smem + wave * EpiScratch<MT, NT>::BYTES);
  if (a.sumsq_part) {
    const float w = wave_sum(ssq);
    if (lane == 0) a.sumsq_part[(size_t)blockIdx.x * NW + wave] = w;
  }
}

template <int BM, int BNT>
static int launch_tn_stag(const VlaserGemmArgs* args, hipStream_t stream) {
  GemmP p;
  p.a = *args;
  p.tiles_m = (args->M + BM - 1) / BM;
  p.tiles_n = (args->N + BNT - 1) / BNT;
  constexpr int lds = 4 * (BM + BNT) * 2 * 32;
  static_assert(lds <= 160 * 1024, "buffer ring exceeds the 160 KiB LDS of a CU");
  VL_CHECK(args->batch <= 1 && args->K % 32 == 0, "vlaser_gemm_tn_lds (staggered configuration): plain (unbatched) products, K a multiple of 32");
  if (args->sumsq_part) {
    const long long need = (long long)p.tiles_m * p.tiles_n * 8;
    VL_CHECK(args->sumsq_cap >= need, "sumsq_part: %d slots given, this launch writes %lld (workgroups x waves)", args->sumsq_cap, need);
  }
  if (int rc = set_max_lds_once(gemm_tn_stag_kernel<BM, BNT>, lds)) return rc;
  hipLaunchKernelGGL((gemm_tn_stag_kernel<BM, BNT>), dim3(p.tiles_m * p.tiles_n), dim3(512), lds, stream, p);
  VL_LAUNCH_CHECK();
  return 0;
}

template <int EPI, int BM, int BNT, int WM, int WN, int NST, bool WKM = false, bool AKM = false, bool IFIRST = GLDS_ISSUE_FIRST != 0, int PRD = 0, bool ASYM = false, bool SPREAD = false,
          bool PIPE = false>
static int launch_glds(const VlaserGemmArgs* args, hipStream_t stream, int splits) {
  GemmP p;
  p.a = *args;
  p.tiles_m = (args->M + BM - 1) / BM;
  p.tiles_n = (args->N + BNT - 1) / BNT;
  constexpr int lds = NST * (BM + BNT) * 128 + (ASYM ? BNT * 128 : 0);
  static_assert(lds <= 160 * 1024, "stage ring exceeds the 160 KiB LDS of a CU");
  if (args->sumsq_part) {
    VL_CHECK(EPI == VL_EPI_NONE && AKM && splits == 1, "sumsq_part: only the TN weight-gradient form (vlaser_gemm_tn_lds) sums its output's squares");
    const long long need = (long long)p.tiles_m * p.tiles_n * (args->batch > 1 ? args->batch : 1) * WM * WN;
    VL_CHECK(args->sumsq_cap >= need, "sumsq_part: %d slots given, this launch writes %lld (workgroups x waves)", args->sumsq_cap, need);
  }
  if (int rc = set_max_lds_once(gemm_glds_kernel<EPI, BM, BNT, WM, WN, NST, WKM, AKM, IFIRST, PRD, ASYM, SPREAD, PIPE>, lds)) return rc;
  hipLaunchKernelGGL((gemm_glds_kernel<EPI, BM, BNT, WM, WN, NST, WKM, AKM, IFIRST, PRD, ASYM, SPREAD, PIPE>), dim3(p.tiles_m * p.tiles_n, splits, args->batch > 1 ? args->batch : 1),
                     dim3((WM * WN + PRD) * 64), lds, stream, p);
  VL_LAUNCH_CHECK();
  return 0;
}

template <int EPI, int BM>
static int launch_bm(const VlaserGemmArgs* args, hipStream_t stream, int splits) {
  GemmP p;
  p.a = *args;
  p.tiles_m = (args->M + BM - 1) / BM;
  p.tiles_n = (args->N + BN - 1) / BN;
  constexpr int lds = 2 * (BM * 128 + 16384);
  if (int rc = set_max_lds_once(gemm_kernel<EPI, BM>, lds)) return rc;
  hipLaunchKernelGGL((gemm_kernel<EPI, BM>), dim3(p.tiles_m * p.tiles_n, splits, args->batch > 1 ? args->batch : 1), dim3(256), lds, stream, p);
  VL_LAUNCH_CHECK();
  return 0;
}

// CUs the tile / split heuristics may count on: 256 on an MI355X of its own.  For a host that runs this library's launches on a CU-MASKED stream (hipExtStreamCreateWithCUMask:
// the only arrangement in which streaming workgroups of another kernel -- RCCL's channels -- did not stretch the single-round GEMM grids, tools/micro/rccl_shadow_lab.py,
// profiles/r05_rccl_contention.md): the grids must be sized for the CUs the mask leaves.  Without a mask a lower budget does not help (measured).  Process-wide, set between
// steps (not while launches of another thread are being queued).
static int g_cu_budget = 256;
extern "C" int vlaser_set_cu_budget(int cus) {
  const int prev = g_cu_budget;
  if (cus >= 64 && cus <= 256) g_cu_budget = cus;
  return prev;
}
extern "C" int vlaser_get_cu_budget(void) { return g_cu_budget; }

// Tile configuration.  Measured on the path's shapes (tools/micro/gemm_lab.cpp, profiles/r02b_gemm_lab.md): the LDS-DMA
// pipelines beat the register-staged kernels whenever the grid is a single round of workgroups (<= one per CU), largest tile
// first losing to smallest: 64x128 (8 waves, 4 stages) for the smallest problems, then 128x128 (4 stages), 128x256 (3 stages),
// 256x256 (2 stages).  Multi-round grids take the configuration with the least modelled time (rounds x tile area / measured
// rate); <= 32 activation rows stay on the 32-row register-staged kernel (a 64-row tile would be 50+ % padding).
template <int EPI, bool WKM = false, bool AKM = false>
static int launch(const VlaserGemmArgs* args, hipStream_t stream) {
  const int splits = (EPI == VL_EPI_PARTIAL && args->k_splits > 1) ? args->k_splits : 1;
  const int tn = (args->N + BN - 1) / BN;
  const int nb = args->batch > 1 ? args->batch : 1;
  auto blocks = [&](int bm, int bn = BN) { return ((args->M + bm - 1) / bm) * ((args->N + bn - 1) / bn) * splits * nb; };
  (void)tn;
  int bm = args->force_bm;
  const int cus = g_cu_budget;         // CUs this launch may count on (vlaser_set_cu_budget): the single-round rule is only as good as this number
  if (bm == 0) {
    if constexpr (AKM) {                 // TN form: 128- and 256-row tiles only (>= 16 slots per staged k-row on both operands)
      // measured at K = 576 (tools/micro/tn_lab.py): qkv / o (192 / 144 tiles of 128x128) 10.2 / 10.0 us; down 1100 / 1200 / 1300 = 33.7 / 29.5 / 26.2 us,
      // gate/up 67.2 / 59.9 / 50.5 us -- multi-round grids want the largest tile
      // r04: grids of more than one round take the staggered two-group kernel (gate/up 46.0 -> 43.8 us; single-round shapes tie: tools/micro/gemm_epilogue_lab.py)
      if (blocks(128, 128) <= cus) bm = 1100;
      else if (blocks(128, 256) <= cus) bm = 1200;
      else {
        static const bool no_stag = getenv("VLASER_TN_NO_STAG") != nullptr;      // diagnostics / same-box A/B: the plain two-stage 256x256 ring for multi-round grids too
        bm = (EPI == VL_EPI_NONE && nb == 1 && blocks(256, 256) > cus && !no_stag) ? 1340 : 1300;
      }
    } else if (args->M <= 32 && !WKM) {
      bm = 32;
    } else if (!WKM && blocks(64, 64) <= cus) {      // (the NN form's transposing reads need >= 16 slots per staged k-row: BNT >= 128)
      bm = 1564;                       // r03: a K-step costs the same ~0.4 us whatever the tile, so the smallest problems want the most workgroups (profiles/r03e_gemm_lab.md)
    } else if (blocks(64, 128) <= cus) {
      bm = 1500;
    } else if (blocks(128, 128) <= cus) {
      bm = 1100;
    } else if (blocks(144, 128) <= cus) {
      bm = 1440;                       // M = 8 x 128 + a few rows (the ViT's 1025 tokens): 144-row tiles cover it in 8 instead of 9 tile rows
    } else if (blocks(128, 256) <= cus) {
      bm = 1200;
    } else if (blocks(192, 256) <= cus && (args->M + 191) / 192 * 192 < (args->M + 255) / 256 * 256) {
      bm = 1900;                       // r03: the SFT step's 560 rows = 3 x 192 (256-row tiles pad 27 %): gate/up forward 53.8 -> 50.5 us (tools/micro/sft_gemm_lab.py)
    } else if (blocks(256, 256) <= cus) {
      bm = 1300;
    } else {
      const struct { int code, bm, bn; float rate; } cand[4] = {{1500, 64, 128, 701.f}, {1100, 128, 128, 850.f}, {1200, 128, 256, 1040.f}, {1300, 256, 256, 1208.f}};
      float best = 1e30f;
      for (const auto& c : cand) {
        const float t = (float)((blocks(c.bm, c.bn) + cus - 1) / cus) * (float)(c.bm * c.bn) / c.rate;
        if (t < best) { best = t; bm = c.code; }
      }
    }
  }
  // LDS-DMA pipeline configurations: 1100 = 128x128 / 4 stages, 1200 = 128x256 / 3, 1300 = 256x256 / 2, 1500 = 64x128 / 4 (8 waves each), 1440 = 144x128 / 4 (6 waves)
  if constexpr (!AKM) {
    static const bool no_asym = getenv("VLASER_GEMM_NO_ASYM") != nullptr;      // diagnostics / same-box A/B: the r03-r04 two-stage rings instead of the asymmetric ones
    if (no_asym) bm = bm == 1900 ? 1901 : bm == 1300 ? 1302 : bm;
  }
  if constexpr (!AKM) {
    // diagnostics / same-box A/B: the refill as one burst per K-step (r02-r04) -- "1" for every configuration, or a list of configuration codes ("1200,1500")
    static const char* no_spread = getenv("VLASER_GEMM_NO_SPREAD");
    if (no_spread) {
      char code[8];
      snprintf(code, sizeof(code), "%d", bm);
      if (!strcmp(no_spread, "1") || strstr(no_spread, code))
        bm = bm == 1100 ? 1101 : bm == 1200 ? 1201 : bm == 1300 ? 1304 : bm == 1440 ? 1441 : bm == 1500 ? 1501 : bm == 1900 ? 1904 : bm;      // (burst forms: plain loops)
    }
  }
  {
    static const bool no_pipe = getenv("VLASER_GEMM_NO_PIPE") != nullptr;      // diagnostics / same-box A/B: the r05 plain K loop on the tiles that pipeline their fragment reads
    if (no_pipe) bm = bm == 1440 ? 1442 : bm == 1500 ? 1502 : bm == 1564 ? 1566 : bm == 1900 ? 1902 : bm;
  }
  if constexpr (AKM) {
    switch (bm) {
      // (r05: neither the asymmetric ring nor the spread refill helps the TN form -- both operands are activations / gradients out of L2, and its 256x256 tile lost 3-6 % with
      // the pieces between the MFMAs: profiles/r05z_spread_lab.md -- so it keeps the r04 rings)
      case 1100: return launch_glds<EPI, 128, 128, 2, 4, 4, true, true>(args, stream, splits);
      case 1110: return launch_glds<EPI, 128, 128, 2, 4, 4, true, true, false, 0, false, false, true>(args, stream, splits);      // lab (r06): PIPE on the TN tiles -- the step moved by 0.05 ms: not the default
      case 1105: return launch_glds<EPI, 128, 128, 2, 4, 5, true, true>(args, stream, splits);
      case 1200: return launch_glds<EPI, 128, 256, 2, 4, 3, true, true>(args, stream, splits);
      case 1210: return launch_glds<EPI, 128, 256, 2, 4, 3, true, true, false, 0, false, false, true>(args, stream, splits);
      case 1300: return launch_glds<EPI, 256, 256, 2, 4, 2, true, true>(args, stream, splits);
      // r04: two staggered wave groups over a ring of four 32-deep buffers (gemm_tn_stag_kernel)
      case 1340: if constexpr (EPI == VL_EPI_NONE) return launch_tn_stag<256, 256>(args, stream); break;
      case 1240: if constexpr (EPI == VL_EPI_NONE) return launch_tn_stag<128, 256>(args, stream); break;
      case 1140: if constexpr (EPI == VL_EPI_NONE) return launch_tn_stag<128, 128>(args, stream); break;
      default: break;
    }
    vlaser_set_error("vlaser_gemm_tn_lds: force_cfg must be 0 or 1100 / 1110 / 1105 / 1200 / 1210 / 1300 / 1140 / 1240 / 1340 (got %d)", bm);
    return -1;
  }
  switch (bm) {
    // r05: the refill's pieces go out one at a time between the K-step's MFMAs (SPREAD: bit-identical, -1 ... -9 % per launch, tools/micro/spread_lab.py); 1101 / 1201 / 1304 / 1441 /
    // 1501 / 1904 = the burst forms of r02-r04 (lab, VLASER_GEMM_NO_SPREAD=1)
    // r06: fragment reads pipelined across the K-step's barrier (PIPE: bit-identical; whole launches -4.1 ... -5.0 % on 144x128, -2.5 ... -5.3 % on 64x128, -6.1 ... -6.3 % on 64x64,
    // -4.3 ... -6.6 % on 192x256, noise on 128x128 / 128x256 which keep the plain loop -- profiles/r06o_pipe_ab.md); 1442 / 1502 / 1566 / 1902 = the r05 defaults for same-box A/B
    // (VLASER_GEMM_NO_PIPE=1)
    case 1100: return launch_glds<EPI, 128, 128, 2, 4, 4, WKM, false, false, 0, false, true>(args, stream, splits);
    case 1110: return launch_glds<EPI, 128, 128, 2, 4, 4, WKM, false, false, 0, false, true, true>(args, stream, splits);      // lab: PIPE on 128x128 (-2.4 ... +3.6 %: noise; not the default)
    case 1101: return launch_glds<EPI, 128, 128, 2, 4, 4, WKM>(args, stream, splits);
    case 1200: return launch_glds<EPI, 128, 256, 2, 4, 3, WKM, false, false, 0, false, true>(args, stream, splits);
    case 1210: return launch_glds<EPI, 128, 256, 2, 4, 3, WKM, false, false, 0, false, true, true>(args, stream, splits);      // lab: PIPE on 128x256 (-2.2 ... +2.0 %: noise; not the default)
    case 1201: return launch_glds<EPI, 128, 256, 2, 4, 3, WKM>(args, stream, splits);
    // r05: the two-stage rings carry a THIRD stage for the W operand alone (ASYM: weights two steps ahead, activations one; 256x256: 2 x 32 + 3 x 32 = 160 KB, 192x256: 144 KB):
    // bit-identical, 3408 x 8192 x 3584 193.9 -> 168.6 us, the SFT forward's gate/up 42.7 -> 37.2 us (tools/micro/asym_ring_lab.py, profiles/r05t_asym_ring_lab.md)
    case 1300: return launch_glds<EPI, 256, 256, 2, 4, 2, WKM, false, false, 0, true, true>(args, stream, splits);
    case 1304: return launch_glds<EPI, 256, 256, 2, 4, 2, WKM, false, false, 0, true>(args, stream, splits);      // lab: asymmetric ring, refill as one burst
    case 1310: if constexpr (!WKM) return launch_glds<EPI, 256, 256, 2, 4, 2, false, false, false, 0, true, true, true>(args, stream, splits); break;      // lab (r06): PIPE on 256x256, NT only (the NN form's transposing reads spill at 256 registers)
    case 1301: return launch_glds<EPI, 256, 256, 2, 4, 2, WKM, false, true>(args, stream, splits);      // lab: the two-stage 256x256 ring with the refill requested first
    case 1302: return launch_glds<EPI, 256, 256, 2, 4, 2, WKM>(args, stream, splits);                   // lab: the r02-r04 two-stage ring
    case 1500: return launch_glds<EPI, 64, 128, 2, 4, 4, WKM, false, false, 0, false, true, true>(args, stream, splits);
    case 1502: return launch_glds<EPI, 64, 128, 2, 4, 4, WKM, false, false, 0, false, true>(args, stream, splits);
    case 1501: return launch_glds<EPI, 64, 128, 2, 4, 4, WKM>(args, stream, splits);
    case 1440: return launch_glds<EPI, 144, 128, 3, 2, 4, WKM, false, false, 0, false, true, true>(args, stream, splits);
    case 1442: return launch_glds<EPI, 144, 128, 3, 2, 4, WKM, false, false, 0, false, true>(args, stream, splits);
    case 1441: return launch_glds<EPI, 144, 128, 3, 2, 4, WKM>(args, stream, splits);
    // r03: deeper rings / smaller tiles for the latency-bound single-round shapes (a K-step of the 64x128 tile takes ~0.38 us with 3 tiles in flight:
    // the LDS-DMA round trip under load is ~1.1 us, so the bytes in flight per CU, not the MFMA pipe, set the rate)
    case 1506: return launch_glds<EPI, 64, 128, 2, 4, 6, WKM>(args, stream, splits);
    case 1105: return launch_glds<EPI, 128, 128, 2, 4, 5, WKM>(args, stream, splits);
    case 1564: if constexpr (!WKM) return launch_glds<EPI, 64, 64, 2, 2, 8, false, false, false, 0, false, false, true>(args, stream, splits); break;      // (burst: the 4-wave 64x64 tile lost 3.5 % with the spread refill; r06: + PIPE, -6.1 %)
    case 1566: if constexpr (!WKM) return launch_glds<EPI, 64, 64, 2, 2, 8, false>(args, stream, splits); break;      // the r03-r05 default (burst, plain loop)
    case 1532: return launch_glds<EPI, 32, 128, 1, 4, 7, WKM>(args, stream, splits);
    case 1900: return launch_glds<EPI, 192, 256, 2, 4, 2, WKM, false, false, 0, true, true, true>(args, stream, splits);      // (r06: + PIPE) r03: 3 tile rows for the SFT step's 560 rows (256-row tiles pad 27 %); r05: ASYM ring
    case 1902: return launch_glds<EPI, 192, 256, 2, 4, 2, WKM, false, false, 0, true, true>(args, stream, splits);      // the r05 default (ASYM + spread, plain loop)
    case 1901: return launch_glds<EPI, 192, 256, 2, 4, 2, WKM, false, true>(args, stream, splits);      // lab: the r03-r04 two-stage ring, refill requested first
    // r05 lab (profiles/r05s_producer_waves.md; its driver left the tree in r06): the 128x128 ring with 4 producer waves (one per SIMD) issuing every LDS-DMA piece: bit-identical,
    // 12-16 % SLOWER (piece issue is serial per wave: 4 waves x 12 pieces take longer than 8 x 6); the 128x256 / 192x256 tiles do not fit 12 waves' 168 registers.  Not used.
    case 2100: return launch_glds<EPI, 128, 128, 2, 4, 4, WKM, false, GLDS_ISSUE_FIRST != 0, 4>(args, stream, splits);
    case 1903: return launch_glds<EPI, 192, 256, 2, 4, 2, WKM, false, true, 0, true>(args, stream, splits);      // lab: ASYM with the refill requested first (41.9 us against 37.2)
    case 1904: return launch_glds<EPI, 192, 256, 2, 4, 2, WKM, false, false, 0, true>(args, stream, splits);      // lab: asymmetric ring, refill as one burst
    default: break;
  }
  if constexpr (WKM) {
    vlaser_set_error("vlaser_gemm_nn: force_bm must be 0 or an LDS-DMA configuration code (got %d)", bm);
    return -1;
  }
  if (bm == 128) return launch_bm<EPI, 128>(args, stream, splits);
  if (bm == 64) return launch_bm<EPI, 64>(args, stream, splits);
  return launch_bm<EPI, 32>(args, stream, splits);
}

// ---------------------------------------------------------------------------------------------- TN GEMM (weight gradients)
// out[M,N] = At^T @ Wt with At [K,M] and Wt [K,N] row-major, i.e. the contraction index is the SLOW axis of both operands:
// dW[n_out, k_in] = sum_s dY[s, n_out] * X[s, k_in] straight from the activations as the forward/backward left them -- no
// transposed copies (they cost 8 launches and ~0.1 ms per layer).  Tiles are staged as they lie in memory ([64 s][cols]
// rows of 16-byte chunks) and the MFMA fragments (8 consecutive s for one column) are gathered by gfx950's transposing
// LDS read: a 16-lane group passes 16 8-byte pieces, piece p = 4 consecutive columns of row p>>2, and lane i receives,
// for j = 0..3, element (i&3) of piece 4j + (i>>2) = T[row j][col i] (probed: tools/micro/tr16_probe.hip).  Row pitches
// of 160 B (64-column tile) / 288 B (128-column tile) put the 8 rows a 32-lane access touches on disjoint banks.
__device__ __forceinline__ bf16x8 tr_frag(const char* lds_tile, int pitch, int col0, int krow0, int fq, int fr) {
  // fragment of 8 k-values for column col0 + fr.  The MFMA only needs A and B to agree on which k a (lane group fq, slot)
  // pair means, so slot (h, j) of group fq is taken as row krow0 + 16 h + 4 fq + j: the two groups of a 32-lane LDS access
  // then read 8 CONSECUTIVE rows, which the 160 B / 288 B pitches spread over disjoint banks (rows r and r + 8 would collide).
  const char* p0 = lds_tile + (krow0 + 4 * fq + (fr >> 2)) * pitch + (col0 + 4 * (fr & 3)) * 2;
  typedef __attribute__((address_space(3))) s16x4_t* lds_p;
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(uint32_t)(uintptr_t)p0);
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(uint32_t)(uintptr_t)(p0 + 16 * pitch));
  union { s16x4_t h[2]; bf16x8 b; } u;
  u.h[0] = lo; u.h[1] = hi;
  return u.b;
}

#ifndef TN_LOAD_LATE
#define TN_LOAD_LATE 0
#endif
struct GemmTnP {
  const bf16_t* At; const bf16_t* Wt; bf16_t* out;
  int M, N, K, ldat, ldwt, ldo, tiles_m, tiles_n;
  // grouped contraction: the K rows are `groups` runs of K rows each (run g starts at At + g*a_gs / Wt + g*w_gs), summed into
  // one output -- dK / dV of grouped-query attention sum over the q heads of a kv group; batch (blockIdx.z) = kv heads
  int groups; long long a_gs, w_gs, a_bs, w_bs, o_bs;
  float* sumsq_part;       // optional: one slot per (workgroup, wave) = the sum of the squares of the bf16 values that wave stored
};

template <int BM>
__global__ __launch_bounds__(256) void gemm_tn_kernel(GemmTnP p) {
  constexpr int WR = 2, WC = 2;
  constexpr int WTM = BM / WR, WTN = BN / WC;
  constexpr int MT = WTM / 16, NT = WTN / 16;
  constexpr int PA = BM * 2 + 32, PW = BN * 2 + 32;      // row pitches (bytes): 160 / 288 -> conflict-free transposing reads
  constexpr int ACH = BM / 32;                           // 16-byte A chunks per thread per K-tile (64 rows x BM/8 chunks / 256)
  constexpr int BUF = 64 * PA + 64 * PW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave / WC, wc = wave % WC;
  const int nwg = gridDim.x;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_m = bid % p.tiles_m, tile_n = bid / p.tiles_m;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int tpg = (p.K + BK - 1) / BK, nk = tpg * p.groups;      // K-tiles per group run, total
  // staging map: A: row = tid / (BM/8) + rows_per_pass * i, chunk = tid % (BM/8); W: row = tid / 16 + 16 * i, chunk = tid % 16
  constexpr int ACPR = BM / 8, ARPP = 256 / ACPR;        // chunks per row, rows per pass
  const int arow = tid / ACPR, achk = tid % ACPR;
  const int wrow = tid >> 4, wchk = tid & 15;
  const bf16_t* pa = p.At + (size_t)blockIdx.z * p.a_bs + min(m0 + achk * 8, ((p.M + 7) & ~7) - 8);   // column clamp (chunk-aligned): columns >= M only feed dropped outputs
  const bf16_t* pw = p.Wt + (size_t)blockIdx.z * p.w_bs + min(n0 + wchk * 8, ((p.N + 7) & ~7) - 8);
  u32x4 ra[2][ACH], rw[2][4];
  auto load_tile = [&](int kt, int st) {
    const int ktc = min(kt, nk - 1);
    const int grp = __builtin_amdgcn_readfirstlane((int)(((float)ktc + 0.5f) * __builtin_amdgcn_rcpf((float)tpg)));   // ktc / tpg (uniform)
    const int k0 = (ktc - grp * tpg) * BK;
    const bf16_t* ga = pa + (size_t)grp * p.a_gs;
    const bf16_t* gw = pw + (size_t)grp * p.w_gs;
#pragma unroll
    for (int i = 0; i < ACH; ++i) {
      const int s = k0 + arow + ARPP * i;
      const u32x4 v = ld_global_16(ga + (size_t)min(s, p.K - 1) * p.ldat);
      ra[st][i] = s < p.K ? v : u32x4{0, 0, 0, 0};               // rows past K must contribute zero (select on data, load unconditional)
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int s = k0 + wrow + 16 * i;
      rw[st][i] = ld_global_16(gw + (size_t)min(s, p.K - 1) * p.ldwt);   // clamped (finite data): the zeroed A rows cancel it
    }
  };
  auto store_tile = [&](int buf, int st) {
    char* base = smem + buf * BUF;
#pragma unroll
    for (int i = 0; i < ACH; ++i) *reinterpret_cast<u32x4*>(base + (arow + ARPP * i) * PA + achk * 16) = ra[st][i];
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(base + 64 * PA + (wrow + 16 * i) * PW + wchk * 16) = rw[st][i];
  };
  f32x4 acc[NT][MT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
  const int fr = lane & 15, fq = lane >> 4;
  load_tile(0, 0);
  load_tile(1, 1);
  store_tile(0, 0);
  __syncthreads();
  auto kstep = [&](int kt, int nst) __attribute__((always_inline)) {
    const int buf = kt & 1;
    const char* As = smem + buf * BUF;
    const char* Ws = As + 64 * PA;
#if !TN_LOAD_LATE
    // (r03) tile kt+2 is requested at the TOP of the step: its register stage (nst ^ 1 held tile kt, which went to LDS before this step) is
    // already free, and the loads then fly under this step's 2 x (fragment reads + MFMAs) instead of starting behind the LDS store of kt+1
    load_tile(kt + 2, nst ^ 1);
    __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 fa[MT], fw[NT];
#pragma unroll
      for (int t = 0; t < MT; ++t) fa[t] = tr_frag(As, PA, wr * WTM + t * 16, ks * 32, fq, fr);
#pragma unroll
      for (int t = 0; t < NT; ++t) fw[t] = tr_frag(Ws, PW, wc * WTN + t * 16, ks * 32, fq, fr);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = mfma16(fw[nt], fa[mt], acc[nt][mt]);
    }
    store_tile(buf ^ 1, nst);
    __builtin_amdgcn_sched_barrier(0);
#if TN_LOAD_LATE
    load_tile(kt + 2, nst ^ 1);            // r02 order: the stage just consumed two steps ago, refilled behind the LDS store
    __builtin_amdgcn_sched_barrier(0);
#endif
    __syncthreads();
  };
  // register stages alternate: tile kt+1 sits in stage (kt+1)&1; after it is stored, stage kt&1 (tile kt, already in LDS) is refilled with kt+2
  int kt = 0;
  for (; kt + 2 <= nk; kt += 2) {
    kstep(kt, 1);
    kstep(kt + 1, 0);
  }
  if (kt < nk) kstep(kt, 1);

  float ssq = 0.f;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = m0 + wr * WTM + mt * 16 + fr;
    if (m >= p.M) continue;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = n0 + wc * WTN + nt * 16 + fq * 4;
      bf16_t* o = p.out + (size_t)blockIdx.z * p.o_bs + (size_t)m * p.ldo + n;
      const f32x4 v = acc[nt][mt];
      if (n + 3 < p.N && (p.ldo & 3) == 0) {
        const u32x2 pk = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        *reinterpret_cast<u32x2*>(o) = pk;
        const float r0 = bf16lo_to_f32(pk[0]), r1 = bf16hi_to_f32(pk[0]), r2 = bf16lo_to_f32(pk[1]), r3 = bf16hi_to_f32(pk[1]);
        ssq += (r0 * r0 + r1 * r1) + (r2 * r2 + r3 * r3);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (n + j < p.N) {
            const bf16_t b = f32_to_bf16(v[j]);
            o[j] = b;
            const float r = bf16_to_f32(b);
            ssq += r * r;
          }
      }
    }
  }
  if (p.sumsq_part) {
    const float w = wave_sum(ssq);
    if (lane == 0) p.sumsq_part[(size_t)(blockIdx.x + gridDim.x * blockIdx.z) * 4 + wave] = w;
  }
}

static int gemm_tn_launch(const void* At, const void* Wt, void* out, int M, int N, int K, int ldat, int ldwt, int ldo, int groups,
                          long long a_gs, long long w_gs, int batch, long long a_bs, long long w_bs, long long o_bs, float* sumsq_part, int sumsq_cap, vl_stream_t s) {
  VL_CHECK(At && Wt && out && M >= 8 && N >= 8 && K > 0 && groups >= 1 && batch >= 1, "vlaser_gemm_tn: bad arguments (M, N >= 8)");
  VL_CHECK(ldat % 8 == 0 && ldwt % 8 == 0 && (((uintptr_t)At | (uintptr_t)Wt) & 15) == 0 && ((a_gs | w_gs | a_bs | w_bs) & 7) == 0,
           "vlaser_gemm_tn: operands must be 16-byte aligned rows / group / batch strides");
  VL_CHECK(ldat >= ((M + 7) & ~7) && ldwt >= ((N + 7) & ~7), "vlaser_gemm_tn: operand rows must be readable up to M, N rounded up to 8 columns");
  GemmTnP p{(const bf16_t*)At, (const bf16_t*)Wt, (bf16_t*)out, M, N, K, ldat, ldwt, ldo, 0, (N + BN - 1) / BN, groups, a_gs, w_gs, a_bs, w_bs, o_bs, sumsq_part};
  hipStream_t stream = reinterpret_cast<hipStream_t>(s);
  constexpr int lds64 = 2 * (64 * (64 * 2 + 32) + 64 * (BN * 2 + 32)), lds128 = 2 * (64 * (128 * 2 + 32) + 64 * (BN * 2 + 32));
  if (int rc = set_max_lds_once(gemm_tn_kernel<64>, lds64)) return rc;
  if (int rc = set_max_lds_once(gemm_tn_kernel<128>, lds128)) return rc;
  const bool big = ((M + 127) / 128) * p.tiles_n * batch >= 512;      // measured: 128-row tiles win on the big weight gradients only
  p.tiles_m = big ? (M + 127) / 128 : (M + 63) / 64;
  VL_CHECK(!sumsq_part || sumsq_cap >= (long long)p.tiles_m * p.tiles_n * batch * 4, "vlaser_gemm_tn: %d sumsq slots given, this launch writes %lld (workgroups x 4 waves)",
           sumsq_cap, (long long)p.tiles_m * p.tiles_n * batch * 4);
  if (big) {
    p.tiles_m = (M + 127) / 128;
    hipLaunchKernelGGL(gemm_tn_kernel<128>, dim3(p.tiles_m * p.tiles_n, 1, batch), dim3(256), lds128, stream, p);
  } else {
    p.tiles_m = (M + 63) / 64;
    hipLaunchKernelGGL(gemm_tn_kernel<64>, dim3(p.tiles_m * p.tiles_n, 1, batch), dim3(256), lds64, stream, p);
  }
  VL_LAUNCH_CHECK();
  return 0;
}

extern "C" int vlaser_gemm_tn_grouped(const void* At, const void* Wt, void* out, int M, int N, int K, int ldat, int ldwt, int ldo, int groups,
                                      long long a_gs, long long w_gs, int batch, long long a_bs, long long w_bs, long long o_bs, vl_stream_t s) {
  return gemm_tn_launch(At, Wt, out, M, N, K, ldat, ldwt, ldo, groups, a_gs, w_gs, batch, a_bs, w_bs, o_bs, nullptr, 0, s);
}

extern "C" int vlaser_gemm_tn(const void* At, const void* Wt, void* out, int M, int N, int K, int ldat, int ldwt, int ldo, float* sumsq_part, int sumsq_cap,
                              vl_stream_t s) {
  return gemm_tn_launch(At, Wt, out, M, N, K, ldat, ldwt, ldo, 1, 0, 0, 1, 0, 0, 0, sumsq_part, sumsq_cap, s);
}

// The TN form on the LDS-DMA pipeline (r03): same result as vlaser_gemm_tn when the contraction axis is PADDED -- K is a whole number of 64-row tiles,
// the rows K_true..K of At are zero and those of Wt finite (the SFT step keeps its activations in buffers of ceil64(S) rows and zeroes the pad rows
// of the dY operands once per backward).  No staging registers, no ds_write pass, 8 waves on 128x128 .. 256x256 tiles.
extern "C" int vlaser_gemm_tn_lds(const void* At, const void* Wt, void* out, int M, int N, int K, int ldat, int ldwt, int ldo, int force_cfg, float* sumsq_part,
                                  int sumsq_cap, vl_stream_t s) {
  VL_CHECK(At && Wt && out && M > 0 && N > 0 && K > 0, "vlaser_gemm_tn_lds: bad args");
  VL_CHECK(K % BK == 0, "vlaser_gemm_tn_lds: K=%d must be a multiple of %d (pad the contraction axis: zero rows in At)", K, BK);
  VL_CHECK(ldat % 8 == 0 && ldwt % 8 == 0 && ldat >= ((M + 7) & ~7) && ldwt >= ((N + 7) & ~7) && ldo >= N,
           "vlaser_gemm_tn_lds: ldat, ldwt must be multiples of 8 and cover M / N rounded up to 8 columns (rows are read in 16-byte pieces)");
  VL_CHECK((((uintptr_t)At | (uintptr_t)Wt) & 15) == 0 && ((uintptr_t)out & 7) == 0, "vlaser_gemm_tn_lds: operands must be 16-byte aligned");
  VlaserGemmArgs a = {};
  a.A = At; a.W = Wt; a.out = out;
  a.M = M; a.N = N; a.K = K; a.lda = ldat; a.ldw = ldwt; a.ldo = ldo;
  a.k_splits = 1; a.force_bm = force_cfg;
  a.sumsq_part = sumsq_part; a.sumsq_cap = sumsq_cap;
  return launch<VL_EPI_NONE, true, true>(&a, reinterpret_cast<hipStream_t>(s));
}

static bool glds_code(int bm) {      // every LDS-DMA configuration code launch<> knows (defaults + lab variants)
  static const int codes[] = {1100, 1101, 1110, 1105, 1200, 1201, 1210, 1310, 1300, 1301, 1302, 1304, 1440, 1441, 1442, 1500, 1501, 1502, 1506, 1532, 1564, 1566, 1900, 1901, 1902, 1903, 1904, 2100};
  for (int c : codes) if (c == bm) return true;
  return false;
}

extern "C" int vlaser_gemm(int epi, const VlaserGemmArgs* a, vl_stream_t s) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(s);
  VL_CHECK(a && a->A && a->W, "vlaser_gemm: null operand");
  VL_CHECK(a->M > 0 && a->N > 0 && a->K > 0, "vlaser_gemm: bad shape M=%d N=%d K=%d", a->M, a->N, a->K);
  VL_CHECK(a->K % BK == 0, "vlaser_gemm: K=%d must be a multiple of %d", a->K, BK);
  VL_CHECK(a->batch <= 1 || (epi == VL_EPI_NONE || epi == VL_EPI_F32 || epi == VL_EPI_BIAS), "vlaser_gemm: batched mode supports NONE / F32 / BIAS epilogues");
  VL_CHECK(!a->sumsq_part, "vlaser_gemm: sumsq_part is honoured by vlaser_gemm_tn_lds only");
  VL_CHECK(a->force_bm == 0 || a->force_bm == 32 || a->force_bm == 64 || a->force_bm == 128 || glds_code(a->force_bm),
           "vlaser_gemm: force_bm must be 0/32/64/128 or an LDS-DMA configuration code 1100/1105/1200/1300/1301/1440/1500/1506/1532/1564/1900, 2100 (lab: producer waves)");
  VL_CHECK(a->lda % 8 == 0 && a->ldw % 8 == 0, "vlaser_gemm: lda/ldw must be multiples of 8 (16-byte rows)");
  VL_CHECK(((uintptr_t)a->A & 15) == 0 && ((uintptr_t)a->W & 15) == 0, "vlaser_gemm: operands must be 16-byte aligned");
  switch (epi) {
    case VL_EPI_NONE: VL_CHECK(a->out, "out null"); return launch<VL_EPI_NONE>(a, stream);
    case VL_EPI_F32: VL_CHECK(a->out, "out null"); return launch<VL_EPI_F32>(a, stream);
    case VL_EPI_PARTIAL:
      VL_CHECK(a->out_f32 && a->N % 4 == 0, "partial: out_f32 null or N %% 4 != 0");
      VL_CHECK(a->k_splits >= 1 && a->K % (a->k_splits * BK) == 0, "partial: K=%d not divisible by k_splits*64 (k_splits=%d)", a->K, a->k_splits);
      return launch<VL_EPI_PARTIAL>(a, stream);
    case VL_EPI_BIAS: VL_CHECK(a->out && a->bias, "out/bias null"); return launch<VL_EPI_BIAS>(a, stream);
    case VL_EPI_BIAS_GELU: VL_CHECK(a->out && a->bias, "out/bias null"); return launch<VL_EPI_BIAS_GELU>(a, stream);
    case VL_EPI_BIAS_LS_RES:
      VL_CHECK(a->out && a->bias && a->res && a->ls, "out/bias/res/ls null");
      return launch<VL_EPI_BIAS_LS_RES>(a, stream);
    case VL_EPI_RES: VL_CHECK(a->out && a->res, "out/res null"); return launch<VL_EPI_RES>(a, stream);
    case VL_EPI_SWIGLU:
      VL_CHECK(a->out && a->N % 32 == 0 && a->ldo % 4 == 0, "swiglu: N must be a multiple of 32, ldo of 4");
      return launch<VL_EPI_SWIGLU>(a, stream);
    case VL_EPI_QKV_ROPE:
      VL_CHECK(a->q_out && a->k_cache && a->vt_cache && a->rope_cos && a->rope_sin && a->pos_ids && a->bias, "qkv_rope: null pointer");
      VL_CHECK(a->N == (a->n_q_heads + 2 * a->n_kv_heads) * 128, "qkv_rope: N mismatch");
      VL_CHECK(a->tok_per_batch > 0 && a->s_max > 0, "qkv_rope: bad cache geometry");
      return launch<VL_EPI_QKV_ROPE>(a, stream);
    case VL_EPI_VIT_QKV:
      VL_CHECK(a->vq && a->vk && a->vvt && a->bias, "vit_qkv: null pointer");
      VL_CHECK(a->N == 3 * a->vit_heads * 64 && a->vit_seq > 0 && a->vit_seq_pad >= a->vit_seq, "vit_qkv: bad geometry");
      return launch<VL_EPI_VIT_QKV>(a, stream);
    default: vlaser_set_error("vlaser_gemm: unknown epilogue %d", epi); return -1;
  }
}

// out[M,N] = A[M,K] @ B[K,N] with B row-major ("k-major": a forward weight [N_out, K_in] read as the B of its own dgrad dX = dY @ W);
// ldw = row stride of B.  NONE (bf16 out), F32 and PARTIAL (split-K fp32 slabs) epilogues; batched like vlaser_gemm (NONE / F32).
extern "C" int vlaser_gemm_nn(int epi, const VlaserGemmArgs* a, vl_stream_t s) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(s);
  VL_CHECK(a && a->A && a->W, "vlaser_gemm_nn: null operand");
  VL_CHECK(a->M > 0 && a->N > 0 && a->K > 0 && a->K % BK == 0, "vlaser_gemm_nn: bad shape M=%d N=%d K=%d (K must be a multiple of %d)", a->M, a->N, a->K, BK);
  VL_CHECK(a->N % 8 == 0 && a->lda % 8 == 0 && a->ldw % 8 == 0 && a->ldw >= a->N, "vlaser_gemm_nn: N, lda, ldw must be multiples of 8 and ldw >= N");
  VL_CHECK(((uintptr_t)a->A & 15) == 0 && ((uintptr_t)a->W & 15) == 0, "vlaser_gemm_nn: operands must be 16-byte aligned");
  VL_CHECK(a->force_bm == 0 || (glds_code(a->force_bm) && a->force_bm != 1564 && a->force_bm != 1566),
           "vlaser_gemm_nn: force_bm must be 0 or an LDS-DMA configuration code 1100/1105/1200/1300/1440/1500/1506/1532/1900, 2100 (lab: producer waves)");
  VL_CHECK(a->batch <= 1 || epi == VL_EPI_NONE || epi == VL_EPI_F32, "vlaser_gemm_nn: batched mode supports the NONE / F32 epilogues");
  VL_CHECK(!a->sumsq_part, "vlaser_gemm_nn: sumsq_part is honoured by vlaser_gemm_tn_lds only");
  switch (epi) {
    case VL_EPI_NONE: VL_CHECK(a->out, "out null"); return launch<VL_EPI_NONE, true>(a, stream);
    case VL_EPI_F32: VL_CHECK(a->out, "out null"); return launch<VL_EPI_F32, true>(a, stream);
    case VL_EPI_SWIGLU_BWD:
      VL_CHECK(a->out && a->res && a->N % 16 == 0 && a->ldo % 4 == 0 && a->ldo >= 2 * a->N, "swiglu_bwd: out / res null, N %% 16 != 0 or ldo < 2 N");
      VL_CHECK((((uintptr_t)a->out | (uintptr_t)a->res) & 7) == 0, "swiglu_bwd: out / res must be 8-byte aligned");
      return launch<VL_EPI_SWIGLU_BWD, true>(a, stream);
    case VL_EPI_PARTIAL:
      VL_CHECK(a->out_f32 && a->N % 4 == 0, "partial: out_f32 null or N %% 4 != 0");
      VL_CHECK(a->k_splits >= 1 && a->K % (a->k_splits * BK) == 0, "partial: K=%d not divisible by k_splits*64 (k_splits=%d)", a->K, a->k_splits);
      return launch<VL_EPI_PARTIAL, true>(a, stream);
    default: break;
  }
  vlaser_set_error("vlaser_gemm_nn: epilogue %d is not available in the NN form (NONE / F32 / PARTIAL / SWIGLU_BWD)", epi);
  return -1;
}
