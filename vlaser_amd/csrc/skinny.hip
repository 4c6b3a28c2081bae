// Skinny (M <= 16 activation rows) weight-streaming GEMV on MFMA for gfx950 -- the HBM-bound regime of the 10
// action-expert Euler steps and of greedy decode: every weight byte is read exactly once per call.
//
// Block = 4 waves; a block owns one "unit" = 32 consecutive (packed) weight rows = two 16-row MFMA tiles, and
// the 4 waves split the block's K range (in-block split-K, reduced through LDS).  blockIdx.y adds cross-block
// split-K for narrow outputs (o_proj / down_proj): fp32 partial slabs go to a workspace and are summed by the
// CONSUMER's prologue -- deterministic, no atomics, no extra launch.
// Weight fragments go global -> VGPR directly in MFMA layout (lane (r, kq) loads 16 B of row r at k + 8 kq): the
// weights are streamed once and never shared between waves, so an LDS round trip would be pure overhead (guide:
// "GEMV / M <= 16 decode weights: load straight to VGPRs, deep unroll").  Activations (a few KB) sit in LDS.
// MFMA operands are swapped (W as A, x as B) so each lane owns 4 consecutive outputs n of one row m, sharing the
// fused epilogues of the big GEMM (bias / SiLU / SwiGLU / RoPE + KV-cache scatter).
#include "common.h"
#include "../../include/vlaser_hip.h"

struct SkinnyP {
  VlaserSkinnyArgs a;
  int xs_stride;  // bytes per LDS activation row
  int kb;         // K per block
};

template <int PRO, int EPI>
__global__ __launch_bounds__(256) void skinny_kernel(SkinnyP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const VlaserSkinnyArgs& a = p.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, g = lane >> 4;
  const int unit = blockIdx.x, ks = blockIdx.y;
  const int kb0 = ks * p.kb;                  // first k of this block
  char* xs = smem;                            // [M][xs_stride]
  float* red = reinterpret_cast<float*>(smem + ((a.M * p.xs_stride + 15) & ~15));  // [4 waves][2][64][4]

  // ------------------------------------------------------------------ weight stream set-up: this wave's K quarter; the first batch of
  // weight loads is issued BEFORE the prologue so HBM latency overlaps the activation / partial-slab reads
  const bf16_t* W = reinterpret_cast<const bf16_t*>(a.W);
  const int kw = p.kb >> 2;                   // K per wave (multiple of 32)
  const int kl0 = wave * kw;                  // block-local k start
  const int row0 = unit * 32 + fr, row1 = row0 + 16;
  const bool r0ok = row0 < a.N, r1ok = row1 < a.N;
  const bf16_t* w0 = W + (size_t)(r0ok ? row0 : 0) * a.ldw + kb0 + kl0 + g * 8;
  const bf16_t* w1 = W + (size_t)(r1ok ? row1 : 0) * a.ldw + kb0 + kl0 + g * 8;
  const char* xrow = xs + (fr < a.M ? fr : 0) * p.xs_stride + (kl0 + g * 8) * 2;
  const bool mok = fr < a.M;
  f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
  const int nsteps = kw >> 5;
  constexpr int U = 4;
  u32x4 c0[U], c1[U];
  auto issue = [&](int s0, u32x4* d0, u32x4* d1) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int s = s0 + u;
      if (s < nsteps) {
        d0[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(w0 + s * 32));
        d1[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(w1 + s * 32));
      }
    }
  };
  issue(0, c0, c1);

  // ------------------------------------------------------------------ prologue: activations -> LDS (bf16)
  if constexpr (PRO == VL_PRO_PLAIN) {
    const bf16_t* X = reinterpret_cast<const bf16_t*>(a.x);
    const int chunks_per_row = p.kb >> 3;
    for (int c = tid; c < a.M * chunks_per_row; c += 256) {
      const int m = c / chunks_per_row, j = c - m * chunks_per_row;
      *reinterpret_cast<u32x4*>(xs + m * p.xs_stride + j * 16) = ld_global_16(X + (size_t)m * a.K + kb0 + j * 8);
    }
  } else {
    // h = bf16(h_in + sum partials); xn = bf16(w * bf16(h * rsqrt(mean(h^2) + eps)))
    // Phase 1: all 256 threads stride over the M*K/8 16-byte chunks; the partial-slab loads of a chunk are issued
    // in independent batches of 4 slabs (8 x 16 B in flight per lane) -- one L2 round trip per batch instead of one
    // per slab.  Phase 2: one wave per row takes the sum of squares from LDS and normalises in place.
    const bf16_t* Hin = reinterpret_cast<const bf16_t*>(a.x);
    const bf16_t* Wn = reinterpret_cast<const bf16_t*>(a.norm_w);
    const bool write_h = (a.h_out != nullptr) && unit == 0 && ks == 0;
    const int cpr = a.K >> 3;  // chunks per row
    const size_t slab = (size_t)a.M * a.K;
    for (int ch = tid; ch < a.M * cpr; ch += 256) {
      const int m = ch / cpr, c = (ch - m * cpr) << 3;
      const size_t off = (size_t)m * a.K + c;
      const u32x4 hv = ld_global_16(Hin + off);
      float v[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[2 * j] = bf16lo_to_f32(hv[j]); v[2 * j + 1] = bf16hi_to_f32(hv[j]); }
      int sp = 0;
      for (; sp + 4 <= a.n_partials; sp += 4) {
        f32x4 q[8];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float* pp = a.partials + (size_t)(sp + u) * slab + off;
          q[2 * u] = *reinterpret_cast<const f32x4*>(pp);
          q[2 * u + 1] = *reinterpret_cast<const f32x4*>(pp + 4);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int j = 0; j < 4; ++j) { v[j] += q[2 * u][j]; v[4 + j] += q[2 * u + 1][j]; }
      }
      for (; sp < a.n_partials; ++sp) {
        const float* pp = a.partials + (size_t)sp * slab + off;
        const f32x4 p0 = *reinterpret_cast<const f32x4*>(pp), p1 = *reinterpret_cast<const f32x4*>(pp + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] += p0[j]; v[4 + j] += p1[j]; }
      }
      u32x4 hr;
#pragma unroll
      for (int j = 0; j < 4; ++j) hr[j] = pack_bf16x2(v[2 * j], v[2 * j + 1]);
      *reinterpret_cast<u32x4*>(xs + m * p.xs_stride + c * 2) = hr;
      if (write_h) st_global_16(reinterpret_cast<bf16_t*>(a.h_out) + off, hr);
    }
    __syncthreads();
    for (int m = wave; m < a.M; m += 4) {
      float ssq = 0.f;
      for (int c = lane * 8; c < a.K; c += 512) {
        const u32x4 hr = *reinterpret_cast<const u32x4*>(xs + m * p.xs_stride + c * 2);
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float lo = bf16lo_to_f32(hr[j]), hi = bf16hi_to_f32(hr[j]); ssq += lo * lo + hi * hi; }
      }
      ssq = wave_sum(ssq);
      const float rs = rsqrtf(ssq / (float)a.K + a.eps);
      for (int c = lane * 8; c < a.K; c += 512) {
        const u32x4 hr = *reinterpret_cast<const u32x4*>(xs + m * p.xs_stride + c * 2);
        const u32x4 wv = ld_global_16(Wn + c);
        u32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float lo = round_bf16(bf16lo_to_f32(hr[j]) * rs) * bf16lo_to_f32(wv[j]);
          const float hi = round_bf16(bf16hi_to_f32(hr[j]) * rs) * bf16hi_to_f32(wv[j]);
          o[j] = pack_bf16x2(lo, hi);
        }
        *reinterpret_cast<u32x4*>(xs + m * p.xs_stride + c * 2) = o;
      }
    }
  }
  __syncthreads();

  // ------------------------------------------------------------------ main loop
  for (int s0 = 0; s0 < nsteps; s0 += U) {
    u32x4 n0[U], n1[U];
    issue(s0 + U, n0, n1);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (s0 + u < nsteps) {
        u32x4 xv = {0, 0, 0, 0};
        if (mok) xv = *reinterpret_cast<const u32x4*>(xrow + (s0 + u) * 64);
        const bf16x8 xf = as_bf16x8(xv);
        acc0 = mfma16(as_bf16x8(r0ok ? c0[u] : u32x4{0, 0, 0, 0}), xf, acc0);
        acc1 = mfma16(as_bf16x8(r1ok ? c1[u] : u32x4{0, 0, 0, 0}), xf, acc1);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) { c0[u] = n0[u]; c1[u] = n1[u]; }
  }

  // ------------------------------------------------------------------ in-block split-K reduce
  if (wave != 0) {
    float* r = red + ((wave - 1) * 2 * 64 + lane) * 4;
    *reinterpret_cast<f32x4*>(r) = acc0;
    *reinterpret_cast<f32x4*>(r + 64 * 4) = acc1;
  }
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int w = 0; w < 3; ++w) {
    const float* r = red + (w * 2 * 64 + lane) * 4;
    const f32x4 t0 = *reinterpret_cast<const f32x4*>(r), t1 = *reinterpret_cast<const f32x4*>(r + 64 * 4);
    acc0 += t0;
    acc1 += t1;
  }

  // ------------------------------------------------------------------ epilogue: lane -> row m = fr, n = unit*32 + t*16 + g*4 + j
  const int m = fr;
  if (m >= a.M) return;
  const int n0 = unit * 32 + g * 4;
  if constexpr (EPI == VL_SK_PARTIAL) {
    float* o = a.out_f32 + ((size_t)ks * a.M + m) * a.N;
    if (n0 + 3 < a.N) *reinterpret_cast<f32x4*>(o + n0) = acc0;
    if (n0 + 19 < a.N) *reinterpret_cast<f32x4*>(o + n0 + 16) = acc1;
  } else if constexpr (EPI == VL_SK_F32) {
    float* o = a.out_f32 + (size_t)m * a.N;
    const bf16_t* bias = reinterpret_cast<const bf16_t*>(a.bias);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (n0 + j < a.N) o[n0 + j] = acc0[j] + (bias ? bf16_to_f32(bias[n0 + j]) : 0.f);
      if (n0 + 16 + j < a.N) o[n0 + 16 + j] = acc1[j] + (bias ? bf16_to_f32(bias[n0 + 16 + j]) : 0.f);
    }
  } else if constexpr (EPI == VL_SK_BIAS || EPI == VL_SK_BIAS_SILU) {
    const bf16_t* bias = reinterpret_cast<const bf16_t*>(a.bias);
    bf16_t* o = reinterpret_cast<bf16_t*>(a.out) + (size_t)m * a.ldo;
    float r0[4], r1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      r0[j] = acc0[j] + bf16_to_f32(bias[n0 + j]);
      r1[j] = acc1[j] + bf16_to_f32(bias[n0 + 16 + j]);
      if constexpr (EPI == VL_SK_BIAS_SILU) { r0[j] = silu(round_bf16(r0[j])); r1[j] = silu(round_bf16(r1[j])); }
    }
    *reinterpret_cast<u32x2*>(o + n0) = u32x2{pack_bf16x2(r0[0], r0[1]), pack_bf16x2(r0[2], r0[3])};
    *reinterpret_cast<u32x2*>(o + n0 + 16) = u32x2{pack_bf16x2(r1[0], r1[1]), pack_bf16x2(r1[2], r1[3])};
  } else if constexpr (EPI == VL_SK_SWIGLU) {
    // unit = [gate16 | up16]; output columns unit*16 + g*4 + j
    bf16_t* o = reinterpret_cast<bf16_t*>(a.out) + (size_t)m * a.ldo + unit * 16 + g * 4;
    float r[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = round_bf16(silu(round_bf16(acc0[j]))) * round_bf16(acc1[j]);
    *reinterpret_cast<u32x2*>(o) = u32x2{pack_bf16x2(r[0], r[1]), pack_bf16x2(r[2], r[3])};
  } else if constexpr (EPI == VL_SK_QKV_ROPE) {
    // packed rows: n = head*128 + 32*j + 16*half + r ; acc0 = half 0 (d = 16*j + r), acc1 = half 1 (d + 64)
    const int head = n0 >> 7, pp = n0 & 127;
    const int d = ((pp >> 5) << 4) + (pp & 15);
    const int pos = a.pos_ids[m];
    const bf16_t* bias = reinterpret_cast<const bf16_t*>(a.bias);
    float x1[4], x2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      x1[j] = round_bf16(acc0[j] + bf16_to_f32(bias[n0 + j]));
      x2[j] = round_bf16(acc1[j] + bf16_to_f32(bias[n0 + 16 + j]));
    }
    const int b = m / a.tok_per_batch;
    const int slot = a.slot_base + (m - b * a.tok_per_batch);
    const int nq = a.n_q_heads, nkv = a.n_kv_heads;
    if (head < nq + nkv) {
      float o1[4], o2[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float c = a.rope_cos[(size_t)pos * 64 + d + j], s = a.rope_sin[(size_t)pos * 64 + d + j];
        o1[j] = x1[j] * c - x2[j] * s;
        o2[j] = x2[j] * c + x1[j] * s;
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

template <int PRO, int EPI>
static int launch(const VlaserSkinnyArgs* a, hipStream_t stream) {
  SkinnyP p;
  p.a = *a;
  p.kb = a->K / a->k_splits;
  p.xs_stride = p.kb * 2 + 16;
  const int lds = ((a->M * p.xs_stride + 15) & ~15) + 3 * 2 * 64 * 4 * 4;
  static int attr_lds = 0;
  if (lds > attr_lds) {
    VL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(skinny_kernel<PRO, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_lds = lds;
  }
  hipLaunchKernelGGL((skinny_kernel<PRO, EPI>), dim3((a->N + 31) / 32, a->k_splits), dim3(256), lds, stream, p);
  VL_LAUNCH_CHECK();
  return 0;
}

extern "C" int vlaser_skinny(int pro, int epi, const VlaserSkinnyArgs* a, vl_stream_t s) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(s);
  VL_CHECK(a && a->x && a->W, "vlaser_skinny: null operand");
  VL_CHECK(a->M >= 1 && a->M <= 16, "vlaser_skinny: M=%d must be in 1..16", a->M);
  VL_CHECK(a->k_splits >= 1 && a->K % (a->k_splits * 128) == 0, "vlaser_skinny: K=%d not divisible by k_splits*128 (k_splits=%d)", a->K, a->k_splits);
  VL_CHECK(a->ldw % 8 == 0 && ((uintptr_t)a->W & 15) == 0 && ((uintptr_t)a->x & 15) == 0, "vlaser_skinny: alignment");
  VL_CHECK((size_t)a->M * (a->K / a->k_splits * 2 + 16) <= 150000, "vlaser_skinny: activation tile does not fit in LDS");
  if (pro == VL_PRO_NORM) {
    VL_CHECK(a->norm_w && a->k_splits == 1, "vlaser_skinny: NORM prologue needs norm_w and k_splits == 1");
    VL_CHECK(a->n_partials == 0 || a->partials, "vlaser_skinny: partials null");
  }
  if (epi != VL_SK_F32 && epi != VL_SK_PARTIAL) VL_CHECK(a->N % 32 == 0, "vlaser_skinny: N must be a multiple of 32 for this epilogue");
  if (epi == VL_SK_PARTIAL) VL_CHECK(a->N % 4 == 0 && a->out_f32, "vlaser_skinny: partial needs N%%4==0 and out_f32");
  if (epi != VL_SK_PARTIAL) VL_CHECK(a->k_splits == 1, "vlaser_skinny: only VL_SK_PARTIAL may split K across blocks");
#define SK_CASE(P, E)                                   \
  if (pro == P && epi == E) return launch<P, E>(a, stream);
  SK_CASE(VL_PRO_PLAIN, VL_SK_PARTIAL)
  SK_CASE(VL_PRO_PLAIN, VL_SK_BIAS)
  SK_CASE(VL_PRO_PLAIN, VL_SK_BIAS_SILU)
  SK_CASE(VL_PRO_PLAIN, VL_SK_F32)
  SK_CASE(VL_PRO_NORM, VL_SK_QKV_ROPE)
  SK_CASE(VL_PRO_NORM, VL_SK_SWIGLU)
  SK_CASE(VL_PRO_NORM, VL_SK_F32)
#undef SK_CASE
  vlaser_set_error("vlaser_skinny: unsupported prologue/epilogue combination %d/%d", pro, epi);
  return -1;
}
