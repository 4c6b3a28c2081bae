// Backward / optimizer kernels of the SFT step for gfx950.  All are HBM-bound row / element kernels:
// 16-byte vector accesses where the layout allows, fp32 statistics, deterministic reductions (no atomics).
// GEMM-shaped backward work (dgrad, wgrad, attention backward through materialised per-head score matrices) goes
// through vlaser_gemm on transposed operands produced by transpose_kernel.
#include <stdlib.h>

#include "common.h"
#include "../../include/vlaser_hip.h"

// ---------------------------------------------------------------------------------------------- transpose (bf16)
// out[b][c*ld_out + r] = in[b][r][c] for r < rows, 0 for rows <= r < pad_rows; 64x64 tiles through LDS.
__global__ __launch_bounds__(256) void transpose_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int rows, int cols, int ld_in,
                                                        int ld_out, int pad_rows, long long in_bs, long long out_bs, int inner, long long in_is,
                                                        long long out_is) {
  __shared__ bf16_t tile[64][66];
  const int b = blockIdx.z / inner, bi = blockIdx.z - b * inner, r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const bf16_t* src = in + (size_t)b * in_bs + (size_t)bi * in_is;
  bf16_t* dst = out + (size_t)b * out_bs + (size_t)bi * out_is;
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int r = i >> 6, c = i & 63;
    tile[r][c] = (r0 + r < rows && c0 + c < cols) ? src[(size_t)(r0 + r) * ld_in + c0 + c] : (bf16_t)0;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int c = i >> 6, r = i & 63;
    if (c0 + c < cols && r0 + r < pad_rows) dst[(size_t)(c0 + c) * ld_out + r0 + r] = tile[r][c];
  }
}
// vector variant: 16-byte global loads and stores on both sides (rows, cols, ld_in, ld_out, pad_rows multiples of 8,
// 16-byte aligned bases); LDS tile row stride 72 elements keeps ds_write_b128 aligned.
__global__ __launch_bounds__(256) void transpose_vec_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int rows, int cols, int ld_in,
                                                            int ld_out, int pad_rows, long long in_bs, long long out_bs, int inner, long long in_is, long long out_is) {
  __shared__ __attribute__((aligned(16))) bf16_t tile[64][72];
  const int b = blockIdx.z / inner, bi = blockIdx.z - b * inner, r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const bf16_t* src = in + (size_t)b * in_bs + (size_t)bi * in_is;
  bf16_t* dst = out + (size_t)b * out_bs + (size_t)bi * out_is;
  const int tr = threadIdx.x >> 3, tc = (threadIdx.x & 7) * 8;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int r = tr + pass * 32;
    u32x4 v = {0, 0, 0, 0};
    if (r0 + r < rows && c0 + tc < cols) v = ld_global_16(src + (size_t)(r0 + r) * ld_in + c0 + tc);
    *reinterpret_cast<u32x4*>(&tile[r][tc]) = v;
  }
  __syncthreads();
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int c = tr + pass * 32;            // output row (= input column)
    if (c0 + c < cols && r0 + tc < pad_rows) {
      bf16_t e[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) e[j] = tile[tc + j][c];
      u32x4 v = {(uint32_t)e[0] | ((uint32_t)e[1] << 16), (uint32_t)e[2] | ((uint32_t)e[3] << 16), (uint32_t)e[4] | ((uint32_t)e[5] << 16),
                 (uint32_t)e[6] | ((uint32_t)e[7] << 16)};
      st_global_16(dst + (size_t)(c0 + c) * ld_out + r0 + tc, v);
    }
  }
}
extern "C" int vlaser_transpose(const void* in, void* out, int rows, int cols, int ld_in, int ld_out, int pad_rows, int batch, long long in_bs,
                                long long out_bs, int inner, long long in_is, long long out_is, vl_stream_t s) {
  VL_CHECK(in && out && rows > 0 && cols > 0 && pad_rows >= rows && ld_out >= pad_rows && batch >= 1 && inner >= 1, "vlaser_transpose: bad args");
  const dim3 grid((cols + 63) / 64, (pad_rows + 63) / 64, batch * inner);
  const bool vec = ((rows | cols | ld_in | ld_out | pad_rows) & 7) == 0 && (((uintptr_t)in | (uintptr_t)out) & 15) == 0 &&
                   ((in_bs | out_bs | in_is | out_is) & 7) == 0;
  if (vec)
    hipLaunchKernelGGL(transpose_vec_kernel, grid, dim3(256), 0, (hipStream_t)s, (const bf16_t*)in, (bf16_t*)out, rows, cols, ld_in, ld_out, pad_rows,
                       in_bs, out_bs, inner, in_is, out_is);
  else
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, (hipStream_t)s, (const bf16_t*)in, (bf16_t*)out, rows, cols, ld_in, ld_out, pad_rows, in_bs,
                       out_bs, inner, in_is, out_is);
  VL_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------- fused softmax + dS
// One wave per (h, q) row, the row (ld <= 1024 columns) stays in registers: P = softmax(scale * s) over k <= q, D = <dO, O>,
// dS = P (dP - D) scale.  dK / dV then come from the grouped TN GEMM, so no transposed copies of P / dS are produced.
__global__ __launch_bounds__(256) void attn_bwd_pds_kernel(const float* __restrict__ sc, const float* __restrict__ dP, const bf16_t* __restrict__ dO,
                                                           const bf16_t* __restrict__ O, bf16_t* __restrict__ P, bf16_t* __restrict__ dS, int H, int S,
                                                           int ld, int hd, float scale, int causal, int kv_valid, int q_off) {
  // visibility of key k for query q: k < kv_valid and (k <= q when causal).  causal = 1, kv_valid = ld: the SFT step's causal mask; causal = 0:
  // the bidirectional valid prefix of the VLA block mask (pizero_internvl.py:517-587) and the ViT's full attention (r03, f1)
  const int lane = threadIdx.x & 63, q = blockIdx.x * 4 + (threadIdx.x >> 6), h = blockIdx.y;
  if (q >= S) return;
  const int klim = causal ? min(q + q_off + 1, kv_valid) : kv_valid;      // q_off: global index of query row 0 (a block of rows of a longer sequence)
  const bf16_t* dorow = dO + (size_t)q * H * hd + h * hd;
  const bf16_t* orow = O + (size_t)q * H * hd + h * hd;
  float d = 0.f;
  for (int i = lane * 2; i < hd; i += 128) {
    const uint32_t a = *reinterpret_cast<const uint32_t*>(dorow + i), b = *reinterpret_cast<const uint32_t*>(orow + i);
    d += bf16lo_to_f32(a) * bf16lo_to_f32(b) + bf16hi_to_f32(a) * bf16hi_to_f32(b);
  }
  d = wave_sum(d);
  const size_t ro = ((size_t)h * S + q) * ld;
  float v[16];
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int k = lane + 64 * j;
    v[j] = (k < klim && k < ld) ? sc[ro + k] * scale : -INFINITY;
    mx = fmaxf(mx, v[j]);
  }
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) { v[j] = __expf(v[j] - mx); sum += v[j]; }       // exp(-inf) = 0 outside the causal range
  sum = wave_sum(sum);
  const float inv = 1.0f / sum;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int k = lane + 64 * j;
    if (k < ld) {
      const bf16_t pb = f32_to_bf16(v[j] * inv);
      P[ro + k] = pb;
      dS[ro + k] = (k < klim) ? f32_to_bf16(bf16_to_f32(pb) * (dP[ro + k] - d) * scale) : (bf16_t)0;
    }
  }
}
// rows longer than 1024 columns (multi-tile samples): same math with the row re-read from memory in three passes
__global__ __launch_bounds__(256) void attn_bwd_pds_long_kernel(const float* __restrict__ sc, const float* __restrict__ dP, const bf16_t* __restrict__ dO,
                                                                const bf16_t* __restrict__ O, bf16_t* __restrict__ P, bf16_t* __restrict__ dS, int H,
                                                                int S, int ld, int hd, float scale, int causal, int kv_valid, int q_off) {
  const int lane = threadIdx.x & 63, q = blockIdx.x * 4 + (threadIdx.x >> 6), h = blockIdx.y;
  if (q >= S) return;
  const int klim = causal ? min(q + q_off + 1, kv_valid) : kv_valid;
  const bf16_t* dorow = dO + (size_t)q * H * hd + h * hd;
  const bf16_t* orow = O + (size_t)q * H * hd + h * hd;
  float d = 0.f;
  for (int i = lane * 2; i < hd; i += 128) {
    const uint32_t a = *reinterpret_cast<const uint32_t*>(dorow + i), b = *reinterpret_cast<const uint32_t*>(orow + i);
    d += bf16lo_to_f32(a) * bf16lo_to_f32(b) + bf16hi_to_f32(a) * bf16hi_to_f32(b);
  }
  d = wave_sum(d);
  const size_t ro = ((size_t)h * S + q) * ld;
  float mx = -INFINITY;
  for (int k = lane; k < klim; k += 64) mx = fmaxf(mx, sc[ro + k] * scale);
  mx = wave_max(mx);
  float sum = 0.f;
  for (int k = lane; k < klim; k += 64) sum += __expf(sc[ro + k] * scale - mx);
  sum = wave_sum(sum);
  const float inv = 1.0f / sum;
  for (int k = lane; k < ld; k += 64) {
    const bf16_t pb = (k < klim) ? f32_to_bf16(__expf(sc[ro + k] * scale - mx) * inv) : (bf16_t)0;
    P[ro + k] = pb;
    dS[ro + k] = (k < klim) ? f32_to_bf16(bf16_to_f32(pb) * (dP[ro + k] - d) * scale) : (bf16_t)0;
  }
}
extern "C" int vlaser_attn_bwd_pds_masked(const float* scores, const float* dP, const void* dO, const void* O, void* P, void* dS, int H, int S, int ld, int hd,
                                          float scale, int causal, int kv_valid, int q_off, vl_stream_t s) {
  VL_CHECK(scores && dP && dO && O && P && dS && hd % 2 == 0 && kv_valid >= 1 && kv_valid <= ld && q_off >= 0, "vlaser_attn_bwd_pds: bad args");
  if (ld <= 1024)
    hipLaunchKernelGGL(attn_bwd_pds_kernel, dim3((S + 3) / 4, H), dim3(256), 0, (hipStream_t)s, scores, dP, (const bf16_t*)dO, (const bf16_t*)O,
                       (bf16_t*)P, (bf16_t*)dS, H, S, ld, hd, scale, causal, kv_valid, q_off);
  else
    hipLaunchKernelGGL(attn_bwd_pds_long_kernel, dim3((S + 3) / 4, H), dim3(256), 0, (hipStream_t)s, scores, dP, (const bf16_t*)dO, (const bf16_t*)O,
                       (bf16_t*)P, (bf16_t*)dS, H, S, ld, hd, scale, causal, kv_valid, q_off);
  VL_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------- RoPE backward + pack
// forward: o1 = x1 c - x2 s, o2 = x2 c + x1 s (x1 = d < 64, x2 = d + 64)  =>  dx1 = do1 c + do2 s, dx2 = do2 c - do1 s.
// Output column order = packed q/k/v rows (ops.head_perm): col = head*128 + 32*(d/16) + 16*half + d%16, d in [0,64).
__global__ __launch_bounds__(256) void rope_bwd_pack_kernel(const bf16_t* __restrict__ dq, const bf16_t* __restrict__ dk, const bf16_t* __restrict__ dv,
                                                            const float* __restrict__ cosT, const float* __restrict__ sinT,
                                                            const int32_t* __restrict__ pos_ids, bf16_t* __restrict__ out, int n_q, int n_kv,
                                                            int kv_per_q_head, const bf16_t* __restrict__ dk_extra, const bf16_t* __restrict__ dv_extra) {
  // dk_extra / dv_extra (optional, [S, n_kv*128]): a second source of key / value gradients for the same rows, added in fp32 before the rotation --
  // the keys of the VLM rows are also read by the proprio / action rows of the joint attention (f1, train_vlm: True)
  // kv_per_q_head != 0: dk / dv hold one partial per Q head ([S, n_q*128], the per-head TN GEMMs of the attention backward);
  // the kv gradient is their sum over the G = n_q / n_kv heads of the group (fp32, fixed order)
  const int s = blockIdx.x, nh = n_q + 2 * n_kv, G = n_q / n_kv;
  const int pos = pos_ids[s];
  for (int i = threadIdx.x; i < nh * 64; i += 256) {
    const int head = i >> 6, d = i & 63;
    float g1, g2;
    bool rot = true;
    if (head < n_q) {
      const bf16_t* src = dq + (size_t)s * n_q * 128 + head * 128;
      g1 = bf16_to_f32(src[d]); g2 = bf16_to_f32(src[d + 64]);
    } else {
      const bool isv = head >= n_q + n_kv;
      const int kvh = head - n_q - (isv ? n_kv : 0);
      const bf16_t* base = isv ? dv : dk;
      rot = !isv;
      if (kv_per_q_head) {
        g1 = 0.f; g2 = 0.f;
        for (int gi = 0; gi < G; ++gi) {
          const bf16_t* src = base + (size_t)s * n_q * 128 + (kvh * G + gi) * 128;
          g1 += bf16_to_f32(src[d]); g2 += bf16_to_f32(src[d + 64]);
        }
      } else {
        const bf16_t* src = base + (size_t)s * n_kv * 128 + kvh * 128;
        g1 = bf16_to_f32(src[d]); g2 = bf16_to_f32(src[d + 64]);
      }
      const bf16_t* ex = isv ? dv_extra : dk_extra;
      if (ex) {
        const bf16_t* src = ex + (size_t)s * n_kv * 128 + kvh * 128;
        g1 += bf16_to_f32(src[d]); g2 += bf16_to_f32(src[d + 64]);
      }
    }
    float x1 = g1, x2 = g2;
    if (rot) {
      const float c = cosT[(size_t)pos * 64 + d], sn = sinT[(size_t)pos * 64 + d];
      x1 = g1 * c + g2 * sn;
      x2 = g2 * c - g1 * sn;
    }
    bf16_t* o = out + (size_t)s * nh * 128 + head * 128 + 32 * (d >> 4) + (d & 15);
    o[0] = f32_to_bf16(x1);
    o[16] = f32_to_bf16(x2);
  }
}
extern "C" int vlaser_rope_bwd_pack_ex(const void* dq, const void* dk, const void* dv, const float* c, const float* sn, const int32_t* pos, void* out,
                                       int S, int n_q, int n_kv, int kv_per_q_head, const void* dk_extra, const void* dv_extra, vl_stream_t s) {
  VL_CHECK(dq && dk && dv && c && sn && pos && out && S > 0 && n_q % n_kv == 0 && (!dk_extra == !dv_extra), "vlaser_rope_bwd_pack: bad args");
  hipLaunchKernelGGL(rope_bwd_pack_kernel, dim3(S), dim3(256), 0, (hipStream_t)s, (const bf16_t*)dq, (const bf16_t*)dk, (const bf16_t*)dv, c, sn, pos,
                     (bf16_t*)out, n_q, n_kv, kv_per_q_head, (const bf16_t*)dk_extra, (const bf16_t*)dv_extra);
  VL_LAUNCH_CHECK();
  return 0;
}
extern "C" int vlaser_rope_bwd_pack(const void* dq, const void* dk, const void* dv, const float* c, const float* sn, const int32_t* pos, void* out,
                                    int S, int n_q, int n_kv, int kv_per_q_head, vl_stream_t s) {
  return vlaser_rope_bwd_pack_ex(dq, dk, dv, c, sn, pos, out, S, n_q, n_kv, kv_per_q_head, nullptr, nullptr, s);
}

// ---------------------------------------------------------------------------------------------- RMSNorm backward
// y = w * bf16(x rs), rs = rsqrt(mean(x^2) + eps).  dx = rs * (g - xhat * mean(g xhat)), g = w dy, xhat = x rs.
// dx_out = dres + dx.  One wave per row.
// dy_part (r04, optional): dy is not given as a bf16 tensor but as the n_part fp32 split-K slabs [n_part][S][C] of the dgrad GEMM that produced it; they
// are added in slab order and rounded to bf16 exactly as vlaser_reduce_norm would have (same bits), once, into an LDS row -- the stand-alone reduction
// launch between the gate/up dgrad and this kernel is gone.
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                          const bf16_t* __restrict__ dres, bf16_t* __restrict__ dx, float* __restrict__ dw_partial,
                                                          int S, int C, float eps, const float* __restrict__ dy_part, int n_part) {
  // dw_partial (optional, fp32 [gridDim.x][C]): this block's share of the weight gradient dw[c] = sum_s dy[s,c] x[s,c] rs_s, so the
  // norm-weight gradient needs no pass of its own over dy / x (finished by colsum_partials_kernel)
  extern __shared__ float dw_lds[];            // [4][C] fp32 when dw_partial, then [4][C] bf16 when dy_part
  const int lane = threadIdx.x & 63, wv_ = threadIdx.x >> 6, row = blockIdx.x * 4 + wv_;
  const bool live = row < S;
  const size_t ro = (size_t)min(row, S - 1) * C;
  bf16_t* dyl = reinterpret_cast<bf16_t*>(dw_lds + (dw_partial ? 4 * C : 0)) + wv_ * C;      // this wave's row of reduced dy
  float ss = 0.f, dot = 0.f;
  for (int c = lane * 8; c < C; c += 512) {
    const u32x4 xv = ld_global_16(x + ro + c), wv = ld_global_16(w + c);
    u32x4 gv;
    if (dy_part) {
      float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      const float* pp = dy_part + ro + c;
      for (int u = 0; u < n_part; ++u, pp += (size_t)S * C) {
        const f32x4 q0 = *reinterpret_cast<const f32x4*>(pp), q1 = *reinterpret_cast<const f32x4*>(pp + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] += q0[j]; v[4 + j] += q1[j]; }
      }
      gv = u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
      *reinterpret_cast<u32x4*>(dyl + c) = gv;              // read back by the same lane in the second pass
    } else {
      gv = ld_global_16(dy + ro + c);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float x0 = bf16lo_to_f32(xv[j]), x1 = bf16hi_to_f32(xv[j]);
      ss += x0 * x0 + x1 * x1;
      dot += x0 * bf16lo_to_f32(gv[j]) * bf16lo_to_f32(wv[j]) + x1 * bf16hi_to_f32(gv[j]) * bf16hi_to_f32(wv[j]);
    }
  }
  ss = wave_sum(ss);
  dot = wave_sum(dot);
  const float rs = rsqrtf(ss / (float)C + eps);
  const float coef = dot * rs * rs * rs / (float)C;   // = rs * mean(g xhat) * rs
  for (int c = lane * 8; c < C; c += 512) {
    const u32x4 xv = ld_global_16(x + ro + c), wv = ld_global_16(w + c);
    const u32x4 gv = dy_part ? *reinterpret_cast<const u32x4*>(dyl + c) : ld_global_16(dy + ro + c);
    u32x4 rv = {0, 0, 0, 0};
    if (dres) rv = ld_global_16(dres + ro + c);
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float lo = bf16lo_to_f32(rv[j]) + rs * bf16lo_to_f32(gv[j]) * bf16lo_to_f32(wv[j]) - coef * bf16lo_to_f32(xv[j]);
      const float hi = bf16hi_to_f32(rv[j]) + rs * bf16hi_to_f32(gv[j]) * bf16hi_to_f32(wv[j]) - coef * bf16hi_to_f32(xv[j]);
      o[j] = pack_bf16x2(lo, hi);
      if (dw_partial) {
        dw_lds[wv_ * C + c + 2 * j] = live ? bf16lo_to_f32(gv[j]) * bf16lo_to_f32(xv[j]) * rs : 0.f;
        dw_lds[wv_ * C + c + 2 * j + 1] = live ? bf16hi_to_f32(gv[j]) * bf16hi_to_f32(xv[j]) * rs : 0.f;
      }
    }
    if (live) st_global_16(dx + ro + c, o);
  }
  if (dw_partial) {
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256)
      dw_partial[(size_t)blockIdx.x * C + c] = dw_lds[c] + dw_lds[C + c] + dw_lds[2 * C + c] + dw_lds[3 * C + c];
  }
}
// r04, C = 512 * NCH (NCH <= 8): the row lives in registers -- x, w, dres and dy (or the slabs, NCH x 2 loads per slab in flight) are requested up front and the second
// pass reads nothing: ONE load round trip (+ one per slab) instead of six (the loops above wait for each chunk's loads in turn: 12 us per launch at S = 560, C = 1536,
// 72 launches per SFT step).  Same operations in the same order per lane: bit-identical to the kernel above.
template <int NCH>
__global__ __launch_bounds__(256) void rmsnorm_bwd_reg_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                              const bf16_t* __restrict__ dres, bf16_t* __restrict__ dx, float* __restrict__ dw_partial,
                                                              int S, float eps, const float* __restrict__ dy_part, int n_part) {
  constexpr int C = 512 * NCH;
  extern __shared__ float dw_lds[];            // [4][C] fp32 when dw_partial
  const int lane = threadIdx.x & 63, wv_ = threadIdx.x >> 6, row = blockIdx.x * 4 + wv_;
  const bool live = row < S;
  const size_t ro = (size_t)min(row, S - 1) * C;
  u32x4 xv[NCH], wv[NCH], gv[NCH], rv[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane * 8 + i * 512;
    xv[i] = ld_global_16(x + ro + c);
    wv[i] = ld_global_16(w + c);
    rv[i] = dres ? ld_global_16(dres + ro + c) : u32x4{0, 0, 0, 0};
    if (!dy_part) gv[i] = ld_global_16(dy + ro + c);
  }
  if (dy_part) {
    float v[NCH][8];
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) v[i][j] = 0.f;
    const float* pp = dy_part + ro + lane * 8;
    for (int u = 0; u < n_part; ++u, pp += (size_t)S * C) {
      f32x4 q0[NCH], q1[NCH];
#pragma unroll
      for (int i = 0; i < NCH; ++i) { q0[i] = *reinterpret_cast<const f32x4*>(pp + i * 512); q1[i] = *reinterpret_cast<const f32x4*>(pp + i * 512 + 4); }
#pragma unroll
      for (int i = 0; i < NCH; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[i][j] += q0[i][j]; v[i][4 + j] += q1[i][j]; }
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i)
      gv[i] = u32x4{pack_bf16x2(v[i][0], v[i][1]), pack_bf16x2(v[i][2], v[i][3]), pack_bf16x2(v[i][4], v[i][5]), pack_bf16x2(v[i][6], v[i][7])};
  }
  float ss = 0.f, dot = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float x0 = bf16lo_to_f32(xv[i][j]), x1 = bf16hi_to_f32(xv[i][j]);
      ss += x0 * x0 + x1 * x1;
      dot += x0 * bf16lo_to_f32(gv[i][j]) * bf16lo_to_f32(wv[i][j]) + x1 * bf16hi_to_f32(gv[i][j]) * bf16hi_to_f32(wv[i][j]);
    }
  ss = wave_sum(ss);
  dot = wave_sum(dot);
  const float rs = rsqrtf(ss / (float)C + eps);
  const float coef = dot * rs * rs * rs / (float)C;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane * 8 + i * 512;
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float lo = bf16lo_to_f32(rv[i][j]) + rs * bf16lo_to_f32(gv[i][j]) * bf16lo_to_f32(wv[i][j]) - coef * bf16lo_to_f32(xv[i][j]);
      const float hi = bf16hi_to_f32(rv[i][j]) + rs * bf16hi_to_f32(gv[i][j]) * bf16hi_to_f32(wv[i][j]) - coef * bf16hi_to_f32(xv[i][j]);
      o[j] = pack_bf16x2(lo, hi);
      if (dw_partial) {
        dw_lds[wv_ * C + c + 2 * j] = live ? bf16lo_to_f32(gv[i][j]) * bf16lo_to_f32(xv[i][j]) * rs : 0.f;
        dw_lds[wv_ * C + c + 2 * j + 1] = live ? bf16hi_to_f32(gv[i][j]) * bf16hi_to_f32(xv[i][j]) * rs : 0.f;
      }
    }
    if (live) st_global_16(dx + ro + c, o);
  }
  if (dw_partial) {
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256)
      dw_partial[(size_t)blockIdx.x * C + c] = dw_lds[c] + dw_lds[C + c] + dw_lds[2 * C + c] + dw_lds[3 * C + c];
  }
}
// out[c] (bf16) = sum_p partial[p][c], fixed order.  64 columns x 4 row-slices per workgroup: slice q sums rows q, q+4, ... with all of
// its loads independent (4 accumulators), the slices meet in LDS in a fixed order.  (One thread per column over the whole column,
// C / 256 = 6 workgroups, took 12 us for the 140 x 1536 partials of a norm-weight gradient: a serial chain of 35 load batches.)
// blockIdx.y > 0 (vlaser_colsum_partials_multi): tensor t = blockIdx.y reads its partials at partial + t * slot and writes out + out_off[t]
__global__ __launch_bounds__(256) void colsum_partials_kernel(const float* __restrict__ partial, int n_part, int C, bf16_t* __restrict__ out,
                                                              long long slot = 0, const long long* __restrict__ out_off = nullptr) {
  partial += (size_t)blockIdx.y * slot;
  if (out_off) out += out_off[blockIdx.y];
  __shared__ float red[4][64];
  const int cl = threadIdx.x & 63, q = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
  const int cc = min(c, C - 1);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int pi = q;
  for (; pi + 12 < n_part; pi += 16) {
    a0 += partial[(size_t)pi * C + cc]; a1 += partial[(size_t)(pi + 4) * C + cc];
    a2 += partial[(size_t)(pi + 8) * C + cc]; a3 += partial[(size_t)(pi + 12) * C + cc];
  }
  for (; pi < n_part; pi += 4) a0 += partial[(size_t)pi * C + cc];
  red[q][cl] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (q == 0 && c < C) out[c] = f32_to_bf16((red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]));
}
extern "C" int vlaser_rmsnorm_bwd(const void* dy, const void* x, const void* w, const void* dres, void* dx, void* dw_out, float* dw_ws, int S, int C,
                                  float eps, const float* dy_partials, int n_partials, vl_stream_t s) {
  VL_CHECK((dy || dy_partials) && x && w && dx && S > 0 && C % 8 == 0, "vlaser_rmsnorm_bwd: bad args");
  VL_CHECK(!dy_partials || (n_partials >= 1 && (((uintptr_t)dy_partials) & 15) == 0), "vlaser_rmsnorm_bwd: dy_partials = n_partials >= 1 fp32 slabs [S][C], 16-byte aligned");
  VL_CHECK(!dw_out || dw_ws, "vlaser_rmsnorm_bwd: the weight gradient needs the [ceil(S/4)][C] fp32 workspace");
  VL_CHECK(!dw_ws || C * 16 <= 64 * 1024, "vlaser_rmsnorm_bwd: weight-gradient partials need C <= 4096");
  const int nb = (S + 3) / 4;
  static const int old_form = getenv("VLASER_RMSNORM_BWD_LOOPS") ? atoi(getenv("VLASER_RMSNORM_BWD_LOOPS")) : 0;          // A/B: 1 = the chunk loops
  const int nch = (C % 512 == 0 && C / 512 <= 8 && !old_form) ? C / 512 : 0;
  const size_t lds = (dw_ws ? (size_t)C * 16 : 0) + (dy_partials && !nch ? (size_t)C * 8 : 0);       // (the register kernel keeps the reduced dy row in registers)
  VL_CHECK(lds <= 64 * 1024, "vlaser_rmsnorm_bwd: C = %d too wide", C);
#define VL_RB_CASE(N_)                                                                                                                                    \
  case N_:                                                                                                                                                \
    hipLaunchKernelGGL(rmsnorm_bwd_reg_kernel<N_>, dim3(nb), dim3(256), dw_ws ? (size_t)C * 16 : 0, (hipStream_t)s, (const bf16_t*)dy, (const bf16_t*)x,  \
                       (const bf16_t*)w, (const bf16_t*)dres, (bf16_t*)dx, dw_ws, S, eps, dy_partials, dy_partials ? n_partials : 0);                    \
    break;
  switch (nch) {
    VL_RB_CASE(1) VL_RB_CASE(2) VL_RB_CASE(3) VL_RB_CASE(4) VL_RB_CASE(5) VL_RB_CASE(6) VL_RB_CASE(7) VL_RB_CASE(8)
    default:
      hipLaunchKernelGGL(rmsnorm_bwd_kernel, dim3(nb), dim3(256), lds, (hipStream_t)s, (const bf16_t*)dy, (const bf16_t*)x,
                         (const bf16_t*)w, (const bf16_t*)dres, (bf16_t*)dx, dw_ws, S, C, eps, dy_partials, dy_partials ? n_partials : 0);
  }
#undef VL_RB_CASE
  if (dw_out) hipLaunchKernelGGL(colsum_partials_kernel, dim3((C + 63) / 64), dim3(256), 0, (hipStream_t)s, dw_ws, nb, C, (bf16_t*)dw_out);
  VL_LAUNCH_CHECK();
  return 0;
}

// Several norm-weight gradients finished by ONE launch (r04): vlaser_rmsnorm_bwd called with dw_ws but WITHOUT dw_out only leaves its per-block partials
// [ceil(S/4)][C] in its slot; tensor t's slot starts at ws + t * slot_stride, its bf16 output at out_base + out_off[t] (device int64 table).  Same sums in
// the same order as the per-call reduction.
extern "C" int vlaser_colsum_partials_multi(const float* ws, long long slot_stride, int n_tensors, int n_part, int C, void* out_base, const long long* out_off,
                                            vl_stream_t s) {
  VL_CHECK(ws && out_base && out_off && n_tensors >= 1 && n_part >= 1 && C >= 1 && slot_stride >= (long long)n_part * C, "vlaser_colsum_partials_multi: bad args");
  hipLaunchKernelGGL(colsum_partials_kernel, dim3((C + 63) / 64, n_tensors), dim3(256), 0, (hipStream_t)s, ws, n_part, C, (bf16_t*)out_base, slot_stride, out_off);
  VL_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------- column sums
// out[c] = sum_s a[s,c] * f(s,c): mode 0: 1; mode 1: b[s,c]; mode 2: b[s,c] * rs_s (rs recomputed per row from b);
// mode 3: LayerNorm-normalised b ((b - mean_s) * rs_s).  Block = 64 columns x 4 row-slices; deterministic.
#define CS_SLICES 16
__global__ __launch_bounds__(256) void colsum_mul_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b, const float* __restrict__ rowstat,
                                                         float* __restrict__ partial, int S, int C, int mode) {
  // grid (C/64, CS_SLICES): block = 64 columns x 4 row lanes over the rows of its slice; partial[slice][c]
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), lane_r = threadIdx.x >> 6;
  const int rows_per = (S + CS_SLICES - 1) / CS_SLICES;
  const int s0 = blockIdx.y * rows_per, s1 = min(S, s0 + rows_per);
  float acc = 0.f;
  if (c < C) {
    for (int s = s0 + lane_r; s < s1; s += 4) {
      float v = bf16_to_f32(a[(size_t)s * C + c]);
      if (mode == 1) v *= bf16_to_f32(b[(size_t)s * C + c]);
      else if (mode == 2) v *= bf16_to_f32(b[(size_t)s * C + c]) * rowstat[2 * s + 1];
      else if (mode == 3) v *= (bf16_to_f32(b[(size_t)s * C + c]) - rowstat[2 * s]) * rowstat[2 * s + 1];
      acc += v;
    }
  }
  red[lane_r][threadIdx.x & 63] = acc;
  __syncthreads();
  if (lane_r == 0 && c < C) partial[(size_t)blockIdx.y * C + c] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ partial, float* __restrict__ out, int C) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float acc = 0.f;
#pragma unroll
  for (int r = 0; r < CS_SLICES; ++r) acc += partial[(size_t)r * C + c];
  out[c] = acc;
}
// per-row statistics (mean, rs) of a bf16 matrix: rms mode (mean = 0, rs = rsqrt(mean(x^2)+eps)) or layernorm mode
__global__ __launch_bounds__(256) void rowstat_kernel(const bf16_t* __restrict__ x, float* __restrict__ st, int S, int C, float eps, int layernorm) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= S) return;
  const bf16_t* xr = x + (size_t)row * C;
  // r04: 16-byte loads (the 2-byte loop was C / 64 load round trips per pass: 36 us for 256 rows of 4096); the LayerNorm variance comes from the same registers when
  // the row fits (C <= 4096), else from a second pass
  const bool vec = (C % 8 == 0) && ((reinterpret_cast<uintptr_t>(xr) & 15) == 0);
  constexpr int MAXCH = 8;                       // 16-byte chunks per lane held in registers: C <= 64 * 8 * MAXCH = 4096
  const int nch = vec ? C / 8 : 0;               // chunks of the row
  u32x4 keep[MAXCH];
  float s1 = 0.f, s2 = 0.f;
  if (vec && nch <= 64 * MAXCH) {
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) keep[i] = (lane + 64 * i < nch) ? ld_global_16(xr + (size_t)(lane + 64 * i) * 8) : u32x4{0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < MAXCH; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float lo = __uint_as_float(keep[i][j] << 16), hi = __uint_as_float(keep[i][j] & 0xffff0000u);
        s1 += lo + hi; s2 += lo * lo + hi * hi;
      }
  } else {
    for (int c = lane; c < C; c += 64) { const float v = bf16_to_f32(xr[c]); s1 += v; s2 += v * v; }
  }
  s1 = wave_sum(s1); s2 = wave_sum(s2);
  float mean = 0.f, rs;
  if (layernorm) {
    mean = s1 / (float)C;
    float vs = 0.f;
    if (vec && nch <= 64 * MAXCH) {
#pragma unroll
      for (int i = 0; i < MAXCH; ++i)
        if (lane + 64 * i < nch) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float lo = __uint_as_float(keep[i][j] << 16) - mean, hi = __uint_as_float(keep[i][j] & 0xffff0000u) - mean;
            vs += lo * lo + hi * hi;
          }
        }
    } else {
      for (int c = lane; c < C; c += 64) { const float v = bf16_to_f32(xr[c]) - mean; vs += v * v; }
    }
    vs = wave_sum(vs);
    rs = rsqrtf(vs / (float)C + eps);
  } else {
    rs = rsqrtf(s2 / (float)C + eps);
  }
  if (lane == 0) { st[2 * row] = mean; st[2 * row + 1] = rs; }
}
extern "C" int vlaser_colsum_mul(const void* a, const void* b, float* out, int S, int C, int mode, float eps, float* ws, vl_stream_t s) {
  VL_CHECK(a && out && ws && S > 0 && C > 0 && mode >= 0 && mode <= 3 && (mode == 0 || b), "vlaser_colsum_mul: bad args (ws = float[2*S + 16*C])");
  float* st = ws;
  float* partial = ws + 2 * S;
  if (mode >= 2) {
    hipLaunchKernelGGL(rowstat_kernel, dim3((S + 3) / 4), dim3(256), 0, (hipStream_t)s, (const bf16_t*)b, st, S, C, eps, mode == 3);
  }
  hipLaunchKernelGGL(colsum_mul_kernel, dim3((C + 63) / 64, CS_SLICES), dim3(256), 0, (hipStream_t)s, (const bf16_t*)a, (const bf16_t*)b, st, partial, S, C,
                     mode);
  hipLaunchKernelGGL(colsum_final_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)s, partial, out, C);
  VL_LAUNCH_CHECK();
  return 0;
}

// bias gradient: out[c] (bf16) = sum_s a[s,c] in ONE launch (64 columns x 4 row lanes per block, fixed order): the matrices here
// are a few hundred rows, where the two-stage column sum's extra launches cost more than its parallelism buys
__global__ __launch_bounds__(1024) void colsum_bf16_kernel(const bf16_t* __restrict__ a, bf16_t* __restrict__ out, int S, int C, int lda) {
  __shared__ float red[16][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), lane_r = threadIdx.x >> 6;      // 16 row lanes x 64 columns
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (c < C) {
    int s = lane_r;
    for (; s + 48 < S; s += 64) {
      a0 += bf16_to_f32(a[(size_t)s * lda + c]); a1 += bf16_to_f32(a[(size_t)(s + 16) * lda + c]);
      a2 += bf16_to_f32(a[(size_t)(s + 32) * lda + c]); a3 += bf16_to_f32(a[(size_t)(s + 48) * lda + c]);
    }
    for (; s < S; s += 16) a0 += bf16_to_f32(a[(size_t)s * lda + c]);
  }
  red[lane_r][threadIdx.x & 63] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (lane_r == 0 && c < C) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) t += red[r][threadIdx.x];
    out[c] = f32_to_bf16(t);
  }
}
extern "C" int vlaser_colsum_bf16(const void* a, void* out, int S, int C, int lda, vl_stream_t s) {
  VL_CHECK(a && out && S > 0 && C > 0 && lda >= C, "vlaser_colsum_bf16: bad args");
  hipLaunchKernelGGL(colsum_bf16_kernel, dim3((C + 63) / 64), dim3(1024), 0, (hipStream_t)s, (const bf16_t*)a, (bf16_t*)out, S, C, lda);
  VL_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------- SwiGLU fwd / bwd
// packed layout: columns [32j, 32j+16) = gate channels 16j.., [32j+16, 32j+32) = up channels 16j..
__global__ __launch_bounds__(256) void swiglu_kernel(const bf16_t* __restrict__ gu, bf16_t* __restrict__ act, long long n, int I) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const long long s = i / I;
    const int c = (int)(i - s * I);
    const bf16_t* p = gu + s * 2 * I + (c >> 4) * 32 + (c & 15);
    const float g = bf16_to_f32(p[0]), u = bf16_to_f32(p[16]);
    act[i] = f32_to_bf16(round_bf16(silu(g)) * u);
  }
}
__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const bf16_t* __restrict__ gu, const bf16_t* __restrict__ dact, bf16_t* __restrict__ dgu,
                                                         long long n, int I) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const long long s = i / I;
    const int c = (int)(i - s * I);
    const size_t o = s * 2 * I + (c >> 4) * 32 + (c & 15);
    const float g = bf16_to_f32(gu[o]), u = bf16_to_f32(gu[o + 16]), d = bf16_to_f32(dact[i]);
    const float sig = 1.0f / (1.0f + __expf(-g));
    dgu[o] = f32_to_bf16(d * u * sig * (1.0f + g * (1.0f - sig)));
    dgu[o + 16] = f32_to_bf16(d * g * sig);
  }
}
extern "C" int vlaser_swiglu(const void* gu, void* act, int S, int I, vl_stream_t s) {
  VL_CHECK(gu && act && S > 0 && I % 16 == 0, "vlaser_swiglu: bad args");
  hipLaunchKernelGGL(swiglu_kernel, dim3(2048), dim3(256), 0, (hipStream_t)s, (const bf16_t*)gu, (bf16_t*)act, (long long)S * I, I);
  VL_LAUNCH_CHECK();
  return 0;
}
extern "C" int vlaser_swiglu_bwd(const void* gu, const void* dact, void* dgu, int S, int I, vl_stream_t s) {
  VL_CHECK(gu && dact && dgu && S > 0 && I % 16 == 0, "vlaser_swiglu_bwd: bad args");
  hipLaunchKernelGGL(swiglu_bwd_kernel, dim3(2048), dim3(256), 0, (hipStream_t)s, (const bf16_t*)gu, (const bf16_t*)dact, (bf16_t*)dgu, (long long)S * I, I);
  VL_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------- CE backward
__global__ __launch_bounds__(256) void ce_dlogits_kernel(const float* __restrict__ logits, const float* __restrict__ lse, const int64_t* __restrict__ labels,
                                                         bf16_t* __restrict__ out, int V, long long ld_in, int ld_out, float scale, long long ignore) {
  const int r = blockIdx.y;
  const int64_t lab = labels[r];
  const float l = lse[r];
  for (int v = blockIdx.x * 256 + threadIdx.x; v < ld_out; v += gridDim.x * 256) {
    float g = 0.f;
    if (v < V && lab != ignore) g = (__expf(logits[(size_t)r * ld_in + v] - l) - (v == lab ? 1.f : 0.f)) * scale;
    out[(size_t)r * ld_out + v] = f32_to_bf16(g);
  }
}
extern "C" int vlaser_ce_dlogits(const float* logits, const float* lse, const int64_t* labels, void* out, int R, int V, long long ld_in, int ld_out,
                                 float scale, long long ignore, vl_stream_t s) {
  VL_CHECK(logits && lse && labels && out && R > 0 && ld_out >= V, "vlaser_ce_dlogits: bad args");
  hipLaunchKernelGGL(ce_dlogits_kernel, dim3(64, R), dim3(256), 0, (hipStream_t)s, logits, lse, labels, (bf16_t*)out, V, ld_in, ld_out, scale, ignore);
  VL_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------- embedding backward
// dEmbed[id] += sum over the text positions (rank < 0) holding that id of dh[position].  `order` lists the positions sorted by id
// (stable), so equal ids are contiguous runs: every run is summed in fp32 in position order by exactly ONE workgroup (the one whose
// chunk holds the run's first position -- a workgroup skips a leading run that started in the previous chunk and finishes its
// last run past its own chunk end) and rounded to bf16 once, like torch's embedding backward; a token that occurs hundreds of times
// in a 16k-token sample no longer loses its later contributions to bf16 re-rounding.  Ids outside [0, vocab) are skipped.
#define ESA_CHUNK 4       // r04: 64 -> 4 positions per workgroup: a chunk is walked serially (one dependent load per position), 9 workgroups took 100 us for S = 560
__global__ __launch_bounds__(256) void embed_scatter_add_kernel(const int64_t* __restrict__ ids, const int32_t* __restrict__ rank, const int32_t* __restrict__ order,
                                                                const bf16_t* __restrict__ dh, bf16_t* __restrict__ dembed, int n, int H, long long vocab) {
  const int c = (blockIdx.y * 256 + threadIdx.x) * 2;               // two adjacent columns per thread (4-byte accesses)
  if (c >= H) return;
  int i = blockIdx.x * ESA_CHUNK;
  const int end = min(n, i + ESA_CHUNK);
  if (i > 0) {                                                      // a run that began before this chunk belongs to the previous workgroup
    const int64_t prev = ids[order[i - 1]];
    while (i < n && ids[order[i]] == prev) ++i;
  }
  while (i < end) {                                                 // runs that START inside the chunk (may end beyond it)
    const int64_t id = ids[order[i]];
    float a0 = 0.f, a1 = 0.f;
    bool any = false;
    for (; i < n && ids[order[i]] == id; ++i) {
      const int s = order[i];
      if (rank[s] >= 0) continue;                                   // <IMG_CONTEXT> positions take the projector's gradient, not the table's
      const uint32_t v = *reinterpret_cast<const uint32_t*>(dh + (size_t)s * H + c);
      a0 += bf16lo_to_f32(v); a1 += bf16hi_to_f32(v);
      any = true;
    }
    if (any && id >= 0 && id < vocab) {
      uint32_t* p = reinterpret_cast<uint32_t*>(dembed + (size_t)id * H + c);
      const uint32_t o = *p;
      *p = pack_bf16x2(bf16lo_to_f32(o) + a0, bf16hi_to_f32(o) + a1);
    }
  }
}
extern "C" int vlaser_embed_scatter_add(const int64_t* ids, const int32_t* rank, const int32_t* order, const void* dh, void* dembed, int n, int H,
                                        long long vocab, vl_stream_t s) {
  VL_CHECK(ids && rank && order && dh && dembed && n > 0 && H % 2 == 0 && vocab > 0, "vlaser_embed_scatter_add: bad args (order = positions sorted by id)");
  hipLaunchKernelGGL(embed_scatter_add_kernel, dim3((n + ESA_CHUNK - 1) / ESA_CHUNK, (H / 2 + 255) / 256), dim3(256), 0, (hipStream_t)s, ids, rank, order,
                     (const bf16_t*)dh, (bf16_t*)dembed, n, H, vocab);
  VL_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------- GELU backward
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy, bf16_t* __restrict__ dx, long long n) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float v = bf16_to_f32(x[i]);
    const float cdf = 0.5f * (1.0f + erff(v * 0.70710678118654752f));
    const float pdf = 0.3989422804014327f * __expf(-0.5f * v * v);
    dx[i] = f32_to_bf16(bf16_to_f32(dy[i]) * (cdf + v * pdf));
  }
}
extern "C" int vlaser_gelu_bwd(const void* x, const void* dy, void* dx, long long n, vl_stream_t s) {
  VL_CHECK(x && dy && dx && n > 0, "vlaser_gelu_bwd: bad args");
  hipLaunchKernelGGL(gelu_bwd_kernel, dim3(1024), dim3(256), 0, (hipStream_t)s, (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)dx, n);
  VL_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------- fused AdamW
// DeepSpeed FusedAdam (adam_w_mode=1, bias_correction=1) on fp32 master weights; bf16 params refreshed from the master.
__global__ __launch_bounds__(256) void adamw_kernel(bf16_t* __restrict__ p, float* __restrict__ master, float* __restrict__ m, float* __restrict__ v,
                                                    const bf16_t* __restrict__ g, long long n, float lr, float b1, float b2, float eps, float wd,
                                                    float gscale, float bc1, float bc2, const float* __restrict__ gnorm2, float max_norm) {
  const float rbc2 = rsqrtf(bc2), step = lr / bc1, decay = 1.0f - lr * wd;
  if (gnorm2) {          // global-norm clipping with the norm left on the device (torch.nn.utils.clip_grad_norm_: coef = max_norm / (norm + 1e-6), applied when < 1)
    const float gn = sqrtf(gnorm2[0]);
    if (max_norm > 0.f && gn > max_norm) gscale *= max_norm / (gn + 1e-6f);
  }
  const long long n4 = n >> 2;                 // 4 elements per thread per iteration (16-byte fp32 vectors, 8-byte bf16 vectors)
  // every byte is touched exactly once per step: non-temporal loads / stores keep the 28 B/parameter stream out of L2's way
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const u32x2 gv = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(g + 4 * i));
    f32x4 mv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(m + 4 * i)), vv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(v + 4 * i)),
          wv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(master + 4 * i));
    const float gr[4] = {bf16lo_to_f32(gv[0]) * gscale, bf16hi_to_f32(gv[0]) * gscale, bf16lo_to_f32(gv[1]) * gscale, bf16hi_to_f32(gv[1]) * gscale};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      mv[j] = b1 * mv[j] + (1.0f - b1) * gr[j];
      vv[j] = b2 * vv[j] + (1.0f - b2) * gr[j] * gr[j];
      wv[j] = wv[j] * decay - step * (mv[j] / (sqrtf(vv[j]) * rbc2 + eps));
    }
    __builtin_nontemporal_store(mv, reinterpret_cast<f32x4*>(m + 4 * i));
    __builtin_nontemporal_store(vv, reinterpret_cast<f32x4*>(v + 4 * i));
    __builtin_nontemporal_store(wv, reinterpret_cast<f32x4*>(master + 4 * i));
    *reinterpret_cast<u32x2*>(p + 4 * i) = u32x2{pack_bf16x2(wv[0], wv[1]), pack_bf16x2(wv[2], wv[3])};      // the next forward reads these: default policy
  }
  if (blockIdx.x == 0)
    for (long long i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) {
      const float gr = bf16_to_f32(g[i]) * gscale;
      const float mi = b1 * m[i] + (1.0f - b1) * gr, vi = b2 * v[i] + (1.0f - b2) * gr * gr;
      m[i] = mi; v[i] = vi;
      const float w = master[i] * decay - step * (mi / (sqrtf(vi) * rbc2 + eps));
      master[i] = w;
      p[i] = f32_to_bf16(w);
    }
}
// Grid: ONE 256-thread workgroup per CU walking the arrays grid-stride.  Measured on a 198 M-parameter bucket (tools/micro/adamw_lab.py):
// 4096 workgroups 4.4-4.5 TB/s, 1024 4.4, 512 5.2, 256 5.5-5.7 (= 0.9 of the 6.3 TB/s a copy reaches), 192 5.1, 128 3.7; 512- and
// 1024-thread workgroups lose; the seven streams thrash less with a narrow window of addresses in flight.
static int adamw_grid_cap() {
  static int forced = -1;                       // tuning: VLASER_ADAMW_BLOCKS caps the grid (fewer workgroups = less HBM pressure on kernels of other streams)
  if (forced < 0) { const char* e = getenv("VLASER_ADAMW_BLOCKS"); forced = e ? atoi(e) : 0; }
  if (forced > 0) return forced;
  static int cus[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (cus[dev] == 0) {
    int n = 0;
    cus[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
  }
  // r06: on a CU-masked stream (VLASER_DP_EXCHANGE=capi: the step's streams own vlaser_get_cu_budget() CUs) one workgroup per CU THE MASK LEAVES -- 256 workgroups on 248 CUs
  // ran a second round of 8 (step 36.6 ms against 30.2 with 224 CUs, tools/micro/capi_ab.py)
  const int budget = vlaser_get_cu_budget();
  return budget < cus[dev] ? budget : cus[dev];
}
extern "C" int vlaser_adamw(void* p, float* master, float* m, float* v, const void* g, long long n, float lr, float b1, float b2, float eps, float wd,
                            float gscale, int step, vl_stream_t s) {
  VL_CHECK(p && master && m && v && g && n > 0 && step >= 1, "vlaser_adamw: bad args");
  VL_CHECK((((uintptr_t)p | (uintptr_t)g) & 7) == 0 && (((uintptr_t)master | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "vlaser_adamw: alignment");
  const float bc1 = 1.0f - powf(b1, (float)step), bc2 = 1.0f - powf(b2, (float)step);
  const int cap = adamw_grid_cap();
  const int blocks = (int)((n + 1023) / 1024 < cap ? (n + 1023) / 1024 : cap);
  hipLaunchKernelGGL(adamw_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, (bf16_t*)p, master, m, v, (const bf16_t*)g, n, lr, b1, b2, eps, wd,
                     gscale, bc1, bc2, (const float*)nullptr, 0.f);
  VL_LAUNCH_CHECK();
  return 0;
}
extern "C" int vlaser_adamw_clipped(void* p, float* master, float* m, float* v, const void* g, long long n, float lr, float b1, float b2, float eps,
                                    float wd, float gscale, const float* gnorm2, float max_norm, int step, vl_stream_t s) {
  VL_CHECK(p && master && m && v && g && gnorm2 && n > 0 && step >= 1, "vlaser_adamw_clipped: bad args");
  VL_CHECK((((uintptr_t)p | (uintptr_t)g) & 7) == 0 && (((uintptr_t)master | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "vlaser_adamw_clipped: alignment");
  const float bc1 = 1.0f - powf(b1, (float)step), bc2 = 1.0f - powf(b2, (float)step);
  const int cap = adamw_grid_cap();
  const int blocks = (int)((n + 1023) / 1024 < cap ? (n + 1023) / 1024 : cap);
  hipLaunchKernelGGL(adamw_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, (bf16_t*)p, master, m, v, (const bf16_t*)g, n, lr, b1, b2, eps, wd,
                     gscale, bc1, bc2, gnorm2, max_norm);
  VL_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------- gradient accumulation
// acc (fp32) = [acc +] w * g (bf16); with `finalize` the bf16 buffer is overwritten by the rounded sum -- micro-batches of a
// gradient-accumulation step / samples of a per-device batch > 1 (the SFT launcher's PER_DEVICE_BATCH_SIZE x GRADIENT_ACC).
__global__ __launch_bounds__(256) void grad_accumulate_kernel(bf16_t* __restrict__ g, float* __restrict__ acc, long long n, float w, int first, int finalize) {
  const long long n4 = n >> 2;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const u32x2 gv = *reinterpret_cast<const u32x2*>(g + 4 * i);
    f32x4 a = first ? f32x4{0, 0, 0, 0} : *reinterpret_cast<const f32x4*>(acc + 4 * i);
    a[0] += w * bf16lo_to_f32(gv[0]); a[1] += w * bf16hi_to_f32(gv[0]); a[2] += w * bf16lo_to_f32(gv[1]); a[3] += w * bf16hi_to_f32(gv[1]);
    *reinterpret_cast<f32x4*>(acc + 4 * i) = a;
    if (finalize) *reinterpret_cast<u32x2*>(g + 4 * i) = u32x2{pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3])};
  }
  if (blockIdx.x == 0)
    for (long long i = (n4 << 2) + threadIdx.x; i < n; i += 256) {
      const float a = (first ? 0.f : acc[i]) + w * bf16_to_f32(g[i]);
      acc[i] = a;
      if (finalize) g[i] = f32_to_bf16(a);
    }
}
extern "C" int vlaser_grad_accumulate(void* g, float* acc, long long n, float w, int first, int finalize, vl_stream_t s) {
  VL_CHECK(g && acc && n > 0 && (((uintptr_t)g) & 7) == 0 && (((uintptr_t)acc) & 15) == 0, "vlaser_grad_accumulate: bad args / alignment");
  const int blocks = (int)((n + 1023) / 1024 < 4096 ? (n + 1023) / 1024 : 4096);
  hipLaunchKernelGGL(grad_accumulate_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, (bf16_t*)g, acc, n, w, first, finalize);
  VL_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------- sum of squares
__global__ __launch_bounds__(256) void sumsq_kernel(const bf16_t* __restrict__ x, long long n, float* __restrict__ partial) {
  __shared__ float red[4];
  float acc = 0.f;
  const long long n8 = n >> 3;
  const u32x4* xv = reinterpret_cast<const u32x4*>(x);     // 16-byte aligned (flat buffer views are 256-byte aligned)
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
    const u32x4 v = xv[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) { const float lo = bf16lo_to_f32(v[j]), hi = bf16hi_to_f32(v[j]); acc += lo * lo + hi * hi; }
  }
  if (blockIdx.x == 0)
    for (long long i = (n8 << 3) + threadIdx.x; i < n; i += 256) { const float v = bf16_to_f32(x[i]); acc += v * v; }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void sumsq_final_kernel(const float* __restrict__ partial, int n, float* __restrict__ out) {
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += 64) acc += partial[i];
  acc = wave_sum(acc);
  if (threadIdx.x == 0) out[0] += acc;
}
extern "C" int vlaser_sumsq(const void* x, long long n, float* out, float* partial_ws, vl_stream_t s) {
  VL_CHECK(x && out && partial_ws && n > 0 && ((uintptr_t)x & 15) == 0, "vlaser_sumsq: bad args (partial_ws = float[1024] workspace; x 16-byte aligned)");
  // one workgroup per CU (like AdamW: a narrow window of addresses streams faster, and the launch leaves room for a GEMM on another stream)
  const int nb = adamw_grid_cap() < 1024 ? adamw_grid_cap() : 1024;
  hipLaunchKernelGGL(sumsq_kernel, dim3(nb), dim3(256), 0, (hipStream_t)s, (const bf16_t*)x, n, partial_ws);
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)s, partial_ws, nb, out);
  VL_LAUNCH_CHECK();
  return 0;
}

// ---- the gradient norm without a second pass over the gradients (r04; sft.py `_norm_bucket`): the weight-gradient GEMMs leave one partial sum per
// (workgroup, wave) in a slot array (gemm.hip, `sumsq_part`); the small tensors (norm weights, biases, the projector) are summed chunk by chunk, the
// embedding table row by row over the rows this step touched; ONE workgroup then adds a bucket's slots in a fixed order: deterministic, and
// clip_grad_norm_'s sqrt(sum g^2) (internvl_chat_finetune.py:1041-1057, max_grad_norm 1.0) up to fp32 summation order.
// part[c] = sum of squares of x[tab[c][0] .. + tab[c][1]) -- one workgroup per chunk
__global__ __launch_bounds__(256) void sumsq_chunks_kernel(const bf16_t* __restrict__ x, const long long* __restrict__ tab, float* __restrict__ part) {
  __shared__ float red[4];
  const long long off = tab[2 * blockIdx.x], n = tab[2 * blockIdx.x + 1];
  const bf16_t* p = x + off;
  float acc = 0.f;
  if (((uintptr_t)p & 15) == 0) {
    const long long n8 = n >> 3;
    for (long long i = threadIdx.x; i < n8; i += 256) {
      const u32x4 v = reinterpret_cast<const u32x4*>(p)[i];
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float lo = bf16lo_to_f32(v[j]), hi = bf16hi_to_f32(v[j]); acc += lo * lo + hi * hi; }
    }
    for (long long i = (n8 << 3) + threadIdx.x; i < n; i += 256) { const float v = bf16_to_f32(p[i]); acc += v * v; }
  } else {
    for (long long i = threadIdx.x; i < n; i += 256) { const float v = bf16_to_f32(p[i]); acc += v * v; }
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
extern "C" int vlaser_sumsq_chunks(const void* x, const long long* tab, int n_chunks, float* part, vl_stream_t s) {
  VL_CHECK(x && tab && part && n_chunks > 0, "vlaser_sumsq_chunks: bad args (tab = int64 [n_chunks][2] = element offset, length; part = float[n_chunks])");
  hipLaunchKernelGGL(sumsq_chunks_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)s, (const bf16_t*)x, tab, part);
  VL_LAUNCH_CHECK();
  return 0;
}
// the embedding gradient after vlaser_embed_scatter_add: position i of the id-sorted order that STARTS a run of equal ids owns that id's row
// (a row no text position wrote is still zero from the step's clear); every other slot of part[0 .. cap) is set to 0.  H % 8 == 0.
__global__ __launch_bounds__(256) void sumsq_rows_kernel(const int64_t* __restrict__ ids, const int32_t* __restrict__ order, const bf16_t* __restrict__ x, int n, int H,
                                                         long long vocab, float* __restrict__ part) {
  __shared__ float red[4];
  const int i = blockIdx.x;
  float acc = 0.f;
  if (i < n) {
    const int64_t id = ids[order[i]];
    const bool head = (i == 0 || ids[order[i - 1]] != id) && id >= 0 && id < vocab;
    if (head) {
      const u32x4* p = reinterpret_cast<const u32x4*>(x + (size_t)id * H);
      for (int c = threadIdx.x; c < H / 8; c += 256) {
        const u32x4 v = p[c];
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float lo = bf16lo_to_f32(v[j]), hi = bf16hi_to_f32(v[j]); acc += lo * lo + hi * hi; }
      }
    }
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[i] = red[0] + red[1] + red[2] + red[3];
}
extern "C" int vlaser_sumsq_rows(const int64_t* ids, const int32_t* order, const void* x, int n, int H, long long vocab, float* part, int cap, vl_stream_t s) {
  VL_CHECK(ids && order && x && part && n > 0 && cap >= n && H % 8 == 0 && vocab > 0 && ((uintptr_t)x & 15) == 0,
           "vlaser_sumsq_rows: bad args (order = the positions sorted by id, as for vlaser_embed_scatter_add; part = float[cap], cap >= n; H %% 8 == 0)");
  hipLaunchKernelGGL(sumsq_rows_kernel, dim3(cap), dim3(256), 0, (hipStream_t)s, ids, order, (const bf16_t*)x, n, H, vocab, part);
  VL_LAUNCH_CHECK();
  return 0;
}
// out[0] (+)= part[0] + part[1] + ... in a fixed association: thread t owns the 16-byte pieces t, t + 1024, ... (four independent running sums, four
// pieces requested per round: 34.7 -> ~6 us for a 4-layer bucket's 91 K slots -- the one-load-per-round version was a chain of 89 round trips), then a
// fixed tree over the lanes and waves
__global__ __launch_bounds__(1024) void sum_partials_kernel(const float* __restrict__ part, long long n, float* __restrict__ out, int accumulate) {
  __shared__ float red[16];
  f32x4 a = {0.f, 0.f, 0.f, 0.f};
  const bool vec = ((uintptr_t)part & 15) == 0;
  const long long n4 = vec ? (n >> 2) : 0;
  const f32x4* p4 = reinterpret_cast<const f32x4*>(part);
  long long i = threadIdx.x;
  for (; i + 3 * 1024 < n4; i += 4 * 1024) {
    const f32x4 v0 = p4[i], v1 = p4[i + 1024], v2 = p4[i + 2048], v3 = p4[i + 3072];
    a += v0; a += v1; a += v2; a += v3;
  }
  for (; i < n4; i += 1024) a += p4[i];
  float acc = (a[0] + a[1]) + (a[2] + a[3]);
  for (long long j = (n4 << 2) + threadIdx.x; j < n; j += 1024) acc += part[j];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += red[w];
    out[0] = accumulate ? out[0] + t : t;
  }
}
extern "C" int vlaser_sum_partials(const float* part, long long n, float* out, int accumulate, vl_stream_t s) {
  VL_CHECK(part && out && n > 0, "vlaser_sum_partials: bad args");
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(1024), 0, (hipStream_t)s, part, n, out, accumulate);
  VL_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------- SiLU (ActionEncoder.linear_2, modules.py:45-52)
__global__ __launch_bounds__(256) void silu_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, long long n) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) y[i] = f32_to_bf16(silu(bf16_to_f32(x[i])));
}
__global__ __launch_bounds__(256) void silu_bwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy, bf16_t* __restrict__ dx, long long n) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float v = bf16_to_f32(x[i]), sg = 1.0f / (1.0f + __expf(-v));
    dx[i] = f32_to_bf16(bf16_to_f32(dy[i]) * sg * (1.0f + v * (1.0f - sg)));
  }
}
extern "C" int vlaser_silu(const void* x, void* y, long long n, vl_stream_t s) {
  VL_CHECK(x && y && n > 0, "vlaser_silu: bad args");
  hipLaunchKernelGGL(silu_fwd_kernel, dim3((unsigned)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024)), dim3(256), 0, (hipStream_t)s, (const bf16_t*)x, (bf16_t*)y, n);
  VL_LAUNCH_CHECK();
  return 0;
}
extern "C" int vlaser_silu_bwd(const void* x, const void* dy, void* dx, long long n, vl_stream_t s) {
  VL_CHECK(x && dy && dx && n > 0, "vlaser_silu_bwd: bad args");
  hipLaunchKernelGGL(silu_bwd_kernel, dim3((unsigned)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024)), dim3(256), 0, (hipStream_t)s, (const bf16_t*)x, (const bf16_t*)dy,
                     (bf16_t*)dx, n);
  VL_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------- attention backward of a few query rows over a KV cache
// The expert rows of the VLA flow-matching training step (pizero_internvl.py:1064-1197; joint attention of joint_model.py:410-696):
// R <= 16 query rows (proprio + action tokens, cache slots [blk_start, blk_start + R)) attend to the frozen VLM prefix [0, valid_len)
// and to their own block (row 0 -- the proprio token -- only to itself when first_tok_self is set).  One workgroup per kv head walks
// its G query heads x R rows in a fixed order (deterministic, no atomics): scores -> softmax -> dP = dO V^T -> dS = P o (dP - <dO, O>)
// -> dQ = dS K; dK / dV are accumulated for the block keys only (the prefix keys belong to the frozen VLM: no gradient is needed).
// K [n_kv, S_max, 128] (post-RoPE), V^T [n_kv, 128, S_max]; q / dO / O / dq bf16 [R, n_q*128]; dk / dv bf16 [R, n_kv*128].
#define ARB_MAXKEYS 2048
// r06: one workgroup per (query head, row) -- 60 workgroups for 12 heads x 5 rows -- instead of one per kv head walking its G x R (head, row) pairs one after the other
// (2 workgroups, 30 sequential passes of ~35 us: 1.04 ms per layer, 29 of the 47 ms of the action-expert training step; profiles/r06aa_vla_train_kernel_stats.md).  The block
// keys' P and dS (R x R per head) go to a small fp32 workspace; attn_rows_bwd_kv_kernel sums dK / dV of the block keys from them in the old fixed order (head group outer,
// row inner): deterministic, no atomics.
__global__ __launch_bounds__(256) void attn_rows_bwd_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ K, const bf16_t* __restrict__ VT,
                                                            const bf16_t* __restrict__ dO, const bf16_t* __restrict__ O, bf16_t* __restrict__ dq,
                                                            float* __restrict__ ws, int R, int n_q, int n_kv, int s_max,
                                                            int valid_len, int blk_start, int first_tok_self, float scale,
                                                            bf16_t* __restrict__ p_out, bf16_t* __restrict__ ds_out) {
  // p_out / ds_out (optional, bf16 [n_q][16][s_max]): softmax probabilities and dS of every key, for the dK / dV of the PREFIX keys when the VLM
  // is trained too (train_vlm: True): dK[kvh] = sum_{g, r} dS[kvh G + g][r]^T q[r, head], dV likewise from P and dO (vlaser_gemm_tn_grouped)
  __shared__ float sc[ARB_MAXKEYS];       // p, then dS
  __shared__ float qs[128], dos[128], red[8];
  const int h = blockIdx.x, r = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int G = n_q / n_kv, kvh = h / G, kv_len = blk_start + R;
  const bf16_t* Kh = K + (size_t)kvh * s_max * 128;
  const bf16_t* Vh = VT + (size_t)kvh * 128 * s_max;
  float* p_blk = ws + ((size_t)h * 16 + r) * 16;                       // [n_q][16][16] P of the block keys, then the same for dS
  float* ds_blk = ws + (size_t)n_q * 256 + ((size_t)h * 16 + r) * 16;
  auto block_sum = [&](float v) {
    v = wave_sum(v);
    if (lane == 0) red[wave] = v;
    __syncthreads();
    const float t = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    return t;
  };
  auto block_max = [&](float v) {
    v = wave_max(v);
    if (lane == 0) red[wave] = v;
    __syncthreads();
    const float t = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    return t;
  };
  const size_t ro = (size_t)r * n_q * 128 + (size_t)h * 128;
  if (tid < 128) { qs[tid] = bf16_to_f32(q[ro + tid]); dos[tid] = bf16_to_f32(dO[ro + tid]); }
  __syncthreads();
  const int hi2 = (r == 0 && first_tok_self) ? blk_start + 1 : kv_len;
  float mx = -3.0e38f;
  for (int j = tid; j < kv_len; j += 256) {
    const bool vis = j < valid_len || (j >= blk_start && j < hi2);
    float s = -3.0e38f;
    if (vis) {
      float a = 0.f;
      const bf16_t* kr = Kh + (size_t)j * 128;
#pragma unroll
      for (int d8 = 0; d8 < 16; ++d8) {
        const u32x4 kv = *reinterpret_cast<const u32x4*>(kr + d8 * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) { a += qs[d8 * 8 + 2 * e] * bf16lo_to_f32(kv[e]); a += qs[d8 * 8 + 2 * e + 1] * bf16hi_to_f32(kv[e]); }
      }
      s = a * scale;
    }
    sc[j] = s;
    mx = fmaxf(mx, s);
  }
  mx = block_max(mx);
  float sum = 0.f;
  for (int j = tid; j < kv_len; j += 256) {
    const float p = sc[j] > -1.0e38f ? __expf(sc[j] - mx) : 0.f;
    sc[j] = p;
    sum += p;
  }
  sum = block_sum(sum);
  const float inv = 1.0f / sum;
  // D = <dO, O>; dP_j = dO . V_j; dS_j = p_j (dP_j - D) * scale
  float dd = 0.f;
  if (tid < 128) dd = dos[tid] * bf16_to_f32(O[ro + tid]);
  const float D = block_sum(dd);
  for (int j = tid; j < kv_len; j += 256) {
    const float p = sc[j] * inv;
    float ds = 0.f;
    if (p > 0.f) {
      float dp = 0.f;
#pragma unroll 8
      for (int d = 0; d < 128; ++d) dp += dos[d] * bf16_to_f32(Vh[(size_t)d * s_max + j]);
      ds = p * (dp - D) * scale;
    }
    if (j >= blk_start) { p_blk[j - blk_start] = p; ds_blk[j - blk_start] = ds; }      // block keys: dV_j += p dO, dK_j += dS q in attn_rows_bwd_kv_kernel
    sc[j] = ds;
    if (p_out) {
      p_out[((size_t)h * 16 + r) * s_max + j] = f32_to_bf16(p);
      ds_out[((size_t)h * 16 + r) * s_max + j] = f32_to_bf16(ds);
    }
  }
  __syncthreads();
  // dQ[d] = sum_j dS_j K_j[d]: two threads per d, each half of the keys
  {
    const int d = tid & 127, half = tid >> 7;
    float a = 0.f;
    for (int j = half; j < kv_len; j += 2) a += sc[j] * bf16_to_f32(Kh[(size_t)j * 128 + d]);
    if (half == 1) qs[d] = a;                       // qs is free now
    __syncthreads();
    if (half == 0) dq[ro + d] = f32_to_bf16(a + qs[d]);
  }
}
// dK / dV of the R block keys of one kv head: sum over its G query heads (outer) and the R query rows (inner) of dS q / P dO -- the order the r03-r05 kernel accumulated in
__global__ __launch_bounds__(128) void attn_rows_bwd_kv_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ dO, const float* __restrict__ ws, bf16_t* __restrict__ dk,
                                                               bf16_t* __restrict__ dv, int R, int n_q, int n_kv) {
  const int kvh = blockIdx.x, jb = blockIdx.y, d = threadIdx.x, G = n_q / n_kv;
  float ak = 0.f, av = 0.f;
  for (int hg = 0; hg < G; ++hg) {
    const int h = kvh * G + hg;
    for (int r = 0; r < R; ++r) {
      const size_t ro = (size_t)r * n_q * 128 + (size_t)h * 128 + d;
      const float p = ws[((size_t)h * 16 + r) * 16 + jb], ds = ws[(size_t)n_q * 256 + ((size_t)h * 16 + r) * 16 + jb];
      av += p * bf16_to_f32(dO[ro]);
      ak += ds * bf16_to_f32(q[ro]);
    }
  }
  dk[(size_t)jb * n_kv * 128 + kvh * 128 + d] = f32_to_bf16(ak);
  dv[(size_t)jb * n_kv * 128 + kvh * 128 + d] = f32_to_bf16(av);
}
extern "C" int vlaser_attn_rows_bwd_ws_floats(int n_q) { return n_q > 0 ? 2 * n_q * 256 : -1; }
extern "C" int vlaser_attn_rows_bwd_ex(const void* q, const void* K, const void* VT, const void* dO, const void* O, void* dq, void* dk, void* dv, int R, int n_q,
                                       int n_kv, int s_max, int valid_len, int blk_start, int first_tok_self, float scale, void* p_out, void* ds_out, float* ws,
                                       vl_stream_t s);
extern "C" int vlaser_attn_rows_bwd(const void* q, const void* K, const void* VT, const void* dO, const void* O, void* dq, void* dk, void* dv, int R, int n_q,
                                    int n_kv, int s_max, int valid_len, int blk_start, int first_tok_self, float scale, float* ws, vl_stream_t s) {
  return vlaser_attn_rows_bwd_ex(q, K, VT, dO, O, dq, dk, dv, R, n_q, n_kv, s_max, valid_len, blk_start, first_tok_self, scale, nullptr, nullptr, ws, s);
}
extern "C" int vlaser_attn_rows_bwd_ex(const void* q, const void* K, const void* VT, const void* dO, const void* O, void* dq, void* dk, void* dv, int R, int n_q,
                                       int n_kv, int s_max, int valid_len, int blk_start, int first_tok_self, float scale, void* p_out, void* ds_out, float* ws,
                                       vl_stream_t s) {
  VL_CHECK(q && K && VT && dO && O && dq && dk && dv && ws && (!p_out == !ds_out), "vlaser_attn_rows_bwd: null pointer (ws = vlaser_attn_rows_bwd_ws_floats(n_q) floats)");
  VL_CHECK(R >= 1 && R <= 16 && n_q % n_kv == 0 && blk_start + R <= s_max && blk_start + R <= ARB_MAXKEYS && valid_len <= blk_start,
           "vlaser_attn_rows_bwd: bad geometry (R <= 16, kv_len <= %d)", ARB_MAXKEYS);
  VL_CHECK((((uintptr_t)K) & 15) == 0 && s_max % 8 == 0, "vlaser_attn_rows_bwd: the K cache must be 16-byte aligned");
  hipLaunchKernelGGL(attn_rows_bwd_kernel, dim3(n_q, R), dim3(256), 0, (hipStream_t)s, (const bf16_t*)q, (const bf16_t*)K, (const bf16_t*)VT, (const bf16_t*)dO,
                     (const bf16_t*)O, (bf16_t*)dq, ws, R, n_q, n_kv, s_max, valid_len, blk_start, first_tok_self, scale, (bf16_t*)p_out, (bf16_t*)ds_out);
  VL_LAUNCH_CHECK();
  hipLaunchKernelGGL(attn_rows_bwd_kv_kernel, dim3(n_kv, R), dim3(128), 0, (hipStream_t)s, (const bf16_t*)q, (const bf16_t*)dO, ws, (bf16_t*)dk, (bf16_t*)dv, R, n_q, n_kv);
  VL_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------- LayerNorm backward (ViT blocks, f1: train_vlm)
// y = gamma * xhat + beta, xhat = (x - mean) rs.  g = gamma dy;  dx = rs * (g - mean(g) - xhat * mean(g xhat));  dx_out = dres + dx.
// One wave per row (C <= 8192).  The affine gradients are column sums over the rows: vlaser_colsum_mul mode 3 (gamma) and mode 0 (beta).
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                            const bf16_t* __restrict__ dres, bf16_t* __restrict__ dx, int S, int C, float eps) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= S) return;
  const size_t ro = (size_t)row * C;
  float sx = 0.f, sxx = 0.f;
  for (int c = lane * 8; c < C; c += 512) {
    const u32x4 xv = ld_global_16(x + ro + c);
#pragma unroll
    for (int j = 0; j < 4; ++j) { const float a = bf16lo_to_f32(xv[j]), b = bf16hi_to_f32(xv[j]); sx += a + b; sxx += a * a + b * b; }
  }
  sx = wave_sum(sx); sxx = wave_sum(sxx);
  const float mean = sx / (float)C;
  const float rs = rsqrtf(fmaxf(sxx / (float)C - mean * mean, 0.f) + eps);
  float sg = 0.f, sgx = 0.f;
  for (int c = lane * 8; c < C; c += 512) {
    const u32x4 xv = ld_global_16(x + ro + c), gv = ld_global_16(dy + ro + c), wv = ld_global_16(w + c);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float g0 = bf16lo_to_f32(gv[j]) * bf16lo_to_f32(wv[j]), g1 = bf16hi_to_f32(gv[j]) * bf16hi_to_f32(wv[j]);
      sg += g0 + g1;
      sgx += g0 * (bf16lo_to_f32(xv[j]) - mean) * rs + g1 * (bf16hi_to_f32(xv[j]) - mean) * rs;
    }
  }
  sg = wave_sum(sg) / (float)C; sgx = wave_sum(sgx) / (float)C;
  for (int c = lane * 8; c < C; c += 512) {
    const u32x4 xv = ld_global_16(x + ro + c), gv = ld_global_16(dy + ro + c), wv = ld_global_16(w + c);
    u32x4 rv = {0, 0, 0, 0};
    if (dres) rv = ld_global_16(dres + ro + c);
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float g0 = bf16lo_to_f32(gv[j]) * bf16lo_to_f32(wv[j]), g1 = bf16hi_to_f32(gv[j]) * bf16hi_to_f32(wv[j]);
      const float lo = bf16lo_to_f32(rv[j]) + rs * (g0 - sg - (bf16lo_to_f32(xv[j]) - mean) * rs * sgx);
      const float hi = bf16hi_to_f32(rv[j]) + rs * (g1 - sg - (bf16hi_to_f32(xv[j]) - mean) * rs * sgx);
      o[j] = pack_bf16x2(lo, hi);
    }
    st_global_16(dx + ro + c, o);
  }
}
extern "C" int vlaser_layernorm_bwd(const void* dy, const void* x, const void* w, const void* dres, void* dx_out, int S, int C, float eps, vl_stream_t s) {
  VL_CHECK(dy && x && w && dx_out && S > 0 && C % 8 == 0, "vlaser_layernorm_bwd: bad args");
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3((S + 3) / 4), dim3(256), 0, (hipStream_t)s, (const bf16_t*)dy, (const bf16_t*)x, (const bf16_t*)w,
                     (const bf16_t*)dres, (bf16_t*)dx_out, S, C, eps);
  VL_LAUNCH_CHECK();
  return 0;
}

// out[s, c] = x[s, c] * alpha * (vec ? vec[c] : 1): the layer-scale factor of a ViT residual branch in its backward (dy = ls o dh,
// modeling_intern_vit.py:291-293) and the 1 / 8 the ViT's q carries (VL_EPI_VIT_QKV stores q * scale).  In place allowed.
__global__ __launch_bounds__(256) void scale_cols_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ vec, bf16_t* __restrict__ out, size_t n8, int C,
                                                         int ldx, int ldo, float alpha) {
  const int c8 = C >> 3;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
    const size_t row = i / c8;
    const int c = (int)(i - row * c8) << 3;
    const u32x4 xv = ld_global_16(x + row * ldx + c);
    u32x4 vv = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};       // bf16 1.0 pairs
    if (vec) vv = ld_global_16(vec + c);
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = pack_bf16x2(bf16lo_to_f32(xv[j]) * bf16lo_to_f32(vv[j]) * alpha, bf16hi_to_f32(xv[j]) * bf16hi_to_f32(vv[j]) * alpha);
    st_global_16(out + row * ldo + c, o);
  }
}
extern "C" int vlaser_scale_cols(const void* x, const void* vec, void* out, int S, int C, int ldx, int ldo, float alpha, vl_stream_t s) {
  VL_CHECK(x && out && S > 0 && C % 8 == 0 && ldx % 8 == 0 && ldo % 8 == 0, "vlaser_scale_cols: bad args");
  const size_t n8 = (size_t)S * (C >> 3);
  const int blocks = (int)((n8 + 255) / 256 < 2048 ? (n8 + 255) / 256 : 2048);
  hipLaunchKernelGGL(scale_cols_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, (const bf16_t*)x, (const bf16_t*)vec, (bf16_t*)out, n8, C, ldx, ldo, alpha);
  VL_LAUNCH_CHECK();
  return 0;
}
