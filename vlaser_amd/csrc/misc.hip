// Memory-bound helper kernels of the Vlaser forward path (gfx950): norms, patch-embed im2col, ViT token
// assembly, pixel-shuffle + LayerNorm gather, embedding / visual-token scatter, argmax, and the tiny pi0 head
// (time embedding, action encoder input, final norm + action decoder + Euler update).
// All are one-wave-per-row or one-block kernels with 16-byte vector accesses and fp32 statistics.
#include "common.h"
#include "../../include/vlaser_hip.h"

// ---------------------------------------------------------------------------------------------- LayerNorm / RMSNorm
// one wave per row, 4 rows per block; C % 8 == 0.
template <bool RMS>
__global__ __launch_bounds__(256) void norm_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                   const bf16_t* __restrict__ bias, bf16_t* __restrict__ out, int rows,
                                                   int C, float eps) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const bf16_t* xr = x + (size_t)row * C;
  float s = 0.f, ss = 0.f;
  for (int c = lane * 8; c < C; c += 512) {
    const u32x4 v = ld_global_16(xr + c);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float lo = bf16lo_to_f32(v[j]), hi = bf16hi_to_f32(v[j]);
      s += lo + hi;
      ss += lo * lo + hi * hi;
    }
  }
  s = wave_sum(s);
  ss = wave_sum(ss);
  float mean = 0.f, rs;
  if constexpr (RMS) {
    rs = rsqrtf(ss / (float)C + eps);
  } else {
    mean = s / (float)C;
    // two-pass variance for accuracy (row is L1/L2 resident)
    float vs = 0.f;
    for (int c = lane * 8; c < C; c += 512) {
      const u32x4 v = ld_global_16(xr + c);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float lo = bf16lo_to_f32(v[j]) - mean, hi = bf16hi_to_f32(v[j]) - mean;
        vs += lo * lo + hi * hi;
      }
    }
    vs = wave_sum(vs);
    rs = rsqrtf(vs / (float)C + eps);
  }
  bf16_t* orow = out + (size_t)row * C;
  for (int c = lane * 8; c < C; c += 512) {
    const u32x4 v = ld_global_16(xr + c), wv = ld_global_16(w + c);
    u32x4 bv = {0, 0, 0, 0};
    if constexpr (!RMS) bv = ld_global_16(bias + c);
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float lo, hi;
      if constexpr (RMS) {  // Qwen2RMSNorm: w * bf16(x * rs)
        lo = round_bf16(bf16lo_to_f32(v[j]) * rs) * bf16lo_to_f32(wv[j]);
        hi = round_bf16(bf16hi_to_f32(v[j]) * rs) * bf16hi_to_f32(wv[j]);
      } else {
        lo = (bf16lo_to_f32(v[j]) - mean) * rs * bf16lo_to_f32(wv[j]) + bf16lo_to_f32(bv[j]);
        hi = (bf16hi_to_f32(v[j]) - mean) * rs * bf16hi_to_f32(wv[j]) + bf16hi_to_f32(bv[j]);
      }
      o[j] = pack_bf16x2(lo, hi);
    }
    st_global_16(orow + c, o);
  }
}

extern "C" int vlaser_layernorm(const void* x, const void* w, const void* b, void* out, int rows, int C, float eps, vl_stream_t s) {
  VL_CHECK(x && w && b && out && rows > 0 && C % 8 == 0, "vlaser_layernorm: bad args");
  hipLaunchKernelGGL(norm_kernel<false>, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)s, (const bf16_t*)x, (const bf16_t*)w,
                     (const bf16_t*)b, (bf16_t*)out, rows, C, eps);
  VL_LAUNCH_CHECK();
  return 0;
}
extern "C" int vlaser_rmsnorm(const void* x, const void* w, void* out, int rows, int C, float eps, vl_stream_t s) {
  VL_CHECK(x && w && out && rows > 0 && C % 8 == 0, "vlaser_rmsnorm: bad args");
  hipLaunchKernelGGL(norm_kernel<true>, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)s, (const bf16_t*)x, (const bf16_t*)w,
                     (const bf16_t*)nullptr, (bf16_t*)out, rows, C, eps);
  VL_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------- patch embed
// im2col for Conv2d(3->C, k=s=14): A[t*G*G + py*G + px][c*196 + ky*14 + kx] = pix[t][c][py*14+ky][px*14+kx],
// zero padded from 588 to Kpad columns.  One block per patch row group.
__global__ __launch_bounds__(256) void im2col_kernel(const bf16_t* __restrict__ pix, bf16_t* __restrict__ A, int T, int img,
                                                     int G, int Kpad) {
  const int patch = blockIdx.x;  // t*G*G + py*G + px
  const int t = patch / (G * G), pp = patch - t * G * G, py = pp / G, px = pp - py * G;
  bf16_t* dst = A + (size_t)patch * Kpad;
  for (int k = threadIdx.x; k < Kpad; k += 256) {
    bf16_t v = 0;
    if (k < 588) {
      const int c = k / 196, r = k - c * 196, ky = r / 14, kx = r - ky * 14;
      v = pix[(((size_t)t * 3 + c) * img + py * 14 + ky) * img + px * 14 + kx];
    }
    dst[k] = v;
  }
}
extern "C" int vlaser_im2col(const void* pix, void* A, int T, int img, int Kpad, vl_stream_t s) {
  VL_CHECK(pix && A && T > 0 && img % 14 == 0 && Kpad >= 588, "vlaser_im2col: bad args");
  const int G = img / 14;
  hipLaunchKernelGGL(im2col_kernel, dim3(T * G * G), dim3(256), 0, (hipStream_t)s, (const bf16_t*)pix, (bf16_t*)A, T, img, G, Kpad);
  VL_LAUNCH_CHECK();
  return 0;
}

// h[t,0,:] = cls + pos[0]; h[t,1+p,:] = patch[t*P+p,:] + pos[1+p]   (modeling_intern_vit.py:166-174)
__global__ __launch_bounds__(128) void vit_assemble_kernel(const bf16_t* __restrict__ patch, const bf16_t* __restrict__ cls,
                                                           const bf16_t* __restrict__ pos, bf16_t* __restrict__ h, int P, int C) {
  const int row = blockIdx.x;  // t*(P+1) + s
  const int t = row / (P + 1), sidx = row - t * (P + 1);
  const bf16_t* src = sidx == 0 ? cls : patch + ((size_t)t * P + sidx - 1) * C;
  for (int c = threadIdx.x * 8; c < C; c += 128 * 8) {
    const u32x4 a = ld_global_16(src + c), b = ld_global_16(pos + (size_t)sidx * C + c);
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = pack_bf16x2(bf16lo_to_f32(a[j]) + bf16lo_to_f32(b[j]), bf16hi_to_f32(a[j]) + bf16hi_to_f32(b[j]));
    st_global_16(h + (size_t)row * C + c, o);
  }
}
extern "C" int vlaser_vit_assemble(const void* patch, const void* cls, const void* pos, void* h, int T, int P, int C, vl_stream_t s) {
  VL_CHECK(patch && cls && pos && h && C % 8 == 0, "vlaser_vit_assemble: bad args");
  hipLaunchKernelGGL(vit_assemble_kernel, dim3(T * (P + 1)), dim3(128), 0, (hipStream_t)s, (const bf16_t*)patch, (const bf16_t*)cls,
                     (const bf16_t*)pos, (bf16_t*)h, P, C);
  VL_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------- pixel shuffle + LN
// out[t, i*G2+j, a*2C + b*C + k] = LN_{4C}( x[t, 1 + (2i+a)*G + (2j+b), k] )   (ps_version v2,
// modeling_internvl_chat.py:257-271,284-290; CLS row dropped).  One wave per output token, 4 per block.
__global__ __launch_bounds__(256) void pixel_shuffle_ln_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                               const bf16_t* __restrict__ bias, bf16_t* __restrict__ out, int T,
                                                               int G, int C, float eps, int transpose_v1) {
  const int G2 = G / 2, lane = threadIdx.x & 63;
  const int tok = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tok >= T * G2 * G2) return;
  const int t = tok / (G2 * G2), r = tok - t * G2 * G2;
  int i = r / G2, j = r - i * G2;
  if (transpose_v1) { const int tmp = i; i = j; j = tmp; }
  const bf16_t* base = x + (size_t)t * (G * G + 1) * C;
  const int C4 = 4 * C;
  // the 4 source rows
  const bf16_t* src[4];
#pragma unroll
  for (int ab = 0; ab < 4; ++ab) src[ab] = base + (size_t)(1 + (2 * i + (ab >> 1)) * G + (2 * j + (ab & 1))) * C;
  float s = 0.f;
  for (int c = lane * 8; c < C4; c += 512) {
    const u32x4 v = ld_global_16(src[c / C] + (c % C));
#pragma unroll
    for (int q = 0; q < 4; ++q) s += bf16lo_to_f32(v[q]) + bf16hi_to_f32(v[q]);
  }
  const float mean = wave_sum(s) / (float)C4;
  float vs = 0.f;
  for (int c = lane * 8; c < C4; c += 512) {
    const u32x4 v = ld_global_16(src[c / C] + (c % C));
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float lo = bf16lo_to_f32(v[q]) - mean, hi = bf16hi_to_f32(v[q]) - mean;
      vs += lo * lo + hi * hi;
    }
  }
  const float rs = rsqrtf(wave_sum(vs) / (float)C4 + eps);
  bf16_t* orow = out + (size_t)tok * C4;
  for (int c = lane * 8; c < C4; c += 512) {
    const u32x4 v = ld_global_16(src[c / C] + (c % C)), wv = ld_global_16(w + c), bv = ld_global_16(bias + c);
    u32x4 o;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      o[q] = pack_bf16x2((bf16lo_to_f32(v[q]) - mean) * rs * bf16lo_to_f32(wv[q]) + bf16lo_to_f32(bv[q]),
                         (bf16hi_to_f32(v[q]) - mean) * rs * bf16hi_to_f32(wv[q]) + bf16hi_to_f32(bv[q]));
    st_global_16(orow + c, o);
  }
}
extern "C" int vlaser_pixel_shuffle_ln(const void* x, const void* w, const void* b, void* out, int T, int G, int C, float eps,
                                       int ps_v1, vl_stream_t s) {
  VL_CHECK(x && w && b && out && G % 2 == 0 && C % 8 == 0, "vlaser_pixel_shuffle_ln: bad args");
  const int toks = T * (G / 2) * (G / 2);
  hipLaunchKernelGGL(pixel_shuffle_ln_kernel, dim3((toks + 3) / 4), dim3(256), 0, (hipStream_t)s, (const bf16_t*)x, (const bf16_t*)w,
                     (const bf16_t*)b, (bf16_t*)out, T, G, C, eps, ps_v1);
  VL_LAUNCH_CHECK();
  return 0;
}
// pure permutation variant (bit-exact data movement check of a5): out = pixel_shuffle(x[:,1:])
__global__ __launch_bounds__(256) void pixel_shuffle_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out, int T, int G, int C,
                                                            int transpose_v1) {
  const int G2 = G / 2;
  const int tok = blockIdx.x;
  const int t = tok / (G2 * G2), r = tok - t * G2 * G2;
  int i = r / G2, j = r - i * G2;
  if (transpose_v1) { const int tmp = i; i = j; j = tmp; }
  const bf16_t* base = x + (size_t)t * (G * G + 1) * C;
  for (int c = threadIdx.x * 8; c < 4 * C; c += 256 * 8) {
    const int ab = c / C;
    const bf16_t* src = base + (size_t)(1 + (2 * i + (ab >> 1)) * G + (2 * j + (ab & 1))) * C + (c % C);
    st_global_16(out + (size_t)tok * 4 * C + c, ld_global_16(src));
  }
}
extern "C" int vlaser_pixel_shuffle(const void* x, void* out, int T, int G, int C, int ps_v1, vl_stream_t s) {
  VL_CHECK(x && out && G % 2 == 0 && C % 8 == 0, "vlaser_pixel_shuffle: bad args");
  hipLaunchKernelGGL(pixel_shuffle_kernel, dim3(T * (G / 2) * (G / 2)), dim3(256), 0, (hipStream_t)s, (const bf16_t*)x, (bf16_t*)out, T,
                     G, C, ps_v1);
  VL_LAUNCH_CHECK();
  return 0;
}

// backward of pixel_shuffle(x[:, 1:]) (f1, train_vlm): the same permutation run the other way, dx [T, G*G+1, C] (CLS row zero) from dout [T*(G/2)^2, 4C]
__global__ __launch_bounds__(256) void pixel_unshuffle_kernel(const bf16_t* __restrict__ dout, bf16_t* __restrict__ dx, int T, int G, int C, int transpose_v1) {
  const int G2 = G / 2;
  const int tok = blockIdx.x;
  const int t = tok / (G2 * G2), r = tok - t * G2 * G2;
  int i = r / G2, j = r - i * G2;
  if (transpose_v1) { const int tmp = i; i = j; j = tmp; }
  bf16_t* base = dx + (size_t)t * (G * G + 1) * C;
  if (r == 0)
    for (int c = threadIdx.x * 8; c < C; c += 256 * 8) st_global_16(base + c, u32x4{0, 0, 0, 0});       // CLS token: dropped by extract_feature (:284)
  for (int c = threadIdx.x * 8; c < 4 * C; c += 256 * 8) {
    const int ab = c / C;
    bf16_t* dst = base + (size_t)(1 + (2 * i + (ab >> 1)) * G + (2 * j + (ab & 1))) * C + (c % C);
    st_global_16(dst, ld_global_16(dout + (size_t)tok * 4 * C + c));
  }
}
extern "C" int vlaser_pixel_unshuffle(const void* dout, void* dx, int T, int G, int C, int ps_v1, vl_stream_t s) {
  VL_CHECK(dout && dx && G % 2 == 0 && C % 8 == 0, "vlaser_pixel_unshuffle: bad args");
  hipLaunchKernelGGL(pixel_unshuffle_kernel, dim3(T * (G / 2) * (G / 2)), dim3(256), 0, (hipStream_t)s, (const bf16_t*)dout, (bf16_t*)dx, T, G, C, ps_v1);
  VL_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------- embedding + scatter
// rank[s] = number of <IMG_CONTEXT> tokens before position s (row-major over [B,S]) if ids[s] is one, else -1.
// Single block exclusive scan (n <= 2^20).  modeling_internvl_chat.py:422-425 / pizero_internvl.py:764-791.
__global__ __launch_bounds__(1024) void img_rank_kernel(const int64_t* __restrict__ ids, int32_t* __restrict__ rank, int n, int64_t img_id,
                                                        int32_t* __restrict__ count_out) {
  __shared__ int wsum[16];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int base = 0; base < n; base += 1024) {
    const int i = base + threadIdx.x;
    const int f = (i < n && ids[i] == img_id) ? 1 : 0;
    int incl = f;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(incl, o, 64);
      if (lane >= o) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int off = carry;
    for (int w = 0; w < wave; ++w) off += wsum[w];
    if (i < n) rank[i] = f ? off + incl - 1 : -1;
    __syncthreads();
    if (threadIdx.x == 0) {
      int tot = 0;
      for (int w = 0; w < 16; ++w) tot += wsum[w];
      carry += tot;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0 && count_out) *count_out = carry;
}
// out[s,:] = rank[s] >= 0 ? vit[rank[s],:] : (zero_pad && ids[s]==pad_id ? 0 : embed[ids[s],:])
__global__ __launch_bounds__(128) void embed_merge_kernel(const int64_t* __restrict__ ids, const int32_t* __restrict__ rank,
                                                          const bf16_t* __restrict__ embed, const bf16_t* __restrict__ vit,
                                                          bf16_t* __restrict__ out, int H, int64_t pad_id, int zero_pad, int n_vit_rows) {
  const int sidx = blockIdx.x;
  const int64_t id = ids[sidx];
  const int r = rank ? rank[sidx] : -1;
  const bf16_t* src = nullptr;
  if (r >= 0) { if (r < n_vit_rows) src = vit + (size_t)r * H; }
  else if (!(zero_pad && id == pad_id)) src = embed + (size_t)id * H;
  for (int c = threadIdx.x * 8; c < H; c += 128 * 8) {
    u32x4 v = {0, 0, 0, 0};
    if (src) v = ld_global_16(src + c);
    st_global_16(out + (size_t)sidx * H + c, v);
  }
}
extern "C" int vlaser_embed_merge(const int64_t* ids, int n, const void* embed, const void* vit, int n_vit_rows, void* out, int H,
                                  long long img_id, long long pad_id, int zero_pad, int32_t* rank_ws, int32_t* count_out, vl_stream_t s) {
  VL_CHECK(ids && embed && out && rank_ws && n > 0 && H % 8 == 0, "vlaser_embed_merge: bad args");
  hipLaunchKernelGGL(img_rank_kernel, dim3(1), dim3(1024), 0, (hipStream_t)s, ids, rank_ws, n, (int64_t)img_id, count_out);
  hipLaunchKernelGGL(embed_merge_kernel, dim3(n), dim3(128), 0, (hipStream_t)s, ids, rank_ws, (const bf16_t*)embed, (const bf16_t*)vit,
                     (bf16_t*)out, H, (int64_t)pad_id, zero_pad, n_vit_rows);
  VL_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------- argmax (greedy)
// out_id[m] = argmax_n logits[m, n] (lowest index wins ties, as torch.argmax); also optional embedding gather of
// the winner into next_h[m,:] so the decode loop never leaves the device.
__global__ __launch_bounds__(1024) void argmax_kernel(const float* __restrict__ logits, int N, int64_t* __restrict__ out_id,
                                                      const bf16_t* __restrict__ embed, bf16_t* __restrict__ next_h, int H) {
  __shared__ float bv[16];
  __shared__ int bi[16];
  __shared__ int winner;
  const int m = blockIdx.x;
  const float* row = logits + (size_t)m * N;
  float best = -INFINITY;
  int idx = 0x7fffffff;
  // r04: 8-byte loads, eight in flight per thread (one float per trip was a chain of 148 load round trips: 54 us of a 1.1 ms decode step)
  auto upd = [&](float v, int n) { if (v > best || (v == best && n < idx)) { best = v; idx = n; } };
  const int N2 = ((reinterpret_cast<uintptr_t>(row) & 7) == 0) ? (N >> 1) : 0;
  const f32x2_t* row2 = reinterpret_cast<const f32x2_t*>(row);
  int n = threadIdx.x;
  for (; n + 7 * 1024 < N2; n += 8 * 1024) {
    f32x2_t v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = row2[n + j * 1024];
#pragma unroll
    for (int j = 0; j < 8; ++j) { upd(v[j][0], 2 * (n + j * 1024)); upd(v[j][1], 2 * (n + j * 1024) + 1); }
  }
  for (; n < N2; n += 1024) { const f32x2_t v = row2[n]; upd(v[0], 2 * n); upd(v[1], 2 * n + 1); }
  for (int k = 2 * N2 + threadIdx.x; k < N; k += 1024) upd(row[k], k);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(idx, o, 64);
    if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
  }
  if ((threadIdx.x & 63) == 0) { bv[threadIdx.x >> 6] = best; bi[threadIdx.x >> 6] = idx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 16; ++w)
      if (bv[w] > best || (bv[w] == best && bi[w] < idx)) { best = bv[w]; idx = bi[w]; }
    out_id[m] = idx;
    winner = idx;
  }
  __syncthreads();
  if (next_h) {
    const bf16_t* src = embed + (size_t)winner * H;
    for (int c = threadIdx.x * 8; c < H; c += 1024 * 8) st_global_16(next_h + (size_t)m * H + c, ld_global_16(src + c));
  }
}
// r05: the row spread over G workgroups.  One workgroup reading a 0.6 MB row is bound by what ONE CU can request (~35 GB/s: 18 us of a 0.96 ms decode step, lesson 46);
// G = 64 workgroups of 256 threads read 9.5 KB each, leave (value, index) pairs, and the LAST one to arrive (device-scope counter, reset by it for the next launch) folds the
// pairs -- max value, lowest index: independent of the arrival order -- and gathers the winner's embedding row.  ws (capacity R rows): [R] counters (zeroed once, re-armed by the last arriver) | [R][G] float | [R][G] int.
constexpr int ARGMAX_G = 64;
__global__ __launch_bounds__(256) void argmax_split_kernel(const float* __restrict__ logits, int N, int64_t* __restrict__ out_id, const bf16_t* __restrict__ embed,
                                                           bf16_t* __restrict__ next_h, int H, float* pv, int* pi, unsigned* ctr) {
  __shared__ float bv[4];
  __shared__ int bi[4];
  __shared__ int winner, is_last;
  const int m = blockIdx.y, g = blockIdx.x, G = gridDim.x;
  const float* row = logits + (size_t)m * N;
  const int per = (((N + G - 1) / G) + 1) & ~1;                  // even: 8-byte loads stay aligned when the row is
  const int lo = min(N, g * per), hi = min(N, lo + per);
  float best = -INFINITY;
  int idx = 0x7fffffff;
  auto upd = [&](float v, int n) { if (v > best || (v == best && n < idx)) { best = v; idx = n; } };
  if ((reinterpret_cast<uintptr_t>(row) & 7) == 0) {
    const f32x2_t* row2 = reinterpret_cast<const f32x2_t*>(row);
    const int lo2 = lo >> 1, hi2 = hi >> 1;                       // pairs [lo2, hi2); an odd last element below
    int n = lo2 + threadIdx.x;
    for (; n + 3 * 256 < hi2; n += 4 * 256) {
      f32x2_t v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = row2[n + j * 256];
#pragma unroll
      for (int j = 0; j < 4; ++j) { upd(v[j][0], 2 * (n + j * 256)); upd(v[j][1], 2 * (n + j * 256) + 1); }
    }
    for (; n < hi2; n += 256) { const f32x2_t v = row2[n]; upd(v[0], 2 * n); upd(v[1], 2 * n + 1); }
    if (threadIdx.x == 0 && (hi & 1)) upd(row[hi - 1], hi - 1);
  } else {
    for (int k = lo + threadIdx.x; k < hi; k += 256) upd(row[k], k);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(idx, o, 64);
    if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
  }
  if ((threadIdx.x & 63) == 0) { bv[threadIdx.x >> 6] = best; bi[threadIdx.x >> 6] = idx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w)
      if (bv[w] > best || (bv[w] == best && bi[w] < idx)) { best = bv[w]; idx = bi[w]; }
    __hip_atomic_store(pv + (size_t)m * G + g, best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(pi + (size_t)m * G + g, idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned old = __hip_atomic_fetch_add(ctr + m, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);      // release: the pair is visible before the count
    is_last = old == (unsigned)(G - 1);
  }
  __syncthreads();
  if (!is_last) return;
  best = -INFINITY; idx = 0x7fffffff;
  if ((int)threadIdx.x < G) {
    best = __hip_atomic_load(pv + (size_t)m * G + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    idx = __hip_atomic_load(pi + (size_t)m * G + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (threadIdx.x < 64) {                                        // G <= 64: one wave holds every pair
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(idx, o, 64);
      if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
    }
    if (threadIdx.x == 0) {
      out_id[m] = idx;
      winner = idx;
      __hip_atomic_store(ctr + m, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // ready for the next launch (stream order)
    }
  }
  __syncthreads();
  if (next_h) {
    const bf16_t* src = embed + (size_t)winner * H;
    for (int c = threadIdx.x * 8; c < H; c += 256 * 8) st_global_16(next_h + (size_t)m * H + c, ld_global_16(src + c));
  }
}
extern "C" int vlaser_argmax_ws_bytes(int M) { return M * (ARGMAX_G * 8 + 4); }
extern "C" int vlaser_argmax(const float* logits, int M, int N, int64_t* out_id, const void* embed, void* next_h, int H, void* ws, int ws_bytes, vl_stream_t s) {
  VL_CHECK(logits && out_id && M > 0 && N > 0, "vlaser_argmax: bad args");
  VL_CHECK(!next_h || (embed && H % 8 == 0), "vlaser_argmax: embed/H");
  if (ws && N >= 4096) {
    VL_CHECK(ws_bytes >= vlaser_argmax_ws_bytes(M) && ((uintptr_t)ws & 3) == 0, "vlaser_argmax: workspace of %d bytes, %d needed (vlaser_argmax_ws_bytes; zeroed once by the caller)", ws_bytes,
             vlaser_argmax_ws_bytes(M));
    // layout by the workspace's CAPACITY, not by this launch's M: the counters must sit where the zeroing (and every earlier launch, whatever its M) left them at 0
    const int cap = ws_bytes / (ARGMAX_G * 8 + 4);
    unsigned* ctr = (unsigned*)ws;
    float* pv = (float*)(ctr + cap);
    int* pi = (int*)(pv + (size_t)cap * ARGMAX_G);
    hipLaunchKernelGGL(argmax_split_kernel, dim3(ARGMAX_G, M), dim3(256), 0, (hipStream_t)s, logits, N, out_id, (const bf16_t*)embed, (bf16_t*)next_h, H, pv, pi, ctr);
  } else {
    hipLaunchKernelGGL(argmax_kernel, dim3(M), dim3(1024), 0, (hipStream_t)s, logits, N, out_id, (const bf16_t*)embed, (bf16_t*)next_h, H);
  }
  VL_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------- pi0 head glue
// vla_prep: x_cat[m, 0:W] = sinusoidal(t) (sin half | cos half, modules.py:9-22), x_cat[m, W:2W] = W1 a[m] + b1
// (modules.py:45-50).  action fp32 [M, adim].  One block per row m.
__global__ __launch_bounds__(256) void vla_prep_kernel(const float* __restrict__ action, const bf16_t* __restrict__ w1,
                                                       const bf16_t* __restrict__ b1, bf16_t* __restrict__ xcat, int Wd, int adim,
                                                       float t, float max_period) {
  const int m = blockIdx.x;
  const int half = Wd / 2;
  const float e = logf(max_period) / (float)(half - 1);
  for (int c = threadIdx.x; c < Wd; c += 256) {
    const int i = c < half ? c : c - half;
    const float ang = t * expf(-e * (float)i);
    xcat[(size_t)m * 2 * Wd + c] = f32_to_bf16(c < half ? sinf(ang) : cosf(ang));
    float acc = bf16_to_f32(b1[c]);
    for (int k = 0; k < adim; ++k) acc += bf16_to_f32(f32_to_bf16(action[m * adim + k])) * bf16_to_f32(w1[c * adim + k]);
    xcat[(size_t)m * 2 * Wd + Wd + c] = f32_to_bf16(acc);
  }
}
extern "C" int vlaser_vla_prep(const float* action, const void* w1, const void* b1, void* xcat, int M, int Wd, int adim, float t,
                               float max_period, vl_stream_t s) {
  VL_CHECK(action && w1 && b1 && xcat && M > 0, "vlaser_vla_prep: bad args");
  hipLaunchKernelGGL(vla_prep_kernel, dim3(M), dim3(256), 0, (hipStream_t)s, action, (const bf16_t*)w1, (const bf16_t*)b1, (bf16_t*)xcat,
                     Wd, adim, t, max_period);
  VL_LAUNCH_CHECK();
  return 0;
}

// proprio encoder: out[m, :] = Wp p[m] + bp  (pizero_internvl.py:823).
__global__ __launch_bounds__(256) void small_linear_kernel(const float* __restrict__ x, const bf16_t* __restrict__ w, const bf16_t* __restrict__ b,
                                                           bf16_t* __restrict__ out, int N, int K) {
  const int m = blockIdx.x;
  for (int n = threadIdx.x; n < N; n += 256) {
    float acc = bf16_to_f32(b[n]);
    for (int k = 0; k < K; ++k) acc += bf16_to_f32(f32_to_bf16(x[m * K + k])) * bf16_to_f32(w[n * K + k]);
    out[(size_t)m * N + n] = f32_to_bf16(acc);
  }
}
extern "C" int vlaser_small_linear(const float* x, const void* w, const void* b, void* out, int M, int N, int K, vl_stream_t s) {
  VL_CHECK(x && w && b && out && M > 0, "vlaser_small_linear: bad args");
  hipLaunchKernelGGL(small_linear_kernel, dim3(M), dim3(256), 0, (hipStream_t)s, x, (const bf16_t*)w, (const bf16_t*)b, (bf16_t*)out, N, K);
  VL_LAUNCH_CHECK();
  return 0;
}

// The update of one integration step (pizero_internvl.py:910-922, `integration_step` :1309-1331).  The reference's `model_step` closure returns the decoder output of THIS
// step's joint pass whatever (x, t) it is handed (:914-917), so heun and rk4 re-combine ONE velocity: method 0 euler a + dt v; 1 heun a + (0.5 dt) (v + v);
// 2 rk4 a + (dt / 6) (((v + 2 v) + 2 v) + v), each product and sum rounded as torch rounds them (no contraction).  `coef` = dt | 0.5 dt | dt / 6, rounded from double by
// the host as torch rounds a Python scalar.  Golden G7c: heun == euler bit for bit, rk4 within 2.4e-7; tests/test_ops_gpu.py::test_vla_glue checks the update against torch bit for bit.
__device__ __forceinline__ float vl_unfused(float x) {       // an opaque copy: keeps hipcc (-ffp-contract=fast) from contracting `a + c * w` into one fma -- torch rounds the product
  asm volatile("" : "+v"(x));
  return x;
}
__device__ __forceinline__ float vl_integrate(float a, float vel, float coef, int method) {
  float w = vel;
  if (method == 2) {
    w = vl_unfused(vel + vl_unfused(2.f * vel));
    w = vl_unfused(w + vl_unfused(2.f * vel));
    w = vl_unfused(w + vel);
  } else if (method == 1) {
    w = vl_unfused(vel + vel);
  }
  // (r06: euler too -- r01-r05 let the compiler fuse `a + dt * vel`; the reference's `action += delta_t * action_vel` rounds twice, and heun must equal euler bit for bit)
  return a + vl_unfused(coef * w);
}

// vla_euler: h = bf16(h_in + sum partials); y = rmsnorm(h) (expert final norm, joint_model.py:804-808);
// vel = Wd y + bd (pizero_internvl.py:911); action += dt * vel (:912); optional clamp on the last step (:927-932).
// One block per row m (<=16 rows); 256 threads.
// One thread = 8 consecutive columns; every load of the row (residual, up to 8 slabs, norm weight, the adim decoder rows) is issued
// up front in straight-line code, then two block reductions (sum of squares; the adim dot products).  The first version walked the
// row 256 columns at a time with scalar loads behind a runtime slab count: 6 dependent round trips + a scalar decoder loop = 11.6 us
// per Euler step for a few KB of work.
template <int ADIM, bool HASP>
__global__ __launch_bounds__(256) void vla_euler_kernel(const bf16_t* __restrict__ h_in, const float* __restrict__ partials, int n_partials,
                                                        int M, const bf16_t* __restrict__ norm_w, float eps, const bf16_t* __restrict__ wd,
                                                        const bf16_t* __restrict__ bd, float* __restrict__ action, int Wd, int adim, float dt,
                                                        float clip, int do_clip, float* __restrict__ vel_out, float* __restrict__ ring,
                                                        const int* __restrict__ ring_ctr, int ring_n, int ring_stride, int method) {
  __shared__ float red[4][ADIM + 1];
  const int m = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool on = (int)threadIdx.x * 8 < Wd;
  const int c = on ? threadIdx.x * 8 : 0;                  // idle threads (Wd < 2048) shadow column 0 and contribute zeros
  const size_t off = (size_t)m * Wd + c, slab = (size_t)M * Wd;
  const u32x4 hv = ld_global_16(h_in + off);
  const u32x4 nw = ld_global_16(norm_w + c);
  f32x4 q[16];
  if constexpr (HASP) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float* pp = partials + off + (size_t)min(u, n_partials - 1) * slab;       // clamped, unconditional (n_partials >= 1 here)
      q[2 * u] = *reinterpret_cast<const f32x4*>(pp);
      q[2 * u + 1] = *reinterpret_cast<const f32x4*>(pp + 4);
    }
  }
  u32x4 wv[ADIM];
#pragma unroll
  for (int j = 0; j < ADIM; ++j) wv[j] = ld_global_16(wd + (size_t)min(j, adim - 1) * Wd + c);
  float v[8];
#pragma unroll
  for (int j = 0; j < 4; ++j) { v[2 * j] = bf16lo_to_f32(hv[j]); v[2 * j + 1] = bf16hi_to_f32(hv[j]); }
  float sl[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if constexpr (HASP) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const bool use = u < n_partials;                     // select, not multiply: clamped duplicates must not count
#pragma unroll
      for (int j = 0; j < 4; ++j) { sl[j] += use ? q[2 * u][j] : 0.f; sl[4 + j] += use ? q[2 * u + 1][j] : 0.f; }
    }
  }
  float ssq = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) { v[j] = on ? round_bf16(v[j] + sl[j]) : 0.f; ssq += v[j] * v[j]; }
  ssq = wave_sum(ssq);
  if (lane == 0) red[wave][ADIM] = ssq;
  __syncthreads();
  const float rs = rsqrtf((red[0][ADIM] + red[1][ADIM] + red[2][ADIM] + red[3][ADIM]) / (float)Wd + eps);
  float y[8];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    y[2 * j] = round_bf16(round_bf16(v[2 * j] * rs) * bf16lo_to_f32(nw[j]));
    y[2 * j + 1] = round_bf16(round_bf16(v[2 * j + 1] * rs) * bf16hi_to_f32(nw[j]));
  }
#pragma unroll
  for (int j = 0; j < ADIM; ++j) {
    float acc = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) acc += y[2 * e] * bf16lo_to_f32(wv[j][e]) + y[2 * e + 1] * bf16hi_to_f32(wv[j][e]);
    acc = wave_sum(acc);
    if (lane == 0) red[wave][j] = acc;
  }
  __syncthreads();
  if ((int)threadIdx.x < adim) {
    const int j = threadIdx.x;
    const float vel = round_bf16(red[0][j] + red[1][j] + red[2][j] + red[3][j] + bf16_to_f32(bd[j]));
    float av = vl_integrate(action[m * adim + j], vel, dt, method);
    if (do_clip) av = fminf(fmaxf(av, -clip), clip);
    action[m * adim + j] = av;
    if (vel_out) vel_out[m * adim + j] = vel;
    // (ABI 5) the caller's copy of the result: slot (call counter mod ring_n) of a small ring, so infer_action returns a view instead of launching a clone.
    // (ABI 6) ring_ctr = {call number k, error word of even calls, error word of odd calls} (vlaser_vla_stage): a dense mask the kernels' descriptors cannot
    // express turns the caller's copy into NaN -- an unsupported mask is never served silently (the host raises at its next poll of the word)
    if (ring) {
      const int k = ring_ctr[0];
      if (ring_ctr[1 + (k & 1)] != 0) av = __builtin_nanf("");
      ring[(size_t)((unsigned)k % (unsigned)ring_n) * ring_stride + m * adim + j] = av;
    }
  }
}
extern "C" int vlaser_vla_euler(const void* h_in, const float* partials, int n_partials, int M, const void* norm_w, float eps,
                                const void* wd, const void* bd, float* action, int Wd, int adim, float dt, float clip, int do_clip,
                                float* vel_out, float* ring, const int* ring_ctr, int ring_n, int ring_stride, int method, vl_stream_t s) {
  VL_CHECK(h_in && norm_w && wd && bd && action && M > 0 && Wd <= 2048 && Wd % 8 == 0, "vlaser_vla_euler: bad args");
  VL_CHECK(method >= 0 && method <= 2, "vlaser_vla_euler: method 0 (euler) | 1 (heun) | 2 (rk4), got %d", method);
  VL_CHECK(!ring || (ring_ctr && ring_n >= 1 && ring_stride >= M * adim), "vlaser_vla_euler: the output ring needs its counter, >= 1 slots of >= M * adim floats");
  VL_CHECK(n_partials >= 0 && n_partials <= 8 && (n_partials == 0 || partials) && adim >= 1 && adim <= 16,
           "vlaser_vla_euler: 0..8 partial slabs, action_dim <= 16 (got %d, %d)", n_partials, adim);
#define VL_EULER(AD_, HP_)                                                                                                          \
  hipLaunchKernelGGL((vla_euler_kernel<AD_, HP_>), dim3(M), dim3(256), 0, (hipStream_t)s, (const bf16_t*)h_in, partials, n_partials, M, \
                     (const bf16_t*)norm_w, eps, (const bf16_t*)wd, (const bf16_t*)bd, action, Wd, adim, dt, clip, do_clip, vel_out, ring, ring_ctr, ring_n, ring_stride, method)
  if (adim <= 8) { if (n_partials > 0) VL_EULER(8, true); else VL_EULER(8, false); }
  else { if (n_partials > 0) VL_EULER(16, true); else VL_EULER(16, false); }
#undef VL_EULER
  VL_LAUNCH_CHECK();
  return 0;
}

// vla_step (r03): everything BETWEEN two passes through the expert's layers in ONE launch -- the tail of Euler step s-1 (vla_euler above: last split-K
// reduce + final norm + action decoder + x += dt v) and the action encoder of step s (pizero_internvl.py:863-900: linear_1, sinusoidal time embedding,
// cat, linear_2, swish, linear_3), which were 4 launches of ~5 us each for a few KB of arithmetic.  The two Linear layers before the swish have no
// nonlinearity between them and the time embedding does not depend on the action, so the host folds them once per checkpoint:
//     pre[m, n] = sum_k W21[n, k] a[m, k] + C[s, n],   W21 = W2[:, Wd:] W1  (Wd x adim),   C[s] = W2[:, :Wd] temb(t_s) + W2[:, Wd:] b1 + b2
// (fp32; the reference rounds linear_1's output to bf16 before linear_2 -- a 2^-9 relative perturbation of each of the Wd addends, far below the
// bf16 rounding of `pre` itself).  Every workgroup recomputes the tail and e2 = swish(pre) for all rows (a few thousand MACs) and owns Wd / gridDim.x
// output columns of linear_3, whose weights it requests first.  Actions ping-pong between two buffers (all workgroups read, workgroup 0 writes).
template <int ADIM, int KJ>       // KJ = Wd / 256: 4-column chunks per lane of a row (columns lane*4 + 256 j)
__global__ __launch_bounds__(256) void vla_step_kernel(const bf16_t* __restrict__ h_in, const float* __restrict__ partials, int n_partials, int rows_in, int row_off,
                                                       const bf16_t* __restrict__ norm_w, float eps, const bf16_t* __restrict__ wd, const bf16_t* __restrict__ bd,
                                                       const float* __restrict__ a_in, float* __restrict__ a_out, float* __restrict__ vel_out, float dt, int finish,
                                                       const float* __restrict__ w21, const float* __restrict__ cs, const bf16_t* __restrict__ w3,
                                                       const bf16_t* __restrict__ b3, bf16_t* __restrict__ h_out, int M, int Wd, int adim, int method) {
  constexpr int MAXM = 16, MAXW = 1024;
  __shared__ float a_s[MAXM * ADIM];
  __shared__ __attribute__((aligned(16))) bf16_t e2_s[MAXM * MAXW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = blockIdx.x * 16;                                                // this workgroup's 16 linear_3 outputs; wave w owns columns n0 + w + 4 i
  // ---- linear_3 weights of this wave's first column: requested before anything else (the only HBM-cold data of the launch)
  constexpr int CPW = 4;                                                         // linear_3 columns per wave (16 per workgroup): all requested up front
  u32x2 w3v[CPW][KJ];
  float b3v[CPW];
#pragma unroll
  for (int i = 0; i < CPW; ++i) {
    const int n = min(n0 + wave + 4 * i, Wd - 1);
#pragma unroll
    for (int j = 0; j < KJ; ++j) w3v[i][j] = *reinterpret_cast<const u32x2*>(w3 + (size_t)n * Wd + lane * 4 + 256 * j);
    b3v[i] = bf16_to_f32(b3[n]);
  }
  // the folded encoder constants of this thread's KJ columns (n = tid + 256 j): also up front
  float w21v[KJ][ADIM], csv[KJ];
#pragma unroll
  for (int j = 0; j < KJ; ++j) {
    csv[j] = cs[tid + 256 * j];
#pragma unroll
    for (int k = 0; k < ADIM; ++k) w21v[j][k] = w21[(tid + 256 * j) * adim + min(k, adim - 1)];
  }
  // ---- tail of the previous Euler step: wave w finishes rows w, w + 4, ... (wave-level reductions only)
  if (finish) {
    for (int m = wave; m < M; m += 4) {
      const size_t off = (size_t)(m + row_off) * Wd, slab = (size_t)rows_in * Wd;
      // every load of the row is issued up front in straight-line code (clamped slab index, select on use): a runtime slab loop is one round trip per slab
      u32x2 hv[KJ];
      f32x4 q[KJ][8];
#pragma unroll
      for (int j = 0; j < KJ; ++j) hv[j] = *reinterpret_cast<const u32x2*>(h_in + off + lane * 4 + 256 * j);
      if (n_partials > 0) {
#pragma unroll
        for (int j = 0; j < KJ; ++j)
#pragma unroll
          for (int u = 0; u < 8; ++u) q[j][u] = *reinterpret_cast<const f32x4*>(partials + (size_t)min(u, n_partials - 1) * slab + off + lane * 4 + 256 * j);
      }
      float v[KJ][4];
      float ssq = 0.f;
#pragma unroll
      for (int j = 0; j < KJ; ++j) {
        f32x4 sl = {0, 0, 0, 0};
        if (n_partials > 0) {
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const bool use = u < n_partials;
            sl[0] += use ? q[j][u][0] : 0.f; sl[1] += use ? q[j][u][1] : 0.f; sl[2] += use ? q[j][u][2] : 0.f; sl[3] += use ? q[j][u][3] : 0.f;
          }
        }
        const float hv4[4] = {bf16lo_to_f32(hv[j][0]), bf16hi_to_f32(hv[j][0]), bf16lo_to_f32(hv[j][1]), bf16hi_to_f32(hv[j][1])};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[j][e] = round_bf16(hv4[e] + sl[e]);
          ssq += v[j][e] * v[j][e];
        }
      }
      ssq = wave_sum(ssq);
      const float rs = rsqrtf(ssq / (float)Wd + eps);
      float acc[ADIM];
#pragma unroll
      for (int k = 0; k < ADIM; ++k) acc[k] = 0.f;
#pragma unroll
      for (int j = 0; j < KJ; ++j) {
        const int c = lane * 4 + 256 * j;
        const u32x2 nw = *reinterpret_cast<const u32x2*>(norm_w + c);
        const float nw4[4] = {bf16lo_to_f32(nw[0]), bf16hi_to_f32(nw[0]), bf16lo_to_f32(nw[1]), bf16hi_to_f32(nw[1])};
        float y[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = round_bf16(round_bf16(v[j][e] * rs) * nw4[e]);
#pragma unroll
        for (int k = 0; k < ADIM; ++k) {
          const u32x2 wv = *reinterpret_cast<const u32x2*>(wd + (size_t)min(k, adim - 1) * Wd + c);
          acc[k] += y[0] * bf16lo_to_f32(wv[0]) + y[1] * bf16hi_to_f32(wv[0]) + y[2] * bf16lo_to_f32(wv[1]) + y[3] * bf16hi_to_f32(wv[1]);
        }
      }
#pragma unroll
      for (int k = 0; k < ADIM; ++k) acc[k] = wave_sum(acc[k]);
      if (lane < adim) {
        float vel = 0.f;
#pragma unroll
        for (int k = 0; k < ADIM; ++k) vel = lane == k ? acc[k] : vel;
        vel = round_bf16(vel + bf16_to_f32(bd[lane]));
        const float av = vl_integrate(a_in[m * adim + lane], vel, dt, method);
        a_s[m * ADIM + lane] = av;
        if (blockIdx.x == 0) {
          a_out[m * adim + lane] = av;
          if (vel_out) vel_out[m * adim + lane] = vel;
        }
      }
    }
  } else {
    for (int i = tid; i < M * adim; i += 256) {
      const float av = a_in[i];
      a_s[(i / adim) * ADIM + i % adim] = av;
      if (blockIdx.x == 0 && a_out != a_in) a_out[i] = av;
    }
  }
  __syncthreads();
  // ---- e2 = swish(bf16(W21 bf16(a) + C[s]))  for all rows (the encoder reads the action through a bf16 cast, as nn.Linear on a bf16 module does)
  for (int m = 0; m < M; ++m) {
    float av[ADIM];
#pragma unroll
    for (int k = 0; k < ADIM; ++k) av[k] = k < adim ? round_bf16(a_s[m * ADIM + k]) : 0.f;
#pragma unroll
    for (int j = 0; j < KJ; ++j) {
      float pre = csv[j];
#pragma unroll
      for (int k = 0; k < ADIM; ++k) pre += w21v[j][k] * av[k];
      e2_s[m * Wd + tid + 256 * j] = f32_to_bf16(silu(round_bf16(pre)));
    }
  }
  __syncthreads();
  // ---- linear_3: columns n0 + wave + 4 i of this workgroup; lane owns k = lane*4 + 256 j.  Four rows at a time, all 16 (row, column) dot products
  // reduced TOGETHER: a wave reduction is 6 dependent cross-lane steps of ~100 cycles, and 16 of them one after the other were 4 us of this launch
  for (int mb = 0; mb < M; mb += 4) {
    float acc[4][CPW];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = min(mb + r, M - 1);
      u32x2 ev[KJ];
#pragma unroll
      for (int j = 0; j < KJ; ++j) ev[j] = *reinterpret_cast<const u32x2*>(e2_s + m * Wd + lane * 4 + 256 * j);
#pragma unroll
      for (int i = 0; i < CPW; ++i) {
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < KJ; ++j)
          a += bf16lo_to_f32(ev[j][0]) * bf16lo_to_f32(w3v[i][j][0]) + bf16hi_to_f32(ev[j][0]) * bf16hi_to_f32(w3v[i][j][0]) +
               bf16lo_to_f32(ev[j][1]) * bf16lo_to_f32(w3v[i][j][1]) + bf16hi_to_f32(ev[j][1]) * bf16hi_to_f32(w3v[i][j][1]);
        acc[r][i] = a;
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < CPW; ++i) acc[r][i] = wave_sum(acc[r][i]);
    if (lane < 4 * CPW) {
      const int r = lane >> 2, i = lane & 3;
      float v = 0.f;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
#pragma unroll
        for (int ii = 0; ii < CPW; ++ii) v = (r == rr && i == ii) ? acc[rr][ii] + b3v[ii] : v;
      if (mb + r < M) h_out[(size_t)(mb + r) * Wd + n0 + wave + 4 * i] = f32_to_bf16(v);
    }
  }
}
extern "C" int vlaser_vla_step(const void* h_in, const float* partials, int n_partials, int rows_in, int row_off, const void* norm_w, float eps, const void* wd,
                               const void* bd, const float* a_in, float* a_out, float* vel_out, float dt, int finish, const float* w21, const float* cs,
                               const void* w3, const void* b3, void* h_out, int M, int Wd, int adim, int method, vl_stream_t s) {
  VL_CHECK(a_in && a_out && w21 && cs && w3 && b3 && h_out && M >= 1 && M <= 16, "vlaser_vla_step: bad args (1..16 rows)");
  VL_CHECK(method >= 0 && method <= 2, "vlaser_vla_step: method 0 (euler) | 1 (heun) | 2 (rk4), got %d", method);
  VL_CHECK(Wd % 256 == 0 && Wd <= 1024 && adim >= 1 && adim <= 16, "vlaser_vla_step: Wd must be a multiple of 256 and <= 1024, action_dim <= 16 (wider experts keep the separate launches)");
  VL_CHECK(adim <= 8 || Wd <= 768, "vlaser_vla_step: action_dim > 8 is built for Wd <= 768 (the 16 x 1024 variant needs scratch: such heads keep the separate launches)");
  VL_CHECK(!finish || (h_in && norm_w && wd && bd && n_partials >= 0 && n_partials <= 8 && (n_partials == 0 || partials) && rows_in >= M + row_off && row_off >= 0 && a_in != a_out),
           "vlaser_vla_step: finish needs h_in / norm / decoder, <= 8 slabs of rows_in >= M + row_off rows, and distinct action buffers");
  const int blocks = Wd / 16;                    // 16 linear_3 outputs per workgroup (48 workgroups at Wd = 768)
#define VL_STEP(AD_, KJ_)                                                                                                                          \
  hipLaunchKernelGGL((vla_step_kernel<AD_, KJ_>), dim3(blocks), dim3(256), 0, (hipStream_t)s, (const bf16_t*)h_in, partials, n_partials, rows_in, row_off, \
                     (const bf16_t*)norm_w, eps, (const bf16_t*)wd, (const bf16_t*)bd, a_in, a_out, vel_out, dt, finish, w21, cs, (const bf16_t*)w3,   \
                     (const bf16_t*)b3, (bf16_t*)h_out, M, Wd, adim, method)
#define VL_STEP_KJ(AD_) { switch (Wd / 256) { case 1: VL_STEP(AD_, 1); break; case 2: VL_STEP(AD_, 2); break; case 3: VL_STEP(AD_, 3); break; default: if constexpr (AD_ <= 8) VL_STEP(AD_, 4); } }
  if (adim <= 8) VL_STEP_KJ(8) else VL_STEP_KJ(16)
#undef VL_STEP_KJ
#undef VL_STEP
  VL_LAUNCH_CHECK();
  return 0;
}

// fp32 -> bf16 cast (pixel values / host inputs)
__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256;
  for (; i < n; i += stride) y[i] = f32_to_bf16(x[i]);
}
extern "C" int vlaser_cast_f32_bf16(const float* x, void* y, long long n, vl_stream_t s) {
  VL_CHECK(x && y && n > 0, "vlaser_cast_f32_bf16: bad args");
  const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, x, (bf16_t*)y, (size_t)n);
  VL_LAUNCH_CHECK();
  return 0;
}

// uint8 image -> normalised bf16 pixel_values [N,3,H,W] on the device: the host hands over 1 byte per sample instead of a
// 4-byte fp32 tensor that then goes through cast_f32_bf16.  Arithmetic = the reference's, in its order, in fp32 (IEEE division), rounded once:
//   mode 0  InternVLAProcessor (processing.py:51-63,303-311): (u8 * (1/255) - mean) / std
//   mode 1  torchvision ToTensor + Normalize (dataset.py:293-300): (u8 / 255 - mean) / std
// layout 0: planar [N,3,H,W] (the VLA processor's input); 1: interleaved [N,H,W,3] (PIL / numpy tiles).  Four pixels per thread.
__global__ __launch_bounds__(256) void normalize_u8_kernel(const uint8_t* __restrict__ in, bf16_t* __restrict__ out, int hw, int layout, int mode,
                                                           float m0, float m1, float m2, float s0, float s1, float s2, size_t total4) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256;
  const int hw4 = hw >> 2;
  const float r255 = (float)(1.0 / 255.0);
  for (; i < total4; i += stride) {
    const size_t plane = i / hw4;                  // (n, c)
    const int p4 = (int)(i - plane * hw4);         // group of 4 pixels inside the plane
    const int c = (int)(plane % 3);
    const size_t n = plane / 3;
    const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
    uint8_t px[4];
    if (layout == 0) {
      const uint32_t w = *reinterpret_cast<const uint32_t*>(in + plane * hw + (size_t)p4 * 4);
      px[0] = w & 255; px[1] = (w >> 8) & 255; px[2] = (w >> 16) & 255; px[3] = w >> 24;
    } else {
      const uint8_t* src = in + (n * hw + (size_t)p4 * 4) * 3 + c;
      px[0] = src[0]; px[1] = src[3]; px[2] = src[6]; px[3] = src[9];
    }
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float x = mode == 0 ? (float)px[j] * r255 : (float)px[j] / 255.0f;
      v[j] = (x - mean) / sd;
    }
    *reinterpret_cast<u32x2*>(out + plane * hw + (size_t)p4 * 4) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
  }
}
extern "C" int vlaser_normalize_u8(const void* in_u8, void* out_bf16, int n_img, int hw, int layout, int mode, const float* mean3, const float* std3,
                                   vl_stream_t s) {
  VL_CHECK(in_u8 && out_bf16 && mean3 && std3 && n_img > 0 && hw > 0 && hw % 4 == 0, "vlaser_normalize_u8: bad args (H*W must be a multiple of 4)");
  VL_CHECK((layout == 0 || layout == 1) && (mode == 0 || mode == 1), "vlaser_normalize_u8: layout / mode must be 0 or 1");
  VL_CHECK(((uintptr_t)in_u8 & 3) == 0 && ((uintptr_t)out_bf16 & 7) == 0, "vlaser_normalize_u8: alignment");
  const size_t total4 = (size_t)n_img * 3 * (hw / 4);
  const int blocks = (int)((total4 + 255) / 256 < 2048 ? (total4 + 255) / 256 : 2048);
  hipLaunchKernelGGL(normalize_u8_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, (const uint8_t*)in_u8, (bf16_t*)out_bf16, hw, layout, mode,
                     mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], total4);
  VL_LAUNCH_CHECK();
  return 0;
}

// vla_stage (ABI 5): EVERY per-call input of infer_action into the static slots of the captured chunk graph in ONE launch.  r03 staged them with six
// torch copy_ / cast launches + a clone of the result: ~90 us of mostly idle GPU between two chunks (tools/chunk_timeline.py: the copies start 13 us apart).
//   block 0           : input_ids (int64 [B, T]) -> ids slot; valid_len[b] = given (int32 / int64), or the zero count of the dense mask's proprio row, or the
//                       number of non-pad ids of row b; proprio and noise (fp32) -> their slots; position ids (int64 -> int32 slots); call counter += 1 (slot of
//                       the output ring, read by the chunk's last kernel); the NEXT call's error word cleared
//   blocks 1 .. nm    : (ABI 6) the reference's dense masks (pizero_internvl.py:517-603: image_text_proprio_mask [B,1,T+1,T+1], action_mask [B,1,na,T+1+na], additive
//                       0 / dtype-min) checked against the ONLY pattern the (valid_len, blk_start) descriptors of the kernels express; a mismatch sets bits of this
//                       call's error word (the chunk's last kernel then returns NaN, the host raises at its next poll) -- r04 copied the mask to the host per call
//                       (a blocking 593 KB D2H) and never read action_mask
//   the other blocks  : pixel_values -> bf16 slot: bf16 copy / fp32 cast / uint8 planar normalise ((u8 * (1/255) - mean) / std, vlaser_normalize_u8 mode 0)
struct VlaStageP {
  const int64_t* ids; int64_t* ids_out; int B, T; long long pad_id;
  const void* valid_in; int valid_is_i64; int32_t* valid_out;
  const float* proprio; float* proprio_out; int n_proprio;
  const float* noise; float* noise_out; int n_noise;
  const void* pix; bf16_t* pix_out; long long n_pix8; int pix_dtype; int hw;      // n_pix8: groups of 8 elements
  float m0, m1, m2, s0, s1, s2;
  int* ctr; int call_no;
  const void* itp_mask; const void* action_mask; int mask_dtype, n_act, n_mask_blocks;
  long long itp_bs, itp_rs, act_bs, act_rs;       // element strides of the masks' batch / row axes (the reference hands out SLICES of the full mask, :589-603)
  const int64_t* pos_vlm; const int64_t* pos_pro; const int64_t* pos_act;
  int32_t* pos_vlm_out; int32_t* pos_pro_out; int32_t* pos_act_out; int32_t* pos_ride_out;
  float* mask_slot; int mask_ld;                  // (ABI 8) general masks: copy instead of check
};
#define VLS_ROWS_PER_BLOCK 8
// element e of an additive mask "lets the key through" iff it is +-0 (the reference writes exactly 0 or finfo(dtype).min)
__device__ __forceinline__ bool vls_mask_open(const void* m, int dt, size_t e) {
  if (dt == 1) return (reinterpret_cast<const uint32_t*>(m)[e] & 0x7fffffffu) == 0;
  return (reinterpret_cast<const uint16_t*>(m)[e] & 0x7fffu) == 0;      // bf16 / fp16
}
__device__ __forceinline__ float vls_mask_f32(const void* m, int dt, size_t e) {
  if (dt == 1) return reinterpret_cast<const float*>(m)[e];
  if (dt == 0) return bf16_to_f32(reinterpret_cast<const bf16_t*>(m)[e]);
  return (float)reinterpret_cast<const _Float16*>(m)[e];
}
__global__ __launch_bounds__(256) void vla_stage_kernel(VlaStageP p) {
  const int tid = threadIdx.x;
  const int T = p.T, W1 = T + 1, na = p.n_act, W2 = T + 1 + na;
  if (blockIdx.x == 0) {
    __shared__ int cnt[4];
    for (int i = tid; i < p.B * p.T; i += 256) p.ids_out[i] = p.ids[i];
    for (int i = tid; i < p.n_proprio; i += 256) p.proprio_out[i] = p.proprio[i];
    for (int i = tid; i < p.n_noise; i += 256) p.noise_out[i] = p.noise[i];
    if (p.pos_vlm) for (int i = tid; i < p.B * T; i += 256) p.pos_vlm_out[i] = (int32_t)p.pos_vlm[i];
    if (p.pos_pro) for (int i = tid; i < p.B; i += 256) p.pos_pro_out[i] = (int32_t)p.pos_pro[i];
    if (p.pos_act) for (int i = tid; i < p.B * na; i += 256) p.pos_act_out[i] = (int32_t)p.pos_act[i];
    if (p.pos_ride_out && p.pos_pro && p.pos_act && tid <= na) p.pos_ride_out[tid] = tid == 0 ? (int32_t)p.pos_pro[0] : (int32_t)p.pos_act[tid - 1];
    if (p.valid_in) {
      if (tid < p.B) p.valid_out[tid] = p.valid_is_i64 ? (int32_t)reinterpret_cast<const int64_t*>(p.valid_in)[tid] : reinterpret_cast<const int32_t*>(p.valid_in)[tid];
    } else {
      for (int b = 0; b < p.B; ++b) {
        // dense mask given: the zero count of the proprio row over the image / text columns (what r04's host-side mask_to_descriptor did); else
        // (input_ids != pad_token_id).sum(-1): what the reference's attention_mask counts for right-padded prompts
        int c = 0;
        if (p.itp_mask) { for (int i = tid; i < T; i += 256) c += vls_mask_open(p.itp_mask, p.mask_dtype, (size_t)(b * p.itp_bs + T * p.itp_rs + i)); }
        else { for (int i = tid; i < T; i += 256) c += p.ids[(size_t)b * T + i] != p.pad_id; }
        c = (int)wave_sum((float)c);
        if ((tid & 63) == 0) cnt[tid >> 6] = c;
        __syncthreads();
        if (tid == 0) p.valid_out[b] = cnt[0] + cnt[1] + cnt[2] + cnt[3];
        __syncthreads();
      }
    }
    if (tid == 0 && p.ctr) {
      // the call number comes from the HOST (it advances only after this launch was accepted): the device copy cannot drift from the host's ring index.
      // ctr[1 + (k & 1)] = this call's error word (cleared by the previous call, OR-ed by the mask blocks below, read by the chunk's last kernel);
      // clear the NEXT call's word: nobody else touches it during this call
      p.ctr[0] = p.call_no;
      p.ctr[1 + ((p.call_no + 1) & 1)] = 0;
    }
    return;
  }
  if ((int)blockIdx.x <= p.n_mask_blocks) {
    // rows [r0, r0 + 8) of batch element b over the concatenation {T+1 rows of image_text_proprio_mask, na rows of action_mask}; one wave per 2 rows
    const int rpb = W1 + na, bpb = (rpb + VLS_ROWS_PER_BLOCK - 1) / VLS_ROWS_PER_BLOCK;
    const int mb = blockIdx.x - 1, b = mb / bpb, r0 = (mb - b * bpb) * VLS_ROWS_PER_BLOCK;
    const int lane = tid & 63, wave = tid >> 6;
    int c;          // valid prefix length of this batch element: given, or the proprio row's zero count (every wave counts it itself: 385 elements)
    if (p.valid_in) c = p.valid_is_i64 ? (int)reinterpret_cast<const int64_t*>(p.valid_in)[b] : reinterpret_cast<const int32_t*>(p.valid_in)[b];
    else if (p.itp_mask) {
      int n = 0;
      for (int i = lane; i < T; i += 64) n += vls_mask_open(p.itp_mask, p.mask_dtype, (size_t)(b * p.itp_bs + T * p.itp_rs + i));
      c = (int)wave_sum((float)n);
    } else {
      int n = 0;
      for (int i = lane; i < T; i += 64) n += p.ids[(size_t)b * T + i] != p.pad_id;
      c = (int)wave_sum((float)n);
    }
    int bad = 0;
    if (p.mask_slot) {
      // general masks: rows as fp32 into the slot the VL_ATTN_DENSE launches read; the only pattern still refused is a prefix row that sees the proprio key
      for (int rr = wave; rr < VLS_ROWS_PER_BLOCK; rr += 4) {
        const int r = r0 + rr;
        if (r >= rpb) break;
        float* dst = p.mask_slot + ((size_t)b * rpb + r) * p.mask_ld;
        const bool itp = r < W1;
        const int wd = itp ? W1 : W2;
        const size_t base = itp ? (size_t)(b * p.itp_bs + r * p.itp_rs) : (size_t)(b * p.act_bs + (r - W1) * p.act_rs);
        for (int j = lane; j < p.mask_ld; j += 64) dst[j] = j < wd ? vls_mask_f32(itp ? p.itp_mask : p.action_mask, p.mask_dtype, base + j) : -3.402823466e38f;
        if (r < T && lane == 0 && vls_mask_f32(p.itp_mask, p.mask_dtype, base + T) > -1.0e30f) bad |= 8;
      }
      if (bad && p.ctr) atomicOr(&p.ctr[1 + (p.call_no & 1)], bad);
      return;
    }
    for (int rr = wave; rr < VLS_ROWS_PER_BLOCK; rr += 4) {
      const int r = r0 + rr;
      if (r >= rpb) break;
      if (r < W1) {
        if (!p.itp_mask || (r >= c && r < T)) continue;           // rows of padded image / text positions: "don't care" (nobody attends to them)
        for (int j = lane; j < W1; j += 64) {
          const bool want = (j < c) || (r == T && j == T);        // valid prefix (+ the proprio row sees itself); prefix rows never see the proprio key
          if (vls_mask_open(p.itp_mask, p.mask_dtype, (size_t)(b * p.itp_bs + r * p.itp_rs + j)) != want) bad |= (r == T && j < T) ? 1 : 2;    // 1: the valid prefix is not contiguous
        }
      } else {
        if (!p.action_mask) continue;
        const int a = r - W1;
        for (int j = lane; j < W2; j += 64) {
          const bool want = (j < c) || (j >= T);                   // valid prefix + proprio + every action token
          if (vls_mask_open(p.action_mask, p.mask_dtype, (size_t)(b * p.act_bs + a * p.act_rs + j)) != want) bad |= 4;
        }
      }
    }
    if (bad && p.ctr) atomicOr(&p.ctr[1 + (p.call_no & 1)], bad);
    return;
  }
  const int pb0 = 1 + p.n_mask_blocks;
  const long long stride = (long long)(gridDim.x - pb0) * 256;
  const float r255 = (float)(1.0 / 255.0);
  for (long long i = (long long)(blockIdx.x - pb0) * 256 + tid; i < p.n_pix8; i += stride) {
    u32x4 o;
    if (p.pix_dtype == 0) {
      o = reinterpret_cast<const u32x4*>(p.pix)[i];
    } else if (p.pix_dtype == 1) {
      const f32x4 a = reinterpret_cast<const f32x4*>(p.pix)[2 * i], b = reinterpret_cast<const f32x4*>(p.pix)[2 * i + 1];
      o = u32x4{pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]), pack_bf16x2(b[0], b[1]), pack_bf16x2(b[2], b[3])};
    } else {
      const long long plane = (i * 8) / p.hw;                // (n, c): hw is a multiple of 8
      const int c = (int)(plane % 3);
      const float mean = c == 0 ? p.m0 : (c == 1 ? p.m1 : p.m2), sd = c == 0 ? p.s0 : (c == 1 ? p.s1 : p.s2);
      const u32x2 w = reinterpret_cast<const u32x2*>(p.pix)[i];
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = ((float)((w[j >> 2] >> (8 * (j & 3))) & 255u) * r255 - mean) / sd;
      o = u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
    }
    reinterpret_cast<u32x4*>(p.pix_out)[i] = o;
  }
}
extern "C" int vlaser_vla_stage(const VlaserVlaStageArgs* a, vl_stream_t s) {
  VL_CHECK(a && a->ids && a->ids_out && a->valid_out && a->B >= 1 && a->B <= 256 && a->T >= 1, "vlaser_vla_stage: ids / valid_len slots missing or batch outside 1..256");
  VL_CHECK((a->n_proprio == 0 || (a->proprio && a->proprio_out)) && (a->n_noise == 0 || (a->noise && a->noise_out)), "vlaser_vla_stage: proprio / noise pointers");
  VL_CHECK(a->n_pix == 0 || (a->pix && a->pix_out && a->n_pix % 8 == 0 && a->pix_dtype >= 0 && a->pix_dtype <= 2), "vlaser_vla_stage: pixel count must be a multiple of 8, dtype 0 bf16 / 1 f32 / 2 u8");
  VL_CHECK(a->n_pix == 0 || a->pix_dtype != 2 || (a->hw > 0 && a->hw % 8 == 0), "vlaser_vla_stage: uint8 pixels need H*W (a multiple of 8)");
  VL_CHECK(a->n_pix == 0 || ((((uintptr_t)a->pix) & (a->pix_dtype == 2 ? 7 : 15)) == 0 && (((uintptr_t)a->pix_out) & 15) == 0), "vlaser_vla_stage: pixel buffers must be 16-byte (uint8: 8-byte) aligned");
  const bool masks = a->itp_mask || a->action_mask;
  VL_CHECK(!masks || (a->call_ctr && a->mask_dtype >= 0 && a->mask_dtype <= 2 && a->n_act >= 1 && a->n_act <= 64),
           "vlaser_vla_stage: dense masks need the call-counter / error words, mask_dtype 0 bf16 / 1 f32 / 2 f16 and 1..64 action tokens");
  VL_CHECK((!a->pos_vlm || a->pos_vlm_out) && (!a->pos_pro || a->pos_pro_out) && (!a->pos_act || (a->pos_act_out && a->n_act >= 1)), "vlaser_vla_stage: position-id slots missing");
  VL_CHECK(!a->pos_ride_out || (a->B == 1 && a->n_act >= 1 && a->n_act < 256), "vlaser_vla_stage: pos_ride_out is the batch-1 [proprio | action] position row");
  VL_CHECK(!a->mask_slot || (a->itp_mask && a->action_mask && a->mask_ld % 32 == 0 && a->mask_ld >= a->T + 1 + a->n_act && (((uintptr_t)a->mask_slot) & 15) == 0),
           "vlaser_vla_stage: general masks need BOTH dense masks and a 16-byte aligned slot with mask_ld a multiple of 32 >= T + 1 + n_act");
  VlaStageP p;
  p.ids = a->ids; p.ids_out = a->ids_out; p.B = a->B; p.T = a->T; p.pad_id = a->pad_id;
  p.valid_in = a->valid_in; p.valid_is_i64 = a->valid_is_i64; p.valid_out = a->valid_out;
  p.proprio = a->proprio; p.proprio_out = a->proprio_out; p.n_proprio = a->n_proprio;
  p.noise = a->noise; p.noise_out = a->noise_out; p.n_noise = a->n_noise;
  p.pix = a->pix; p.pix_out = (bf16_t*)a->pix_out; p.n_pix8 = a->n_pix / 8; p.pix_dtype = a->pix_dtype; p.hw = a->hw;
  p.m0 = a->mean[0]; p.m1 = a->mean[1]; p.m2 = a->mean[2]; p.s0 = a->std[0]; p.s1 = a->std[1]; p.s2 = a->std[2];
  p.ctr = a->call_ctr; p.call_no = a->call_no;
  p.itp_mask = a->itp_mask; p.action_mask = a->action_mask; p.mask_dtype = a->mask_dtype; p.n_act = a->n_act > 0 ? a->n_act : 0;
  p.itp_bs = a->itp_bs > 0 ? a->itp_bs : (long long)(a->T + 1) * (a->T + 1); p.itp_rs = a->itp_rs > 0 ? a->itp_rs : a->T + 1;
  p.act_bs = a->act_bs > 0 ? a->act_bs : (long long)a->n_act * (a->T + 1 + a->n_act); p.act_rs = a->act_rs > 0 ? a->act_rs : a->T + 1 + a->n_act;
  p.n_mask_blocks = masks ? a->B * ((a->T + 1 + a->n_act + VLS_ROWS_PER_BLOCK - 1) / VLS_ROWS_PER_BLOCK) : 0;
  p.pos_vlm = a->pos_vlm; p.pos_pro = a->pos_pro; p.pos_act = a->pos_act;
  p.pos_vlm_out = a->pos_vlm_out; p.pos_pro_out = a->pos_pro_out; p.pos_act_out = a->pos_act_out; p.pos_ride_out = a->pos_ride_out;
  p.mask_slot = a->mask_slot; p.mask_ld = a->mask_ld;
  long long pb = (p.n_pix8 + 255) / 256;
  if (pb > 1024) pb = 1024;
  hipLaunchKernelGGL(vla_stage_kernel, dim3(1 + p.n_mask_blocks + (int)pb), dim3(256), 0, (hipStream_t)s, p);
  VL_LAUNCH_CHECK();
  return 0;
}

// EMA / SWA of the trained parameters (train.py:524-528, model_averaging.py:8-72 = torch.optim.swa_utils.AveragedModel): one streaming pass
// over the rank's fp32 master shard.  first: avg = p (AveragedModel copies on its first update); else avg += (p - avg) * c with
// c = 1 - decay (EMA, get_ema_multi_avg_fn: lerp) or 1 / (n_averaged + 1) (SWA).  One workgroup per CU, grid-stride (lesson 22).
__global__ __launch_bounds__(256) void avg_update_kernel(float* __restrict__ avg, const float* __restrict__ p, size_t n4, size_t n, float c, int first) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256;
  for (; i < n4; i += stride) {
    const f32x4 pv = reinterpret_cast<const f32x4*>(p)[i];
    f32x4 av = reinterpret_cast<f32x4*>(avg)[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) av[j] = first ? pv[j] : fmaf(c, pv[j] - av[j], av[j]);
    reinterpret_cast<f32x4*>(avg)[i] = av;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const size_t k = (n4 << 2) + threadIdx.x;
    avg[k] = first ? p[k] : fmaf(c, p[k] - avg[k], avg[k]);
  }
}
extern "C" int vlaser_avg_update(float* avg, const float* p, long long n, float c, int first, vl_stream_t s) {
  VL_CHECK(avg && p && n > 0, "vlaser_avg_update: bad args");
  VL_CHECK(((uintptr_t)avg & 15) == 0 && ((uintptr_t)p & 15) == 0, "vlaser_avg_update: alignment");
  const size_t n4 = (size_t)n >> 2;
  const int blocks = (int)((n4 + 255) / 256 < 256 ? (n4 + 255) / 256 + 1 : 256);
  hipLaunchKernelGGL(avg_update_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, avg, p, n4, (size_t)n, c, first);
  VL_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------- cross entropy rows
// loss_row[r] = logsumexp(logits[r,:]) - logits[r,label]  (0 when label == ignore_index); one block per row.
// CrossEntropyLoss of modeling_internvl_chat.py:231-243 = sum(loss_row) / count(label != -100).
__global__ __launch_bounds__(1024) void ce_rows_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels, int N,
                                                       long long ld, float* __restrict__ loss_row, float* __restrict__ lse_row,
                                                       long long ignore_index) {
  __shared__ float red[16];
  const int r = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t lab = labels[r];
  const float* row = logits + (size_t)r * ld;
  // r04: 8-byte loads, four in flight per thread (rows of the 151 674-wide vocabulary are 8-byte aligned at best): the one-float-per-trip loops were a chain of 148
  // load round trips per pass, 88 us for 128 rows
  const bool al8 = ((reinterpret_cast<uintptr_t>(row) & 7) == 0);
  const int N2 = al8 ? (N >> 1) : 0;                       // float2 pieces
  const f32x2_t* row2 = reinterpret_cast<const f32x2_t*>(row);
  float mx = -INFINITY;
  {
    int n = threadIdx.x;
    for (; n + 3 * 1024 < N2; n += 4 * 1024) {
      const f32x2_t a = row2[n], b = row2[n + 1024], c = row2[n + 2048], d = row2[n + 3072];
      mx = fmaxf(mx, fmaxf(fmaxf(fmaxf(a[0], a[1]), fmaxf(b[0], b[1])), fmaxf(fmaxf(c[0], c[1]), fmaxf(d[0], d[1]))));
    }
    for (; n < N2; n += 1024) { const f32x2_t a = row2[n]; mx = fmaxf(mx, fmaxf(a[0], a[1])); }
    for (int k = 2 * N2 + threadIdx.x; k < N; k += 1024) mx = fmaxf(mx, row[k]);
  }
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = red[0];
  for (int w = 1; w < 16; ++w) mx = fmaxf(mx, red[w]);
  __syncthreads();
  float s = 0.f;
  {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int n = threadIdx.x;
    for (; n + 3 * 1024 < N2; n += 4 * 1024) {
      const f32x2_t a = row2[n], b = row2[n + 1024], c = row2[n + 2048], d = row2[n + 3072];
      s0 += __expf(a[0] - mx) + __expf(a[1] - mx); s1 += __expf(b[0] - mx) + __expf(b[1] - mx);
      s2 += __expf(c[0] - mx) + __expf(c[1] - mx); s3 += __expf(d[0] - mx) + __expf(d[1] - mx);
    }
    for (; n < N2; n += 1024) { const f32x2_t a = row2[n]; s0 += __expf(a[0] - mx) + __expf(a[1] - mx); }
    for (int k = 2 * N2 + threadIdx.x; k < N; k += 1024) s1 += __expf(row[k] - mx);
    s = (s0 + s1) + (s2 + s3);
  }
  s = wave_sum(s);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.f;
    for (int w = 0; w < 16; ++w) tot += red[w];
    const float lse = mx + logf(tot);
    if (lse_row) lse_row[r] = lse;
    loss_row[r] = (lab == ignore_index) ? 0.f : lse - row[lab];
  }
}
extern "C" int vlaser_ce_rows(const float* logits, const int64_t* labels, int R, int N, long long ld, float* loss_row, float* lse_row,
                              long long ignore_index, vl_stream_t s) {
  VL_CHECK(logits && labels && loss_row && R > 0 && N > 0, "vlaser_ce_rows: bad args");
  hipLaunchKernelGGL(ce_rows_kernel, dim3(R), dim3(1024), 0, (hipStream_t)s, logits, labels, N, ld, loss_row, lse_row, ignore_index);
  VL_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------- split-K seam
// h = h_in + [ls *] (sum_s partials[s] [+ bias]); x_out = norm(h).  One wave per row, 4 rows per block; the row
// (C <= 4096) stays in registers between the reduction and the normalisation; slab loads are issued 4 slabs at a
// time (independent 16-byte loads) so the reduction costs ~ceil(S/4) L2 round trips.
// One wave per row and ONE row per workgroup: the path's seams have M = 384..1025 rows, so 4-row workgroups left two thirds of
// the CUs idle and made each busy CU pull 4 rows x S slabs through its own L1 (7.1-7.5 us per launch for 7-12 MB of traffic);
// SX > 0 = exact slab count (no clamped duplicate loads), SX = 0 = runtime count in batches of 8.
template <int NORM, int MAXC, int SX>  // NORM: 0 none, 1 RMS, 2 LayerNorm; MAXC: 16-byte chunks per lane (C <= 512*MAXC)
__global__ __launch_bounds__(64) void reduce_norm_kernel(const bf16_t* __restrict__ h_in, const float* __restrict__ partials, int S,
                                                          const bf16_t* __restrict__ bias, const bf16_t* __restrict__ ls,
                                                          const bf16_t* __restrict__ nw, const bf16_t* __restrict__ nb, float eps,
                                                          bf16_t* __restrict__ h_out, bf16_t* __restrict__ x_out, int M, int C) {
  const int lane = threadIdx.x & 63, row = blockIdx.x;
  if (row >= M) return;
  u32x4 hv[MAXC];
  const size_t slab = (size_t)M * C;
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = (lane + 64 * i) * 8;
    if (c < C) {
      const size_t off = (size_t)row * C + c;
      float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      if constexpr (SX > 0) {
        f32x4 q[2 * SX];
#pragma unroll
        for (int u = 0; u < SX; ++u) {
          const float* pp = partials + off + (size_t)u * slab;
          q[2 * u] = *reinterpret_cast<const f32x4*>(pp);
          q[2 * u + 1] = *reinterpret_cast<const f32x4*>(pp + 4);
        }
#pragma unroll
        for (int u = 0; u < SX; ++u)
#pragma unroll
          for (int j = 0; j < 4; ++j) { v[j] += q[2 * u][j]; v[4 + j] += q[2 * u + 1][j]; }
      } else if (S > 0) {
        const float* pb = partials + off;
        int Sr = S;
        for (; Sr > 8; Sr -= 8, pb += 8 * slab) add_slabs_clamped<8>(v, pb, slab, 8);
        add_slabs_clamped<8>(v, pb, slab, Sr);
      }
      u32x4 hi = {0, 0, 0, 0};
      if (h_in) hi = ld_global_16(h_in + off);
      u32x4 bv = {0, 0, 0, 0}, lv = {0, 0, 0, 0};
      if (bias) bv = ld_global_16(bias + c);
      if (ls) lv = ld_global_16(ls + c);
      u32x4 hr;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float lo = v[2 * j] + bf16lo_to_f32(bv[j]), hi2 = v[2 * j + 1] + bf16hi_to_f32(bv[j]);
        if (ls) { lo *= bf16lo_to_f32(lv[j]); hi2 *= bf16hi_to_f32(lv[j]); }
        hr[j] = pack_bf16x2(bf16lo_to_f32(hi[j]) + lo, bf16hi_to_f32(hi[j]) + hi2);
        const float a0 = bf16lo_to_f32(hr[j]), a1 = bf16hi_to_f32(hr[j]);
        s1 += a0 + a1;
        s2 += a0 * a0 + a1 * a1;
      }
      hv[i] = hr;
      st_global_16(h_out + off, hr);
    }
  }
  if constexpr (NORM == 0) return;
  s1 = wave_sum(s1);
  float mean = 0.f, rs;
  if constexpr (NORM == 1) {
    rs = rsqrtf(wave_sum(s2) / (float)C + eps);
  } else {
    mean = s1 / (float)C;
    float vs = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i)
      if ((lane + 64 * i) * 8 < C)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float a0 = bf16lo_to_f32(hv[i][j]) - mean, a1 = bf16hi_to_f32(hv[i][j]) - mean;
          vs += a0 * a0 + a1 * a1;
        }
    rs = rsqrtf(wave_sum(vs) / (float)C + eps);
  }
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = (lane + 64 * i) * 8;
    if (c < C) {
      const u32x4 wv = ld_global_16(nw + c);
      u32x4 bv = {0, 0, 0, 0};
      if constexpr (NORM == 2) bv = ld_global_16(nb + c);
      u32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float lo, hi;
        if constexpr (NORM == 1) {
          lo = round_bf16(bf16lo_to_f32(hv[i][j]) * rs) * bf16lo_to_f32(wv[j]);
          hi = round_bf16(bf16hi_to_f32(hv[i][j]) * rs) * bf16hi_to_f32(wv[j]);
        } else {
          lo = (bf16lo_to_f32(hv[i][j]) - mean) * rs * bf16lo_to_f32(wv[j]) + bf16lo_to_f32(bv[j]);
          hi = (bf16hi_to_f32(hv[i][j]) - mean) * rs * bf16hi_to_f32(wv[j]) + bf16hi_to_f32(bv[j]);
        }
        o[j] = pack_bf16x2(lo, hi);
      }
      st_global_16(x_out + (size_t)row * C + c, o);
    }
  }
}
// Exact-shape variant: C = NCH x 512 columns, SX slabs, residual present, (bias, layer-scale) both present (BL) or both absent --
// every load of the row (slabs, residual, bias / layer-scale, norm weights) is issued up front in straight-line code.  The generic
// kernel above walks the row's 512-column chunks behind `c < C` / `if (ptr)` branches, i.e. one L2 round trip per chunk plus one
// more for the norm weights after the statistics: 6.5-8.3 us per launch on the path's seams against 4.x us here.
template <int NORM, int NCH, int SX, bool BL>
__global__ __launch_bounds__(64) void reduce_norm_exact_kernel(const bf16_t* __restrict__ h_in, const float* __restrict__ partials,
                                                                const bf16_t* __restrict__ bias, const bf16_t* __restrict__ ls,
                                                                const bf16_t* __restrict__ nw, const bf16_t* __restrict__ nb, float eps,
                                                                bf16_t* __restrict__ h_out, bf16_t* __restrict__ x_out, int M) {
  constexpr int C = NCH * 512;
  const int lane = threadIdx.x & 63, row = blockIdx.x;
  const size_t slab = (size_t)M * C;
  f32x4 q[NCH][2 * SX];
  u32x4 hi[NCH], bv[NCH], lv[NCH], wv[NCH], nbv[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = (lane + 64 * i) * 8;
    const size_t off = (size_t)row * C + c;
#pragma unroll
    for (int u = 0; u < SX; ++u) {
      const float* pp = partials + off + (size_t)u * slab;
      q[i][2 * u] = *reinterpret_cast<const f32x4*>(pp);
      q[i][2 * u + 1] = *reinterpret_cast<const f32x4*>(pp + 4);
    }
    hi[i] = ld_global_16(h_in + off);
    if constexpr (BL) { bv[i] = ld_global_16(bias + c); lv[i] = ld_global_16(ls + c); }
    if constexpr (NORM != 0) wv[i] = ld_global_16(nw + c);
    if constexpr (NORM == 2) nbv[i] = ld_global_16(nb + c);
  }
  u32x4 hv[NCH];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < SX; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[j] += q[i][2 * u][j]; v[4 + j] += q[i][2 * u + 1][j]; }
    u32x4 hr;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float lo = v[2 * j], hi2 = v[2 * j + 1];
      if constexpr (BL) {
        lo = (lo + bf16lo_to_f32(bv[i][j])) * bf16lo_to_f32(lv[i][j]);
        hi2 = (hi2 + bf16hi_to_f32(bv[i][j])) * bf16hi_to_f32(lv[i][j]);
      }
      hr[j] = pack_bf16x2(bf16lo_to_f32(hi[i][j]) + lo, bf16hi_to_f32(hi[i][j]) + hi2);
      const float a0 = bf16lo_to_f32(hr[j]), a1 = bf16hi_to_f32(hr[j]);
      s1 += a0 + a1;
      s2 += a0 * a0 + a1 * a1;
    }
    hv[i] = hr;
    st_global_16(h_out + (size_t)row * C + (lane + 64 * i) * 8, hr);
  }
  if constexpr (NORM == 0) return;
  s1 = wave_sum(s1);
  float mean = 0.f, rs;
  if constexpr (NORM == 1) {
    rs = rsqrtf(wave_sum(s2) / (float)C + eps);
  } else {
    mean = s1 / (float)C;
    float vs = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float a0 = bf16lo_to_f32(hv[i][j]) - mean, a1 = bf16hi_to_f32(hv[i][j]) - mean;
        vs += a0 * a0 + a1 * a1;
      }
    rs = rsqrtf(wave_sum(vs) / (float)C + eps);
  }
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float lo, hi2;
      if constexpr (NORM == 1) {
        lo = round_bf16(bf16lo_to_f32(hv[i][j]) * rs) * bf16lo_to_f32(wv[i][j]);
        hi2 = round_bf16(bf16hi_to_f32(hv[i][j]) * rs) * bf16hi_to_f32(wv[i][j]);
      } else {
        lo = (bf16lo_to_f32(hv[i][j]) - mean) * rs * bf16lo_to_f32(wv[i][j]) + bf16lo_to_f32(nbv[i][j]);
        hi2 = (bf16hi_to_f32(hv[i][j]) - mean) * rs * bf16hi_to_f32(wv[i][j]) + bf16hi_to_f32(nbv[i][j]);
      }
      o[j] = pack_bf16x2(lo, hi2);
    }
    st_global_16(x_out + (size_t)row * C + (lane + 64 * i) * 8, o);
  }
}

template <int NORM, int NCH, bool BL>
static bool launch_reduce_norm_exact(int S, dim3 grid, hipStream_t st, const bf16_t* h_in, const float* partials, const bf16_t* bias, const bf16_t* ls,
                                     const bf16_t* nw, const bf16_t* nb, float eps, bf16_t* h_out, bf16_t* x_out, int M) {
#define RNE(SX_) case SX_: hipLaunchKernelGGL((reduce_norm_exact_kernel<NORM, NCH, SX_, BL>), grid, dim3(64), 0, st, h_in, partials, bias, ls, nw, nb, eps, h_out, x_out, M); return true;
  switch (S) {
    RNE(1) RNE(2) RNE(3) RNE(4)
    default: break;
  }
  if constexpr (NCH <= 3) {
    switch (S) {
      RNE(5) RNE(6) RNE(7) RNE(8)
      default: break;
    }
  }
#undef RNE
  return false;
}

extern "C" int vlaser_reduce_norm(const void* h_in, const float* partials, int S, const void* bias, const void* ls, int norm_kind,
                                  const void* nw, const void* nb, float eps, void* h_out, void* x_out, int M, int C, vl_stream_t s) {
  VL_CHECK(h_out && (S == 0 || partials) && M > 0 && C % 8 == 0 && C <= 4096, "vlaser_reduce_norm: bad args (C=%d)", C);
  VL_CHECK(norm_kind == 0 || (nw && x_out && (norm_kind == 1 || nb)), "vlaser_reduce_norm: norm weights / x_out missing");
  dim3 grid(M), blk(64);
  if (h_in && S >= 1 && ((bias != nullptr) == (ls != nullptr))) {
    // exact-shape kernels for the path's widths: ViT 1024 (bias + layer-scale, LayerNorm / none), Qwen2.5 1536 and 3584 (RMSNorm / none)
    const bf16_t *hi_ = (const bf16_t*)h_in, *b_ = (const bf16_t*)bias, *l_ = (const bf16_t*)ls, *nw_ = (const bf16_t*)nw, *nb_ = (const bf16_t*)nb;
    bf16_t *ho_ = (bf16_t*)h_out, *xo_ = (bf16_t*)x_out;
    hipStream_t st = (hipStream_t)s;
    bool done = false;
    if (C == 1024 && bias && norm_kind == 2) done = launch_reduce_norm_exact<2, 2, true>(S, grid, st, hi_, partials, b_, l_, nw_, nb_, eps, ho_, xo_, M);
    else if (C == 1024 && bias && norm_kind == 0) done = launch_reduce_norm_exact<0, 2, true>(S, grid, st, hi_, partials, b_, l_, nw_, nb_, eps, ho_, xo_, M);
    else if (C == 1536 && !bias && norm_kind == 1) done = launch_reduce_norm_exact<1, 3, false>(S, grid, st, hi_, partials, b_, l_, nw_, nb_, eps, ho_, xo_, M);
    else if (C == 1536 && !bias && norm_kind == 0) done = launch_reduce_norm_exact<0, 3, false>(S, grid, st, hi_, partials, b_, l_, nw_, nb_, eps, ho_, xo_, M);
    else if (C == 3584 && !bias && norm_kind == 1) done = launch_reduce_norm_exact<1, 7, false>(S, grid, st, hi_, partials, b_, l_, nw_, nb_, eps, ho_, xo_, M);
    else if (C == 3584 && !bias && norm_kind == 0) done = launch_reduce_norm_exact<0, 7, false>(S, grid, st, hi_, partials, b_, l_, nw_, nb_, eps, ho_, xo_, M);
    if (done) { VL_LAUNCH_CHECK(); return 0; }
  }
#define RN_ARGS (const bf16_t*)h_in, partials, S, (const bf16_t*)bias, (const bf16_t*)ls, (const bf16_t*)nw, (const bf16_t*)nb, eps, \
                (bf16_t*)h_out, (bf16_t*)x_out, M, C
#define RN_LAUNCH(MAXC, SX)                                                                                                   \
  do {                                                                                                                        \
    if (norm_kind == 0) hipLaunchKernelGGL((reduce_norm_kernel<0, MAXC, SX>), grid, blk, 0, (hipStream_t)s, RN_ARGS);         \
    else if (norm_kind == 1) hipLaunchKernelGGL((reduce_norm_kernel<1, MAXC, SX>), grid, blk, 0, (hipStream_t)s, RN_ARGS);    \
    else hipLaunchKernelGGL((reduce_norm_kernel<2, MAXC, SX>), grid, blk, 0, (hipStream_t)s, RN_ARGS);                        \
  } while (0)
#define RN_SWITCH(MAXC)                            \
  switch (S) {                                     \
    case 1: RN_LAUNCH(MAXC, 1); break;             \
    case 2: RN_LAUNCH(MAXC, 2); break;             \
    case 3: RN_LAUNCH(MAXC, 3); break;             \
    case 4: RN_LAUNCH(MAXC, 4); break;             \
    case 5: RN_LAUNCH(MAXC, 5); break;             \
    case 6: RN_LAUNCH(MAXC, 6); break;             \
    case 7: RN_LAUNCH(MAXC, 7); break;             \
    case 8: RN_LAUNCH(MAXC, 8); break;             \
    default: RN_LAUNCH(MAXC, 0); break;            \
  }
  if (C <= 2048) { RN_SWITCH(4) } else { RN_SWITCH(8) }
#undef RN_SWITCH
#undef RN_LAUNCH
#undef RN_ARGS
  VL_LAUNCH_CHECK();
  return 0;
}
