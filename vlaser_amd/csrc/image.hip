// Image preparation on the device (ABI 8): the bicubic resize of `dynamic_preprocess` / `build_transform` (Vlaser_VLM/internvl_chat/internvl/train/dataset.py:276-310,830-866;
// eval_example.py:38-82) and the cut into normalised 448-px tiles.  The reference runs this on the host through Pillow (`Image.resize`, BICUBIC): 8-bit fixed-point resampling,
// two separable passes with the intermediate image rounded to 8 bits (Pillow src/libImaging/Resample.c; restated in oracle/resize.py, pinned against Pillow itself).  Byte
// work, bit-exact: the weights are computed on the HOST in doubles exactly as Pillow computes them (vlaser_resample_coeffs: a table of out_size x ksize int32, no image data),
// the passes run here.  HBM-bound: a pass reads its input once (windows of neighbouring outputs overlap in LDS / L2) and writes its output once.
#include "common.h"
#include "../../include/vlaser_hip.h"

#include <math.h>

#define VL_RS_BITS 22                      // Pillow's PRECISION_BITS = 32 - 8 - 2

// ------------------------------------------------------------------------------------------------------------------ host: the weight tables
#pragma clang fp contract(off)             // (x86-64 baseline has no FMA anyway: the doubles below must round like Pillow's own build)
static double vl_bicubic(double x) {
  const double a = -0.5;
  if (x < 0.0) x = -x;
  if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
  if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
  return 0.0;
}

extern "C" int vlaser_resample_ksize(int in_size, int out_size) {
  if (in_size < 1 || out_size < 1) return -1;
  double filterscale = (double)(float)in_size / out_size;
  if (filterscale < 1.0) filterscale = 1.0;
  return (int)ceil(2.0 * filterscale) * 2 + 1;
}

extern "C" int vlaser_resample_coeffs(int in_size, int out_size, int* bounds, int* kk_t) {
  VL_CHECK(in_size >= 1 && out_size >= 1 && in_size < (1 << 24) && bounds && kk_t, "vlaser_resample_coeffs: sizes in [1, 2^24), non-null tables");
  const double scale = (double)(float)in_size / out_size;
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = 2.0 * filterscale;
  const int ksize = (int)ceil(support) * 2 + 1;
  const double ss = 1.0 / filterscale;
  double* k = new double[ksize];
  bool big = false;
  for (int xx = 0; xx < out_size; ++xx) {
    const double center = 0.0 + (xx + 0.5) * scale;
    double ww = 0.0;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    for (int x = 0; x < xmax; ++x) {
      const double w = vl_bicubic((x + xmin - center + 0.5) * ss);
      k[x] = w;
      ww += w;
    }
    for (int x = 0; x < xmax; ++x)
      if (ww != 0.0) k[x] /= ww;
    for (int x = 0; x < ksize; ++x) {
      const double v = x < xmax ? k[x] : 0.0;
      const int q = v < 0 ? (int)(-0.5 + v * (1 << VL_RS_BITS)) : (int)(0.5 + v * (1 << VL_RS_BITS));
      if (q >= (1 << 23) || q < -(1 << 23)) big = true;            // the device multiplies pixel x weight with 24-bit operands (full rate; a 32-bit integer multiply is not)
      kk_t[(size_t)x * out_size + xx] = q;
    }
    bounds[2 * xx] = xmin;
    bounds[2 * xx + 1] = xmax;
  }
  delete[] k;
  VL_CHECK(!big, "vlaser_resample_coeffs: a normalised weight of %d -> %d reaches 2.0 (24-bit multiplies on the device)", in_size, out_size);
  return ksize;
}

// ------------------------------------------------------------------------------------------------------------------ device: the two passes
// COMPILER TRAP (hipcc / ROCm 7.2, found on the GPU in r06): `clip(a0 >> 22) | clip(a1 >> 22) << 8 | clip(a2 >> 22) << 16 | ...` is matched to gfx950's
// `v_ashr_pk_u8_i32 d, a0, a1, 22` and OR-ed with the other bytes as if the instruction zeroed bits 31:16 of d -- on the hardware they come back non-zero, and bytes 2 / 3 of
// every stored dword were garbage (the byte-store paths were right).  An empty asm statement between the shift and the clamp keeps the pattern from forming.
__device__ __forceinline__ uint32_t vl_clip8(int v) {
  int x = v >> VL_RS_BITS;
  asm volatile("" : "+v"(x));
  return (uint32_t)min(max(x, 0), 255);
}

struct ResampleP {
  const uint8_t* src; uint8_t* dst;
  const int* bounds; const int* kk;         // [2 * out] (first, count); [ksize][out]
  long long ld_in, ld_out;                  // bytes per row
  int out_n, ksize, rows, row_bytes, xb;
#ifdef VL_RS_LAB
  int lab;                                  // tools/micro/resize_lab.hip: 1 no stores, 2 no staging loads, 4 one tap, 8 no weight loads
#endif
};
#ifdef VL_RS_LAB
#define VL_RS_LABBIT(b) (p.lab & (b))
#else
#define VL_RS_LABBIT(b) false
#endif

// Multiplies: pixel (8 bits) x weight (|k| < 2.0 in 22-bit fixed point = 24 bits signed, checked when the table is built) as v_mad_i32_i24 -- full rate; the plain `int * int`
// compiles to v_mad_u64_u32, and 24 of them per tap made the first version 5 x slower than its loads and stores (tools/micro/resize_lab.hip: 7 us per tap).
// Horizontal: one workgroup = VL_RS_ROWS input rows x `xb` consecutive output pixels.  The input windows of those outputs (monotonic in the output index) are staged in LDS
// once per row; a thread owns ONE output column and walks its taps once for all the rows: the weight (one coalesced, cache-served load per tap) and the index arithmetic
// are shared by the rows, a tap of a row is one LDS dword (the 3 interleaved channel bytes + 1) and three multiply-adds.  r06 first version: one row per workgroup, three
// LDS byte reads and a weight load per tap and row -- 95 us for the 12-megapixel frame of BASELINE's 13-tile case; this one: see profiles/r06v_image_lab.md.
#define VL_RS_ROWS 8
// LDS, per row: the raw bytes of the window (dword copies of the image row) and, expanded from them, ONE ALIGNED DWORD PER PIXEL (R, G, B, next R).  A tap reads its pixel
// as that dword.  Measured (tools/micro/resize_lab.hip, the 12-megapixel frame): walking the raw bytes with dword reads at byte offsets 3 k -- which the compiler does emit as
// `ds_read_b32`, the target allowing unaligned LDS access -- costs 100 us per pass against 39 us with aligned reads: an unaligned LDS dword is several times the price
// of an aligned one, and so were the three `ds_read_u8` of the first version.
__global__ __launch_bounds__(256) void resample_h_kernel(ResampleP p) {
  extern __shared__ __attribute__((aligned(16))) uint8_t span[];
  const int y0 = blockIdx.y * VL_RS_ROWS, x0 = blockIdx.x * p.xb, x1 = min(x0 + p.xb, p.out_n);
  const int first = p.bounds[2 * x0], last = p.bounds[2 * (x1 - 1)] + p.bounds[2 * (x1 - 1) + 1];
  const int npx = last - first, nbytes = npx * 3;
  // a row's window as whole dwords from its dword-aligned-down start (`pad` bytes early: still inside the image when the image itself is dword-aligned), bytes for the tail
  const bool al = ((uintptr_t)p.src & 3) == 0;
  const int rs = (nbytes + 3 + 8 + 15) & ~15;                  // raw bytes per row (16-byte multiple): pad (<= 3) + window + the dword pair the last pixel's expansion reads
  uint32_t* px = reinterpret_cast<uint32_t*>(span + VL_RS_ROWS * rs);
  if (!VL_RS_LABBIT(2)) {
    // 32 threads per row, all rows at once: 16-byte pieces of the window, four per thread in flight before the first LDS store (one row after the other with dword loads
    // was 23 of the pass's 47 us: eight dependent memory round trips per workgroup)
    const int r = threadIdx.x >> 5, l = threadIdx.x & 31;
    const uint8_t* row = p.src + (size_t)min(y0 + r, p.rows - 1) * p.ld_in + (size_t)first * 3;
    const int pad = al ? (int)((uintptr_t)row & 3) : 0;
    const uint8_t* arow = row - pad;
    const int tot = pad + nbytes, nw = al ? tot >> 2 : 0, nq = al ? nbytes >> 4 : 0;        // (nq from nbytes: the same count for every row, whatever its pad)
    uint8_t* dst = span + r * rs;
    typedef uint32_t __attribute__((ext_vector_type(4), aligned(4))) u32x4_a4;           // global_load_dwordx4 needs dword alignment only
    for (int i0 = 0; i0 < nq; i0 += 128) {
      u32x4 v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int i = i0 + l + 32 * j;
        if (i < nq) v[j] = reinterpret_cast<const u32x4_a4*>(arow)[i];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int i = i0 + l + 32 * j;
        if (i < nq) reinterpret_cast<u32x4*>(dst)[i] = v[j];
      }
    }
    for (int i = 4 * nq + l; i < nw; i += 32) reinterpret_cast<uint32_t*>(dst)[i] = reinterpret_cast<const uint32_t*>(arow)[i];
    for (int i = 4 * nw + l; i < tot; i += 32) dst[i] = arow[i];
  }
  __syncthreads();
#pragma unroll 1
  for (int r = 0; r < VL_RS_ROWS; ++r) {
    const uint8_t* row = p.src + (size_t)min(y0 + r, p.rows - 1) * p.ld_in + (size_t)first * 3;
    const int pad = al ? (int)((uintptr_t)row & 3) : 0;
    for (int q = threadIdx.x; q < npx; q += 256) {
      const int b = pad + 3 * q;                               // byte offset of pixel q in the raw row
      const uint32_t* wp = reinterpret_cast<const uint32_t*>(span + r * rs + (b & ~3));
      px[r * npx + q] = __builtin_amdgcn_alignbyte(wp[1], wp[0], (uint32_t)(b & 3));
    }
  }
  __syncthreads();
  for (int xx = x0 + threadIdx.x; xx < x1; xx += 256) {
    const int xmin = p.bounds[2 * xx] - first, n = p.bounds[2 * xx + 1];
    int acc[VL_RS_ROWS][3];
#pragma unroll
    for (int r = 0; r < VL_RS_ROWS; ++r) acc[r][0] = acc[r][1] = acc[r][2] = 1 << (VL_RS_BITS - 1);
    const uint32_t* s = px + xmin;
    const int* kk = p.kk + xx;
    for (int k = 0; k < (VL_RS_LABBIT(4) ? 1 : n); ++k) {
      const int c = VL_RS_LABBIT(8) ? k + 77 : *kk;
      kk += p.out_n;
#pragma unroll
      for (int r = 0; r < VL_RS_ROWS; ++r) {
        const uint32_t w = s[r * npx + k];
        acc[r][0] += __mul24((int)(w & 255), c); acc[r][1] += __mul24((int)((w >> 8) & 255), c); acc[r][2] += __mul24((int)((w >> 16) & 255), c);
      }
    }
#pragma unroll
    for (int r = 0; r < VL_RS_ROWS; ++r)
      if (y0 + r < p.rows && !(VL_RS_LABBIT(1) && acc[r][0] != 12345)) {
        uint8_t* out = p.dst + (size_t)(y0 + r) * p.ld_out + 3 * xx;
        out[0] = (uint8_t)vl_clip8(acc[r][0]); out[1] = (uint8_t)vl_clip8(acc[r][1]); out[2] = (uint8_t)vl_clip8(acc[r][2]);
      }
  }
}

// Windows that do not fit the LDS stage even for a single output column (downscales by hundreds): one thread per output pixel straight from memory.
__global__ __launch_bounds__(256) void resample_h_direct_kernel(ResampleP p) {
  const int xx = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (xx >= p.out_n) return;
  const int xmin = p.bounds[2 * xx], n = p.bounds[2 * xx + 1];
  const uint8_t* s = p.src + (size_t)y * p.ld_in + (size_t)xmin * 3;
  const int* kk = p.kk + xx;
  int a0 = 1 << (VL_RS_BITS - 1), a1 = a0, a2 = a0;
  for (int k = 0; k < n; ++k) {
    const int c = kk[(size_t)k * p.out_n];
    a0 += __mul24((int)s[3 * k], c); a1 += __mul24((int)s[3 * k + 1], c); a2 += __mul24((int)s[3 * k + 2], c);
  }
  uint8_t* out = p.dst + (size_t)y * p.ld_out + 3 * xx;
  out[0] = (uint8_t)vl_clip8(a0); out[1] = (uint8_t)vl_clip8(a1); out[2] = (uint8_t)vl_clip8(a2);
}

// Vertical: a row of the image is `row_bytes` independent byte columns; one thread = 4 of them (a dword per tap, coalesced), one workgroup row = one output row (its window
// and weights are uniform: scalar loads).  VEC = false: one byte per thread (rows that are not dword-aligned).
template <bool VEC>
__global__ __launch_bounds__(256) void resample_v_kernel(ResampleP p) {
  const int yy = blockIdx.y;
  const int ymin = p.bounds[2 * yy], n = p.bounds[2 * yy + 1];
  const int* kk = p.kk + yy;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if constexpr (VEC) {
    const int b0 = 4 * i;
    if (b0 >= p.row_bytes) return;
    const uint8_t* s = p.src + (size_t)ymin * p.ld_in + b0;
    uint8_t* d = p.dst + (size_t)yy * p.ld_out + b0;
    int a0 = 1 << (VL_RS_BITS - 1), a1 = a0, a2 = a0, a3 = a0;
    if (b0 + 4 <= p.row_bytes) {
      for (int k = 0; k < n; ++k) {
        const int c = kk[(size_t)k * p.out_n];
        const uint32_t w = *reinterpret_cast<const uint32_t*>(s + (size_t)k * p.ld_in);
        a0 += __mul24((int)(w & 255), c); a1 += __mul24((int)((w >> 8) & 255), c); a2 += __mul24((int)((w >> 16) & 255), c); a3 += __mul24((int)(w >> 24), c);
      }
      *reinterpret_cast<uint32_t*>(d) = vl_clip8(a0) | (vl_clip8(a1) << 8) | (vl_clip8(a2) << 16) | (vl_clip8(a3) << 24);
    } else {
      const int nb = p.row_bytes - b0;                    // 1..3 trailing bytes of the row
      for (int k = 0; k < n; ++k) {
        const int c = kk[(size_t)k * p.out_n];
        const uint8_t* q = s + (size_t)k * p.ld_in;
        a0 += __mul24((int)q[0], c);
        if (nb > 1) a1 += __mul24((int)q[1], c);
        if (nb > 2) a2 += __mul24((int)q[2], c);
      }
      d[0] = (uint8_t)vl_clip8(a0);
      if (nb > 1) d[1] = (uint8_t)vl_clip8(a1);
      if (nb > 2) d[2] = (uint8_t)vl_clip8(a2);
    }
  } else {
    if (i >= p.row_bytes) return;
    const uint8_t* s = p.src + (size_t)ymin * p.ld_in + i;
    int a = 1 << (VL_RS_BITS - 1);
    for (int k = 0; k < n; ++k) a += __mul24((int)s[(size_t)k * p.ld_in], kk[(size_t)k * p.out_n]);
    p.dst[(size_t)yy * p.ld_out + i] = (uint8_t)vl_clip8(a);
  }
}

static int launch_h(const uint8_t* src, long long ld_in, int rows, int in_w, uint8_t* dst, long long ld_out, int out_w, const int* bounds, const int* kk, int ksize,
                    hipStream_t stream) {
  ResampleP p;
  p.src = src; p.dst = dst; p.bounds = bounds; p.kk = kk; p.ld_in = ld_in; p.ld_out = ld_out; p.out_n = out_w; p.ksize = ksize; p.rows = rows; p.row_bytes = out_w * 3;
  // outputs per workgroup: 256 unless the VL_RS_ROWS joint windows (raw bytes + one dword per pixel) would not fit 60 KB of LDS (strong downscales)
  const double scale = (double)in_w / out_w, fs = scale > 1.0 ? scale : 1.0;
  auto row_lds = [&](int xb) { const long long px = (long long)(xb * fs + ksize + 2); return ((px * 3 + 26 + 15) & ~15LL) + px * 4; };
  int xb = 256;
  while (xb > 1 && row_lds(xb) * VL_RS_ROWS > 60 * 1024) xb >>= 1;
  if (row_lds(xb) * VL_RS_ROWS > 60 * 1024) {
    hipLaunchKernelGGL(resample_h_direct_kernel, dim3((out_w + 255) / 256, rows), dim3(256), 0, stream, p);
    VL_LAUNCH_CHECK();
    return 0;
  }
  p.xb = xb;
  const int lds = (int)(row_lds(xb) * VL_RS_ROWS);
  if (int rc = set_max_lds_once(resample_h_kernel, lds)) return rc;
  hipLaunchKernelGGL(resample_h_kernel, dim3((out_w + xb - 1) / xb, (rows + VL_RS_ROWS - 1) / VL_RS_ROWS), dim3(256), lds, stream, p);
  VL_LAUNCH_CHECK();
  return 0;
}

static int launch_v(const uint8_t* src, long long ld_in, uint8_t* dst, long long ld_out, int out_h, int row_bytes, const int* bounds, const int* kk, int ksize, hipStream_t stream) {
  ResampleP p;
  p.src = src; p.dst = dst; p.bounds = bounds; p.kk = kk; p.ld_in = ld_in; p.ld_out = ld_out; p.out_n = out_h; p.ksize = ksize; p.rows = out_h; p.row_bytes = row_bytes;
  p.xb = 0;
  const bool vec = ld_in % 4 == 0 && ld_out % 4 == 0 && ((uintptr_t)src & 3) == 0 && ((uintptr_t)dst & 3) == 0;
  if (vec) hipLaunchKernelGGL((resample_v_kernel<true>), dim3(((row_bytes + 3) / 4 + 255) / 256, out_h), dim3(256), 0, stream, p);
  else hipLaunchKernelGGL((resample_v_kernel<false>), dim3((row_bytes + 255) / 256, out_h), dim3(256), 0, stream, p);
  VL_LAUNCH_CHECK();
  return 0;
}

extern "C" int vlaser_resize_u8(const void* src, int H, int W, long long ld_src, void* tmp, long long ld_tmp, void* dst, int h, int w, long long ld_dst, const int* bounds_x,
                                const int* kk_x, int ksize_x, const int* bounds_y, const int* kk_y, int ksize_y, vl_stream_t s) {
  VL_CHECK(src && dst && H >= 1 && W >= 1 && h >= 1 && w >= 1, "vlaser_resize_u8: null image / empty size");
  VL_CHECK(ld_src >= (long long)W * 3 && ld_dst >= (long long)w * 3, "vlaser_resize_u8: row strides shorter than a row of RGB bytes");
  VL_CHECK((w == W || (bounds_x && kk_x && ksize_x == vlaser_resample_ksize(W, w))) && (h == H || (bounds_y && kk_y && ksize_y == vlaser_resample_ksize(H, h))),
           "vlaser_resize_u8: weight tables missing or built for other sizes (vlaser_resample_coeffs)");
  VL_CHECK(!(w != W && h != H) || (tmp && ld_tmp >= (long long)w * 3), "vlaser_resize_u8: both axes change: the [H, w, 3] intermediate image is needed");
  hipStream_t stream = (hipStream_t)s;
  if (w == W && h == H) {                   // Pillow: same size = a copy
    VL_HIP(hipMemcpy2DAsync(dst, (size_t)ld_dst, src, (size_t)ld_src, (size_t)W * 3, (size_t)H, hipMemcpyDeviceToDevice, stream));
    return 0;
  }
  if (w != W && h == H) return launch_h((const uint8_t*)src, ld_src, H, W, (uint8_t*)dst, ld_dst, w, bounds_x, kk_x, ksize_x, stream);
  if (w == W) return launch_v((const uint8_t*)src, ld_src, (uint8_t*)dst, ld_dst, h, w * 3, bounds_y, kk_y, ksize_y, stream);
  if (int rc = launch_h((const uint8_t*)src, ld_src, H, W, (uint8_t*)tmp, ld_tmp, w, bounds_x, kk_x, ksize_x, stream)) return rc;
  return launch_v((const uint8_t*)tmp, ld_tmp, (uint8_t*)dst, ld_dst, h, w * 3, bounds_y, kk_y, ksize_y, stream);
}

// ------------------------------------------------------------------------------------------------------------------ tiles -> normalised pixel_values
// [rows * tile, cols * tile, 3] uint8 (row stride ld) -> bf16 [cols * rows, 3, tile, tile]: the crop loop of dynamic_preprocess (dataset.py:851-862) + ToTensor + Normalize
// (:297-299) in one pass; arithmetic of vlaser_normalize_u8 (mode 0 / 1), 4 pixels x 3 channels per thread (12 bytes in, 3 x 8 bytes out).
__global__ __launch_bounds__(256) void tiles_normalize_kernel(const uint8_t* __restrict__ src, long long ld, int cols, int tile, bf16_t* __restrict__ out, int mode, float m0, float m1,
                                                              float m2, float s0, float s1, float s2, size_t total) {
  const int t4 = tile >> 2;
  const float r255 = (float)(1.0 / 255.0);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int x4 = (int)(i % t4);
    const size_t r = i / t4;
    const int y = (int)(r % tile), n = (int)(r / tile);
    const int ty = n / cols, tx = n - ty * cols;
    const uint32_t* s = reinterpret_cast<const uint32_t*>(src + ((size_t)ty * tile + y) * ld + ((size_t)tx * tile + 4 * x4) * 3);
    const uint32_t w[3] = {s[0], s[1], s[2]};
    uint8_t b[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) b[j] = (w[j >> 2] >> (8 * (j & 3))) & 255;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float x = mode == 0 ? (float)b[3 * j + c] * r255 : (float)b[3 * j + c] / 255.0f;
        v[j] = (x - mean) / sd;
      }
      *reinterpret_cast<u32x2*>(out + (((size_t)n * 3 + c) * tile + y) * tile + 4 * x4) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
    }
  }
}

extern "C" int vlaser_tiles_normalize_u8(const void* src_u8, long long ld, int cols, int rows, int tile, void* out_bf16, int mode, const float* mean3, const float* std3,
                                         vl_stream_t s) {
  VL_CHECK(src_u8 && out_bf16 && mean3 && std3 && cols >= 1 && rows >= 1 && tile >= 4 && tile % 4 == 0, "vlaser_tiles_normalize_u8: bad args (tile must be a multiple of 4)");
  VL_CHECK(ld >= (long long)cols * tile * 3 && ld % 4 == 0 && ((uintptr_t)src_u8 & 3) == 0 && ((uintptr_t)out_bf16 & 7) == 0 && (mode == 0 || mode == 1),
           "vlaser_tiles_normalize_u8: row stride (multiple of 4, >= cols * tile * 3) / alignment / mode");
  const size_t total = (size_t)cols * rows * tile * (tile / 4);
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(tiles_normalize_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, (const uint8_t*)src_u8, ld, cols, tile, (bf16_t*)out_bf16, mode, mean3[0], mean3[1],
                     mean3[2], std3[0], std3[1], std3[2], total);
  VL_LAUNCH_CHECK();
  return 0;
}
