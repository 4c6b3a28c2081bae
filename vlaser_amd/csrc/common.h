// Shared device helpers for the Vlaser gfx950 (MI355X / CDNA4) kernels.
// Wave = 64 lanes everywhere; MFMA = v_mfma_f32_16x16x32_bf16 (A: lane l -> row l&15, k (l>>4)*8..+8;
// B: lane l -> col l&15, same k; C/D: col l&15, rows (l>>4)*4 + reg).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <map>
#include <mutex>
#include <utility>

typedef uint16_t bf16_t;  // raw bf16 bits in memory
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define WAVE 64

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ float bf16lo_to_f32(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16hi_to_f32(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// round-to-nearest-even f32 -> bf16: gfx950 has the conversion in hardware (v_cvt_pk_bf16_f32, two values per
// instruction); hipcc selects it for __bf16 conversions.  A software RNE costs ~7 VALU ops per value, which made the
// norm / epilogue phases of the small kernels instruction-issue bound.
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ float round_bf16(float f) { return (float)(__bf16)f; }

__device__ __forceinline__ bf16x8 as_bf16x8(u32x4 v) {
  union { u32x4 u; bf16x8 b; } x;
  x.u = v;
  return x.b;
}

__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// Sum over the 16 lanes of a DPP row, every lane receiving the SAME bits: the xor butterfly (1, 2, 4, 8) on the VALU's own cross-lane path.  `__shfl_xor` compiles
// to ds_bpermute_b32 + `s_waitcnt lgkmcnt(0)` per step (four LDS round trips in a row); quad_perm swaps lanes 1 and 2 apart, and once the quads (halves) hold
// uniform values the 8-lane (16-lane) mirror pairs every lane with a lane of the partner quad (half) -- the same operands as xor 4 (xor 8).
template <int CTRL>
__device__ __forceinline__ float vl_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float group16_sum(float v) {
  v += vl_dpp<0xB1>(v);      // quad_perm [1,0,3,2]
  v += vl_dpp<0x4E>(v);      // quad_perm [2,3,0,1]
  v += vl_dpp<0x141>(v);     // row_half_mirror
  v += vl_dpp<0x140>(v);     // row_mirror
  return v;
}

// xor-16 / xor-32 butterfly steps without the LDS crossbar: v_permlane16_swap (v_permlane32_swap) exchanges the odd 16-lane rows (the upper half) of its first operand
// with the even rows (the lower half) of its second; with both operands = v the two results hold {own row pair's even row, odd row} in every lane.
// COMPILER TRAP (hipcc / ROCm 7.2, found on the GPU in r05): a float add / max of the two results of the builtin is folded to `op(result0, result0)`
// (`v_permlane16_swap v1, v2; v_add_f32 v1, v1, v1`) -- every reduction through the first version of these helpers was wrong.  Storing the results separately is
// compiled correctly; passing them through an empty asm statement restores `v_add_f32 v1, v1, v2`.
struct vl_pair { float a, b; };
__device__ __forceinline__ vl_pair vl_swap16(float v) {
  const uint32_t x = __builtin_bit_cast(uint32_t, v);
  const u32x2 s = __builtin_amdgcn_permlane16_swap(x, x, false, false);
  uint32_t s0 = s[0], s1 = s[1];
  asm volatile("" : "+v"(s0), "+v"(s1));
  return {__builtin_bit_cast(float, s0), __builtin_bit_cast(float, s1)};
}
__device__ __forceinline__ vl_pair vl_swap32(float v) {
  const uint32_t x = __builtin_bit_cast(uint32_t, v);
  const u32x2 s = __builtin_amdgcn_permlane32_swap(x, x, false, false);
  uint32_t s0 = s[0], s1 = s[1];
  asm volatile("" : "+v"(s0), "+v"(s1));
  return {__builtin_bit_cast(float, s0), __builtin_bit_cast(float, s1)};
}
__device__ __forceinline__ float vl_xor16_sum(float v) { const vl_pair p = vl_swap16(v); return p.a + p.b; }
__device__ __forceinline__ float vl_xor32_sum(float v) { const vl_pair p = vl_swap32(v); return p.a + p.b; }
__device__ __forceinline__ float vl_xor16_max(float v) { const vl_pair p = vl_swap16(v); return fmaxf(p.a, p.b); }
__device__ __forceinline__ float vl_xor32_max(float v) { const vl_pair p = vl_swap32(v); return fmaxf(p.a, p.b); }
// Whole-wave butterfly reductions (xor 32, 16, 8, 4, 2, 1 -- every lane receives the same bits) on the VALU's cross-lane paths.  r01-r04 spelled them with
// `__shfl_xor`, which hipcc compiles to ds_bpermute_b32 + `s_waitcnt lgkmcnt(0)` per step: six dependent LDS-crossbar round trips (~0.3 us) in every norm / seam /
// argmax kernel.  Same operand pairs, hence identical results: permlane swaps for the two row-crossing steps, row_ror:8 = xor 8 inside a 16-lane row, row_ror:4 =
// xor 4 once the values are 8-periodic, quad_perm for xor 2 and xor 1.
__device__ __forceinline__ float wave_sum(float v) {
  v = vl_xor32_sum(v);
  v = vl_xor16_sum(v);
  v += vl_dpp<0x128>(v);
  v += vl_dpp<0x124>(v);
  v += vl_dpp<0x4E>(v);
  v += vl_dpp<0xB1>(v);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
  v = vl_xor32_max(v);
  v = vl_xor16_max(v);
  v = fmaxf(v, vl_dpp<0x128>(v));
  v = fmaxf(v, vl_dpp<0x124>(v));
  v = fmaxf(v, vl_dpp<0x4E>(v));
  v = fmaxf(v, vl_dpp<0xB1>(v));
  return v;
}
// sum over the 16 lanes that share lane & 3 (the 16 blocks of a 4x4x4 MFMA), every lane receiving the same bits: rotate by 8 (= xor 8 inside a row), by 4 (= xor 4 once the
// values are 8-periodic), then the rows
__device__ __forceinline__ float vl_blocks16_sum(float v) {
  v += vl_dpp<0x128>(v);     // row_ror:8
  v += vl_dpp<0x124>(v);     // row_ror:4
  v = vl_xor16_sum(v);
  return vl_xor32_sum(v);
}

// erf by Abramowitz & Stegun 7.1.26 (|err| <= 1.5e-7, far below bf16 resolution): ~12 VALU ops instead of the ~60 of the
// device library's erff, which made the GELU epilogue of the ViT fc1 GEMM cost as much as half its K loop.
__device__ __forceinline__ float fast_erf(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float r = 1.0f - poly * __expf(-ax * ax);
  return copysignf(r, x);
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + fast_erf(x * 0.70710678118654752f)); }
__device__ __forceinline__ float silu(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }   // rcp (1 ulp) instead of an IEEE division (~10 ops)

__device__ __forceinline__ u32x4 ld_global_16(const void* p) { return *reinterpret_cast<const u32x4*>(p); }
__device__ __forceinline__ void st_global_16(void* p, u32x4 v) { *reinterpret_cast<u32x4*>(p) = v; }

// A block-uniform int32 read through the SCALAR cache (s_load_dword, lgkmcnt): `valid_len[b]` and the like.  As a vector load it sits in the vmcnt queue, and
// hipcc waits for it with vmcnt(0) at its first use -- in attn_skinny that drained Q's round trip before the first K / V^T request went out (r05, read in the ISA:
// two serialised memory round trips in front of the first MFMA).  The constant address space promises that nobody writes the word during the launch.
typedef const __attribute__((address_space(4))) int32_t* vl_cptr_i32;
__device__ __forceinline__ int vl_sload_i32(const int32_t* p) { return *reinterpret_cast<vl_cptr_i32>(reinterpret_cast<uintptr_t>(p)); }

// ---- kernel arguments in ONE scalar-memory round trip.  hipcc issues the s_load of a by-value argument struct piecemeal, where each field is first
// needed, with an `s_waitcnt lgkmcnt(0)` in front of every first use: the <= 16-row kernels started with 3-6 SERIALISED round trips to the kernarg segment
// (scalar cache cold at every kernel start) before their first global load went out (r04, read in the ISA: skinny_kernel 4 rounds, the since-removed attention + o_proj kernel 6).
// Touching every 16-byte piece of the struct in one empty asm statement at the top of the kernel makes the compiler fetch all of it up front, back to back,
// behind a single wait.  MEASURED (r04f, same-box A/B of the whole chunk): SLOWER, 13.75 vs 12.99 ms -- inside a replayed HIP graph a kernarg line costs only
// 40-80 ns (r04 lab, profiles/r04i_attn_oproj_timeline.md: the graph keeps its arguments in device memory; eager launches with host-resident kernargs pay 1.2 us per
// line), the burst fetches all 5 lines of a 264-byte struct where the piecemeal code touches what the variant needs and overlaps the rest with address
// arithmetic.  Kept behind -DVL_KERNARG_UP_FRONT for the record; OFF.
template <int NQ, class P>
__device__ __forceinline__ void vl_kernargs_touch(const P& p) {      // the first NQ 16-byte pieces (+ the 0-3 trailing dwords when NQ covers the struct)
  constexpr int N4 = sizeof(P) / 16, REM = (sizeof(P) % 16) / 4;
  const u32x4* q = reinterpret_cast<const u32x4*>(&p);
  const uint32_t* w = reinterpret_cast<const uint32_t*>(&p) + N4 * 4;
#define VL_KA(i) "s"(q[(i) < N4 ? (i) : N4 - 1])
  if constexpr (NQ <= 8)
    asm volatile("" ::VL_KA(0), VL_KA(1), VL_KA(2), VL_KA(3), VL_KA(4), VL_KA(5), VL_KA(6), VL_KA(7), "s"(w[REM > 0 ? 0 : -1]), "s"(w[REM > 1 ? 1 : -1]), "s"(w[REM > 2 ? 2 : -1]));
  else if constexpr (NQ <= 12)
    asm volatile("" ::VL_KA(0), VL_KA(1), VL_KA(2), VL_KA(3), VL_KA(4), VL_KA(5), VL_KA(6), VL_KA(7), VL_KA(8), VL_KA(9), VL_KA(10), VL_KA(11),
                 "s"(w[REM > 0 ? 0 : -1]), "s"(w[REM > 1 ? 1 : -1]), "s"(w[REM > 2 ? 2 : -1]));
  else if constexpr (NQ <= 14)
    asm volatile("" ::VL_KA(0), VL_KA(1), VL_KA(2), VL_KA(3), VL_KA(4), VL_KA(5), VL_KA(6), VL_KA(7), VL_KA(8), VL_KA(9), VL_KA(10), VL_KA(11), VL_KA(12), VL_KA(13),
                 "s"(w[REM > 0 ? 0 : -1]), "s"(w[REM > 1 ? 1 : -1]), "s"(w[REM > 2 ? 2 : -1]));
  else if constexpr (NQ <= 17)
    asm volatile("" ::VL_KA(0), VL_KA(1), VL_KA(2), VL_KA(3), VL_KA(4), VL_KA(5), VL_KA(6), VL_KA(7), VL_KA(8), VL_KA(9), VL_KA(10), VL_KA(11), VL_KA(12), VL_KA(13),
                 VL_KA(14), VL_KA(15), VL_KA(16), "s"(w[REM > 0 ? 0 : -1]), "s"(w[REM > 1 ? 1 : -1]), "s"(w[REM > 2 ? 2 : -1]));
  else
    asm volatile("" ::VL_KA(0), VL_KA(1), VL_KA(2), VL_KA(3), VL_KA(4), VL_KA(5), VL_KA(6), VL_KA(7), VL_KA(8), VL_KA(9), VL_KA(10), VL_KA(11), VL_KA(12), VL_KA(13),
                 VL_KA(14), VL_KA(15), VL_KA(16), VL_KA(17), VL_KA(18), VL_KA(19), VL_KA(20), VL_KA(21), VL_KA(22), "s"(w[REM > 0 ? 0 : -1]), "s"(w[REM > 1 ? 1 : -1]),
                 "s"(w[REM > 2 ? 2 : -1]));
#undef VL_KA
}
template <class P>
__device__ __forceinline__ void vl_kernargs_up_front(const P& p) {
  static_assert(sizeof(P) % 4 == 0 && sizeof(P) <= 23 * 16 + 12, "argument struct: whole dwords, <= 380 bytes");
  vl_kernargs_touch<sizeof(P) / 16>(p);
}

// ---- split-K slab reduction: all loads of one call are independent and in flight together (one L2 round trip)
// sum of the first S (<= NB) fp32 slabs into v; always issues NB independent load pairs (index clamped) -> one trip
template <int NB>
__device__ __forceinline__ void add_slabs_clamped(float (&v)[8], const float* base, size_t slab, int S) {
  f32x4 q[2 * NB];
#pragma unroll
  for (int u = 0; u < NB; ++u) {
    const float* pp = base + (size_t)min(u, S - 1) * slab;
    q[2 * u] = *reinterpret_cast<const f32x4*>(pp);
    q[2 * u + 1] = *reinterpret_cast<const f32x4*>(pp + 4);
  }
#pragma unroll
  for (int u = 0; u < NB; ++u) {
    const bool on = u < S;
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] += on ? q[2 * u][j] : 0.f; v[4 + j] += on ? q[2 * u + 1][j] : 0.f; }
  }
}


// host-side error plumbing (api.cpp)
void vlaser_set_error(const char* fmt, ...);
#define VL_CHECK(cond, ...)            \
  do {                                 \
    if (!(cond)) {                     \
      vlaser_set_error(__VA_ARGS__);   \
      return -1;                       \
    }                                  \
  } while (0)
#define VL_HIP(expr)                                                        \
  do {                                                                      \
    hipError_t _e = (expr);                                                 \
    if (_e != hipSuccess) {                                                 \
      vlaser_set_error("%s failed: %s", #expr, hipGetErrorString(_e));      \
      return -2;                                                            \
    }                                                                       \
  } while (0)
#define VL_LAUNCH_CHECK() VL_HIP(hipGetLastError())

// one attribute call per (kernel, device): the C ABI promises thread safety w.r.t. distinct streams, and a process may drive
// several GPUs
template <class K>
static int set_max_lds_once(K kernel, int lds) {
  // keyed by the kernel's ADDRESS: every instantiation of a kernel template has the same function-pointer type, so a table per
  // type would let the first (largest) instantiation mask all the others
  static std::mutex mu;
  static std::map<std::pair<const void*, int>, int> done;
  int dev = 0;
  VL_HIP(hipGetDevice(&dev));
  const void* fn = reinterpret_cast<const void*>(kernel);
  std::lock_guard<std::mutex> g(mu);
  int& have = done[{fn, dev}];
  if (have < lds) {
    VL_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    have = lds;
  }
  return 0;
}
