// Lab: take the HBM-cold operand OUT of LDS.  The product's NT tile (gemm_glds_kernel) stages both operands through an LDS ring whose depth the 160 KB cap
// limits (192x256: A 2 + W 3 stages); its K-step runs at 1.35 us against 0.90 us for the same loop over resident operands (profiles/r05r_mfma_shape_lab.md).  Here the W
// fragments go STRAIGHT INTO VGPRs -- `global_load_dwordx4` from a fragment-major copy of W (every wave-level load = one contiguous 1 KiB = one 16-row x 32-k MFMA operand),
// D K-steps ahead in a register ring -- and all of LDS becomes a deeper ring for the L2-resident A operand alone.  The M-waves of a tile column fetch the same W fragments
// (L1 / L2 hits).  Same MFMA order per accumulator as the product (kt ascending, two 32-deep halves per step): outputs must be BIT-IDENTICAL to vlaser_gemm's.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form tools/micro/wreg_lab.hip -Iinclude -Lvlaser_amd/csrc -lvlaser_hip \
//         -Wl,-rpath,'$ORIGIN/../../vlaser_amd/csrc' -o tools/micro/wreg_lab && tools/micro/wreg_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "vlaser_hip.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef uint16_t bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ int lds_off(int row, int slot) { return row * 128 + ((slot ^ (row & 7)) << 4); }
// LDS-DMA piece: scalar base (the tile's K offset) + the lane's 32-bit byte offset (3 registers for 3 pieces instead of 3 pointer pairs + the 64-bit adds)
__device__ __forceinline__ void glds16(uint32_t voff, const char* sbase, uint32_t lds_dst) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2" : : "v"(voff), "s"(lds_dst), "s"(sbase) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// W fragment (16 rows x 32 k, lane-ordered, 1 KiB) straight into four VGPRs: scalar base (advanced per K-step) + the lane's 32-bit offset + immediate.  NOT counted by hipcc:
// the K loop waits with its own vmcnt arithmetic before the first MFMA that reads the registers (guide 5.7 form iii).
template <int IMM>
__device__ __forceinline__ void wload(bf16x8& dst, uint32_t voff, const char* sbase) {
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(sbase), "n"(IMM) : "memory");
}

struct LabP {
  const bf16_t* A; const char* Wp; bf16_t* out;
  int M, N, K, tiles_m, tiles_n;
};

// BM x 256 tile, WM x WN waves, NSTA A stages in LDS, W ring of D K-steps in registers (D <= NSTA - 1).  Fragment-major W: [N / 16][K / 64][2 halves][64 lanes][16 B].
// WM = 2, WN = 4 (the product's wave grid): the two M-waves of a tile column both fetch the column's W fragments -- 64 KB of W requests per K-step and CU for 32 KB of data.
// WM = 1, WN = 8 (second table): every wave owns ALL BM rows of 32 columns -- each W fragment is fetched by exactly one wave (32 KB per step), a wave's ring is 16 registers
// per K-step of depth, and every wave reads the whole A tile out of LDS (the same LDS read volume as 2 x 4: 8 waves x BM x 128 B).
template <int BM, int WM, int NSTA, int D, bool SPREAD, int WN = 4>
__global__ __launch_bounds__(WM * WN * 64) void wreg_kernel(LabP p) {
  constexpr int BNT = 256, NW = WM * WN, WTM = BM / WM, MT = WTM / 16, WTN = BNT / WN, NT = WTN / 16, NPA = BM / 8, PA = (NPA + NW - 1) / NW, GRP = PA + 2 * NT;
  static_assert(D <= NSTA - 1 && (D - 1) * GRP <= 63, "ring depths");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wr = wave / WN, wc = wave % WN, fr = lane & 15, fq = lane >> 4;
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_m = bid % p.tiles_m, tile_n = bid / p.tiles_m, m0 = tile_m * BM, n0 = tile_n * BNT, nk = p.K / 64;
  // A pieces of this wave (8 rows x 128 B each; XOR swizzle on the source side)
  uint32_t offA[PA];
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int row = min(wave * PA + i, NPA - 1) * 8 + (lane >> 3);
    offA[i] = (uint32_t)min(m0 + row, p.M - 1) * (uint32_t)(p.K * 2) + (((lane & 7) ^ (row & 7)) << 4);
  }
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  auto issue_a = [&](int j, int kt, int st) {
    const char* sa = reinterpret_cast<const char*>(p.A) + min(kt, nk - 1) * 128;
    glds16(offA[j], sa, __builtin_amdgcn_readfirstlane(lds0 + st * (BM * 128) + min(wave * PA + j, NPA - 1) * 1024));
  };
  // W streams of this wave: n-tiles n0 / 16 + wc * 4 + t
  uint32_t voff[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) voff[t] = (uint32_t)(min(n0 / 16 + wc * NT + t, p.N / 16 - 1)) * (uint32_t)(nk * 2048) + lane * 16;
  static_assert(PA <= 2 * NT, "the spread form hangs one A piece behind each MFMA group");
  bf16x8 w[D][2 * NT];
  f32x4 acc[NT][MT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
  // prologue: A tiles D .. NSTA-2 FIRST (older than every group: no wait ever has to let them fly), then the groups of K-steps 0 .. D-1 (A tile g + W step g)
#pragma unroll
  for (int g = D; g < NSTA - 1; ++g)
#pragma unroll
    for (int j = 0; j < PA; ++j) issue_a(j, g, g);
#pragma unroll
  for (int g = 0; g < D; ++g) {
#pragma unroll
    for (int j = 0; j < PA; ++j) issue_a(j, g, g);
    const char* sb = p.Wp + (size_t)min(g, nk - 1) * 2048;
#pragma unroll
    for (int t = 0; t < NT; ++t) { wload<0>(w[g][t], voff[t], sb); wload<1024>(w[g][NT + t], voff[t], sb); }
  }
  int st = 0;
  for (int kt0 = 0; kt0 < nk; kt0 += D) {
#pragma unroll
    for (int s = 0; s < D; ++s) {
      const int kt = kt0 + s;
      wait_vmcnt<(D - 1) * GRP>();          // everything issued for step kt (D groups ago) has landed; the D - 1 younger groups may fly
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      int stn = st + NSTA - 1; if (stn >= NSTA) stn -= NSTA;      // stage read in step kt - 1: refilled with tile kt + NSTA - 1
      const char* As = smem + st * (BM * 128);
      const char* sb = p.Wp + (size_t)min(kt + D, nk - 1) * 2048;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 fa[MT];
#pragma unroll
        for (int t = 0; t < MT; ++t) fa[t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(As + lds_off(wr * WTM + t * 16 + fr, ks * 4 + fq)));
        if constexpr (!SPREAD) {
          if (ks == 1) {
#pragma unroll
            for (int j = 0; j < PA; ++j) issue_a(j, kt + NSTA - 1, stn);
          }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[s][ks * NT + nt], fa[mt], acc[nt][mt], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (SPREAD) {                              // A pieces first (oldest), one per MFMA group; then the W fragments whose registers are free again
            const int g = ks * NT + nt;
            if (g < PA) issue_a(g, kt + NSTA - 1, stn);
          }
          // this fragment's registers are free: request the fragment of step kt + D.  (With !SPREAD the A pieces of this step were issued in front of ks = 1's MFMAs, i.e.
          // between W loads 3 and 4: the queue order differs, the COUNT per step does not, and every wait lets whole steps fly.)
          if (ks == 0) wload<0>(w[s][nt], voff[nt], sb); else wload<1024>(w[s][NT + nt], voff[nt], sb);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (++st == NSTA) st = 0;
    }
  }
  wait_vmcnt<0>();
  // fragment stores (8 bytes per lane: the product's mode-0 epilogue) -- not what the lab prices
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = m0 + wr * WTM + mt * 16 + fr;
    if (m < p.M) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int n = n0 + wc * WTN + nt * 16 + fq * 4;
        if (n < p.N) *reinterpret_cast<u32x2*>(p.out + (size_t)m * p.N + n) = u32x2{pack_bf16x2(acc[nt][mt][0], acc[nt][mt][1]), pack_bf16x2(acc[nt][mt][2], acc[nt][mt][3])};
      }
    }
  }
}

__global__ void pack_w(const bf16_t* W, bf16_t* Wp, int N, int K) {      // row-major [N][K] -> [N/16][K/64][2][64][8]
  const size_t total = (size_t)N * K / 8;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int lane = i & 63, ks = (i >> 6) & 1;
    const size_t r = i >> 7;
    const int nkk = K / 64, kt = r % nkk, nt = r / nkk;
    const bf16_t* src = W + (size_t)(nt * 16 + (lane & 15)) * K + kt * 64 + ks * 32 + (lane >> 4) * 8;
    *reinterpret_cast<u32x4*>(Wp + i * 8) = *reinterpret_cast<const u32x4*>(src);
  }
}
__global__ void fill_bf16(bf16_t* p, size_t n, unsigned seed, float scale) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = (bf16_t)(__float_as_uint(((h & 0xffff) / 32768.0f - 1.0f) * scale) >> 16);
  }
}
__global__ void count_diff(const bf16_t* a, const bf16_t* b, size_t n, unsigned* out) {
  unsigned c = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) c += a[i] != b[i];
  if (c) atomicAdd(out, c);
}

template <int BM, int WM, int NSTA, int D, bool SPREAD, int WN = 4>
static void launch_lab(const LabP& p0, hipStream_t s) {
  LabP p = p0;
  p.tiles_m = (p.M + BM - 1) / BM; p.tiles_n = (p.N + 255) / 256;
  constexpr int lds = NSTA * BM * 128;
  static bool once = false;
  if (!once) { CK(hipFuncSetAttribute((const void*)wreg_kernel<BM, WM, NSTA, D, SPREAD, WN>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); once = true; }
  hipLaunchKernelGGL((wreg_kernel<BM, WM, NSTA, D, SPREAD, WN>), dim3(p.tiles_m * p.tiles_n), dim3(WM * WN * 64), lds, s, p);
}

int main(int argc, char** argv) {
  hipStream_t s; CK(hipStreamCreate(&s));
  struct Shape { int M, N, K; const char* name; int cfg; };
  // cfg = the product's tile for the shape (force_bm): 1900 = 192x256 ASYM, 1300 = 256x256 ASYM, 1200 = 128x256 / 3 stages
  const bool warm_mode = argc > 1 && !strcmp(argv[1], "warm");
  // `wreg_lab pipe`: the product's tile against its PIPE variant (fragment reads pipelined across the K-step's barrier, csrc/gemm.hip) on the shapes that use the tile
  if (argc > 1 && !strcmp(argv[1], "pipe")) {
    struct PS { int M, N, K; const char* name; int base, pipe; };
    const PS ps[] = {{1025, 3072, 1024, "ViT qkv (128x128)", 1100, 1110}, {1025, 4096, 1024, "ViT fc1 (144x128)", 1442, 1440}, {384, 17920, 1536, "LLM prefill gate/up (128x256)", 1200, 1210},
                     {384, 2048, 1536, "LLM prefill qkv-sized (64x128)", 1502, 1500}, {384, 2048, 1536, "LLM prefill qkv-sized (64x64, 4 waves)", 1566, 1564},
                     {560, 17920, 1536, "SFT forward gate/up (192x256)", 1902, 1900}, {560, 2048, 1536, "SFT qkv (64x128)", 1502, 1500},
                     {560, 8960, 1536, "SFT 128x256", 1200, 1210}, {3408, 8192, 3584, "8B-sized (192x256)", 1902, 1900}, {3408, 8192, 3584, "8B-sized (128x256)", 1200, 1210},
                     {13 * 1025, 4096, 1024, "ViT fc1 x 13 tiles (128x256)", 1200, 1210},
                     {3408, 8192, 3584, "8B-sized (256x256 ASYM)", 1300, 1310}, {3408, 37888, 3584, "8B gate/up (256x256 ASYM)", 1300, 1310}, {3408, 3584, 18944, "8B down (256x256 ASYM)", 1300, 1310},
                     {13 * 1025, 4096, 1024, "ViT fc1 x 13 tiles (256x256 ASYM)", 1300, 1310}};
    printf("| shape | base us (code) | PIPE us (code) | delta | bit-identical |\n|---|---|---|---|---|\n");
    hipStream_t s; CK(hipStreamCreate(&s));
    unsigned* dcnt; CK(hipMalloc(&dcnt, 4));
    for (const PS& q : ps) {
      const int NL = 8;
      bf16_t *x, *w[NL], *o1, *o2;
      CK(hipMalloc(&x, (size_t)q.M * q.K * 2)); CK(hipMalloc(&o1, (size_t)q.M * q.N * 2)); CK(hipMalloc(&o2, (size_t)q.M * q.N * 2));
      fill_bf16<<<1024, 256, 0, s>>>(x, (size_t)q.M * q.K, 1, 1.0f);
      for (int i = 0; i < NL; ++i) { CK(hipMalloc(&w[i], (size_t)q.N * q.K * 2)); fill_bf16<<<1024, 256, 0, s>>>(w[i], (size_t)q.N * q.K, 100 + i, 0.03f); }
      double us[2][3];
      for (int rep = 0; rep < 3; ++rep)
        for (int v = 0; v < 2; ++v) {                          // interleaved: base, pipe, base, pipe, ...
          VlaserGemmArgs a; memset(&a, 0, sizeof a);
          a.A = x; a.M = q.M; a.N = q.N; a.K = q.K; a.lda = q.K; a.ldw = q.K; a.ldo = q.N; a.force_bm = v ? q.pipe : q.base; a.out = v ? o2 : o1;
          hipGraph_t g; hipGraphExec_t ge;
          CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
          for (int i = 0; i < NL; ++i) { a.W = w[i]; if (vlaser_gemm(VL_EPI_NONE, &a, s)) { printf("vlaser_gemm: %s\n", vlaser_last_error()); return 1; } }
          CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
          hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
          CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
          CK(hipEventRecord(e0, s));
          for (int r = 0; r < 10; ++r) CK(hipGraphLaunch(ge, s));
          CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1));
          us[v][rep] = ms * 1e3 / (10 * NL);
          CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        }
      CK(hipMemsetAsync(dcnt, 0, 4, s));
      count_diff<<<1024, 256, 0, s>>>(o1, o2, (size_t)q.M * q.N, dcnt);
      unsigned c; CK(hipMemcpyAsync(&c, dcnt, 4, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
      auto med = [](double* v) { double a = v[0], b = v[1], c3 = v[2]; return a + b + c3 - (a < b ? (a < c3 ? a : c3) : (b < c3 ? b : c3)) - (a > b ? (a > c3 ? a : c3) : (b > c3 ? b : c3)); };
      const double b = med(us[0]), pp = med(us[1]);
      printf("| %d x %d x %d %s | %.2f (%d) | %.2f (%d) | %+.1f %% | %s |\n", q.M, q.N, q.K, q.name, b, q.base, pp, q.pipe, 100.0 * (pp - b) / b, c ? "NO" : "yes");
      fflush(stdout);
      CK(hipFree(x)); CK(hipFree(o1)); CK(hipFree(o2));
      for (int i = 0; i < NL; ++i) CK(hipFree(w[i]));
    }
    return 0;
  }
  const Shape shapes_warm[] = {{1025, 3072, 1024, "ViT qkv", 0}, {1025, 1024, 1024, "ViT proj (whole K)", 0}, {1025, 4096, 1024, "ViT fc1", 0}, {1025, 1024, 4096, "ViT fc2 (whole K)", 0},
                               {384, 2048, 1536, "LLM qkv", 0}, {384, 1536, 1536, "LLM o_proj (whole K)", 0}, {384, 17920, 1536, "LLM gate/up", 0}, {384, 1536, 8960, "LLM down (whole K)", 0}};
  const Shape shapes[] = {{3408, 8192, 3584, "8B-sized (VERDICT r05 #1)", 0}, {3408, 8192, 3456, "the same, K = 54 steps (D = 3 divides)", 0}, {560, 17920, 1536, "SFT forward gate/up", 0},
                          {384, 17920, 1536, "chunk prefill gate/up", 0}, {1025, 4096, 1024, "ViT fc1", 0}};
  const int NL = 8;
  // `wreg_lab zero`: zero-filled operands (the DVFS uplift of guide 5.4 rule 25 on the same kernels: clock, not work); `wreg_lab <n>`: only the first n shapes
  const bool zero_fill = argc > 1 && !strcmp(argv[1], "zero");
  const float fs = zero_fill ? 0.f : 1.f;
  printf("operands: %s\n\n", zero_fill ? "ZERO-filled" : "full-range random bf16");
  unsigned* dcnt; CK(hipMalloc(&dcnt, 4));
  printf("| shape (M x N x K) | kernel | us per launch | TFLOP/s | us per K-step of the busiest CU | differs from the product in |\n|---|---|---|---|---|---|\n");
  int n_shapes = zero_fill ? 1 : 100;
  std::vector<Shape> run_shapes;
  if (warm_mode) run_shapes.assign(std::begin(shapes_warm), std::end(shapes_warm)); else run_shapes.assign(std::begin(shapes), std::end(shapes));
  if (warm_mode) printf("product launches with HBM-cold weights (8 buffers cycled) vs the SAME weight buffer every launch (L2 / Infinity-Cache warm)\n\n");
  for (const Shape& sh : run_shapes) {
    if (n_shapes-- <= 0) break;
    const int M = sh.M, N = sh.N, K = sh.K;
    bf16_t *x, *w[NL], *wp[NL], *out, *ref;
    CK(hipMalloc(&x, (size_t)M * K * 2)); CK(hipMalloc(&out, (size_t)M * N * 2)); CK(hipMalloc(&ref, (size_t)M * N * 2));
    fill_bf16<<<1024, 256, 0, s>>>(x, (size_t)M * K, 1, 1.0f * fs);
    for (int i = 0; i < NL; ++i) {
      CK(hipMalloc(&w[i], (size_t)N * K * 2)); CK(hipMalloc(&wp[i], (size_t)N * K * 2));
      fill_bf16<<<1024, 256, 0, s>>>(w[i], (size_t)N * K, 100 + i, 0.03f * fs);
      pack_w<<<2048, 256, 0, s>>>(w[i], wp[i], N, K);
    }
    const double fl = 2.0 * M * N * K;
    auto time_graph = [&](auto&& launch) {
      hipGraph_t g; hipGraphExec_t ge;
      CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      for (int i = 0; i < NL; ++i) launch(i);
      CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
      const int reps = 10;
      CK(hipEventRecord(e0, s));
      for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s));
      CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
      return ms * 1e3 / (reps * NL);
    };
    auto report = [&](const char* what, double us, int bm, const char* diff) {
      const int tiles = ((M + bm - 1) / bm) * ((N + 255) / 256), rounds = (tiles + 255) / 256;
      printf("| %d x %d x %d %s | %s | %.1f | %.0f | %.3f | %s |\n", M, N, K, sh.name, what, us, fl / (us * 1e-6) / 1e12, us / (rounds * (K / 64)), diff);
      fflush(stdout);
    };
    if (warm_mode) {
      VlaserGemmArgs a; memset(&a, 0, sizeof a);
      a.A = x; a.M = M; a.N = N; a.K = K; a.lda = K; a.ldw = K; a.ldo = N; a.force_bm = 0;
      auto cold = [&](int i) { a.W = w[i % NL]; a.out = out; if (vlaser_gemm(VL_EPI_NONE, &a, s)) { printf("vlaser_gemm: %s\n", vlaser_last_error()); exit(1); } };
      auto warm = [&](int i) { a.W = w[0]; a.out = out; if (vlaser_gemm(VL_EPI_NONE, &a, s)) { printf("vlaser_gemm: %s\n", vlaser_last_error()); exit(1); } };
      const double uc = time_graph(cold), uw = time_graph(warm);
      printf("| %d x %d x %d %s | cold %.2f us | warm %.2f us | -%.2f us (%.0f %%) |\n", M, N, K, sh.name, uc, uw, uc - uw, 100.0 * (uc - uw) / uc);
      fflush(stdout);
      CK(hipFree(x)); CK(hipFree(out)); CK(hipFree(ref));
      for (int i = 0; i < NL; ++i) { CK(hipFree(w[i])); CK(hipFree(wp[i])); }
      continue;
    }
    // the product (vlaser_gemm, its own tile choice and the forced 192x256 / 256x256 / 128x256 rings)
    for (int cfg : {0, 1902, 1300, 1200}) {
      VlaserGemmArgs a; memset(&a, 0, sizeof a);
      a.A = x; a.M = M; a.N = N; a.K = K; a.lda = K; a.ldw = K; a.ldo = N; a.force_bm = cfg;
      auto launch = [&](int i) { a.W = w[i % NL]; a.out = cfg == 1902 ? ref : out; if (vlaser_gemm(VL_EPI_NONE, &a, s)) { printf("vlaser_gemm: %s\n", vlaser_last_error()); exit(1); } };
      const double us = time_graph(launch);
      char nm[64]; snprintf(nm, sizeof nm, "product force_bm=%d", cfg);
      report(nm, us, cfg == 1300 ? 256 : cfg == 1200 ? 128 : 192, cfg == 1902 ? "(reference)" : "");
    }
    LabP p{x, nullptr, out, M, N, K, 0, 0};
    auto lab = [&](const char* what, int bm, auto&& fn, bool divides) {
      if (!divides) return;
      CK(hipMemsetAsync(out, 0, (size_t)M * N * 2, s));
      auto launch = [&](int i) { p.Wp = (const char*)wp[i % NL]; fn(p, s); };
      const double us = time_graph(launch);        // (the last launch of the graph used weight buffer NL - 1, like the product's reference run)
      CK(hipMemsetAsync(dcnt, 0, 4, s));
      count_diff<<<1024, 256, 0, s>>>(out, ref, (size_t)M * N, dcnt);
      unsigned c; CK(hipMemcpyAsync(&c, dcnt, 4, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
      char d[64]; snprintf(d, sizeof d, "%u of %zu elements", c, (size_t)M * N);
      report(what, us, bm, d);
    };
    const int nk = K / 64;
    lab("W in VGPRs: 192x256, A 4 stages, W ring 3, spread", 192, [](const LabP& q, hipStream_t st) { launch_lab<192, 2, 4, 3, true>(q, st); }, nk % 3 == 0);
    lab("W in VGPRs: 192x256, A 6 stages, W ring 3, spread", 192, [](const LabP& q, hipStream_t st) { launch_lab<192, 2, 6, 3, true>(q, st); }, nk % 3 == 0);
    lab("W in VGPRs: 192x256, A 6 stages, W ring 3, A burst", 192, [](const LabP& q, hipStream_t st) { launch_lab<192, 2, 6, 3, false>(q, st); }, nk % 3 == 0);
    lab("W in VGPRs: 192x256, A 3 stages, W ring 2, spread", 192, [](const LabP& q, hipStream_t st) { launch_lab<192, 2, 3, 2, true>(q, st); }, nk % 2 == 0);
    lab("W in VGPRs: 192x256, A 6 stages, W ring 2, spread", 192, [](const LabP& q, hipStream_t st) { launch_lab<192, 2, 6, 2, true>(q, st); }, nk % 2 == 0);
    lab("W in VGPRs: 128x256, A 5 stages, W ring 4, spread", 128, [](const LabP& q, hipStream_t st) { launch_lab<128, 2, 5, 4, true>(q, st); }, nk % 4 == 0);
    lab("W in VGPRs: 128x256, A 8 stages, W ring 3, spread", 128, [](const LabP& q, hipStream_t st) { launch_lab<128, 2, 8, 3, true>(q, st); }, nk % 3 == 0);
    lab("W in VGPRs: 128x256, A 8 stages, W ring 2, spread", 128, [](const LabP& q, hipStream_t st) { launch_lab<128, 2, 8, 2, true>(q, st); }, nk % 2 == 0);
    // one wave per 32 columns over ALL rows: no duplicate W fetch
    lab("W in VGPRs, 1 x 8 waves: 128x256, A 9 stages, W ring 4", 128, [](const LabP& q, hipStream_t st) { launch_lab<128, 1, 9, 4, true, 8>(q, st); }, nk % 4 == 0);
    lab("W in VGPRs, 1 x 8 waves: 128x256, A 9 stages, W ring 2", 128, [](const LabP& q, hipStream_t st) { launch_lab<128, 1, 9, 2, true, 8>(q, st); }, nk % 2 == 0);
    lab("W in VGPRs, 1 x 8 waves: 128x256, A 5 stages, W ring 4", 128, [](const LabP& q, hipStream_t st) { launch_lab<128, 1, 5, 4, true, 8>(q, st); }, nk % 4 == 0);
    lab("W in VGPRs, 1 x 8 waves: 128x256, A 9 stages, W ring 6", 128, [](const LabP& q, hipStream_t st) { launch_lab<128, 1, 9, 6, true, 8>(q, st); }, nk % 6 == 0);
    lab("W in VGPRs, 1 x 8 waves: 192x256, A 6 stages, W ring 4", 192, [](const LabP& q, hipStream_t st) { launch_lab<192, 1, 6, 4, true, 8>(q, st); }, nk % 4 == 0);
    lab("W in VGPRs, 1 x 8 waves: 192x256, A 6 stages, W ring 2", 192, [](const LabP& q, hipStream_t st) { launch_lab<192, 1, 6, 2, true, 8>(q, st); }, nk % 2 == 0);
    lab("W in VGPRs, 1 x 8 waves: 256x256, A 5 stages, W ring 2", 256, [](const LabP& q, hipStream_t st) { launch_lab<256, 1, 5, 2, true, 8>(q, st); }, nk % 2 == 0);
    lab("W in VGPRs, 1 x 8 waves: 256x256, A 4 stages, W ring 2", 256, [](const LabP& q, hipStream_t st) { launch_lab<256, 1, 4, 2, true, 8>(q, st); }, nk % 2 == 0);
    lab("W in VGPRs, 1 x 8 waves: 256x256, A 5 stages, W ring 3", 256, [](const LabP& q, hipStream_t st) { launch_lab<256, 1, 5, 3, true, 8>(q, st); }, nk % 3 == 0);
    lab("W in VGPRs, 1 x 8 waves: 64x256, A 9 stages, W ring 4", 64, [](const LabP& q, hipStream_t st) { launch_lab<64, 1, 9, 4, true, 8>(q, st); }, nk % 4 == 0);
    CK(hipFree(x)); CK(hipFree(out)); CK(hipFree(ref));
    for (int i = 0; i < NL; ++i) { CK(hipFree(w[i])); CK(hipFree(wp[i])); }
  }
  return 0;
}
