// Lab (VERDICT r04 #2c: "the one experiment never run"): does the MFMA SHAPE matter for the GEMM K loop of this path?
// One 512-thread workgroup per CU holds the 192x256 tile's operands of ONE 64-deep K-step in LDS (A 24 KB + W 32 KB, never refilled: no global traffic, no LDS-DMA) and
// walks `steps` K-steps over them exactly as gemm_glds_kernel<.., 192, 256, 2, 4, ..> does -- per wave a 96x64 tile, all fragment reads of a half step, then its MFMAs:
//   shape 16: v_mfma_f32_16x16x32_bf16, per K-step 2 x (6 A + 4 W fragment reads of 16 bytes per lane) and 48 MFMAs of 8 passes
//   shape 32: v_mfma_f32_32x32x16_bf16, per K-step 4 x (3 A + 2 W fragment reads)                    and 24 MFMAs of 16 passes
// (same bytes out of LDS, same accumulator count, half the MFMA issue slots), each as: both | reads only | MFMAs only.  What it prices: the loop's issue structure at the
// tile shape that bounds the SFT step's gate/up GEMMs -- with the barrier per K-step the product kernel has, and without.
//   hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -mllvm -amdgpu-mfma-vgpr-form tools/micro/mfma_shape_lab.hip -o tools/micro/mfma_shape_lab && tools/micro/mfma_shape_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdint>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

__device__ __forceinline__ int lds_off(int row, int slot) { return row * 128 + ((slot ^ ((row >> 1) & 7)) << 4); }      // conflict-free for 16 consecutive rows at one k-slot

// DMA (second table): on top of the 16x16 loop every wave also issues the 7 LDS-DMA pieces of a refill per K-step (1 KiB each, from an L2-resident buffer into a spare LDS region,
// waited for at the top of the next step like the product's two-stage ring), as one burst in front of the second half's reads:
//   1 = the product's form: per piece a 64-bit per-lane pointer + the K offset (2 VALU), s_mov m0 + s_nop, global_load_lds_dwordx4 v[ptr], off
//   2 = scalar base + 32-bit per-lane offset, ONE m0 write per 4 pieces, the piece's LDS / global displacement in the instruction's immediate offset
//       (global_load_lds_dwordx4 v_off, s[base:base+1] offset:1024 i; the immediate moves BOTH addresses -- probed below -- so v_off carries -1024 i)
// FLY: refills (of 7 pieces) that may still be in flight at the top of a step; CONTIG: a piece = 1 KiB contiguous in memory (pre-tiled operand) instead of 8 rows x 128 B
__device__ int g_random_fill = 0;      // r06: 1 = full-range random bf16 operands in [-1, 1) instead of the near-constant pattern (DVFS: guide 5.4 rule 25)
__device__ __forceinline__ uint32_t lab_rnd_pair(uint32_t i) {
  uint32_t h = i * 2654435761u ^ 0x9e3779b9u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
  const float a = ((h & 0xffff) / 32768.0f - 1.0f), b = ((h >> 16) / 32768.0f - 1.0f);
  return (__float_as_uint(a) >> 16) | (__float_as_uint(b) & 0xffff0000u);
}
template <int SHAPE, bool READS, bool MFMAS, bool BARRIER, int DMA, int FLY, bool CONTIG, bool SHARED>
__global__ __launch_bounds__(512) void lab_kernel(int steps, float* sink, const char* gsrc = nullptr) {
  __shared__ __attribute__((aligned(16))) char smem[(192 + 256) * 128 + (DMA ? 8 * 7 * 1024 : 0)];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wr = wave >> 2, wc = wave & 3;
  if (g_random_fill) {
    for (int i = tid; i < (192 + 256) * 8; i += 512) reinterpret_cast<u32x4*>(smem)[i] = u32x4{lab_rnd_pair(4 * i), lab_rnd_pair(4 * i + 1), lab_rnd_pair(4 * i + 2), lab_rnd_pair(4 * i + 3)};
  } else {
    for (int i = tid; i < (192 + 256) * 8; i += 512) reinterpret_cast<u32x4*>(smem)[i] = u32x4{0x3c003c00u + i, 0x3c003c00u, 0x3c013c00u, 0x3c003c02u};
  }
  __syncthreads();
  const char* As = smem;
  const char* Ws = smem + 192 * 128;
  float acc_sum = 0.f;
  if constexpr (SHAPE == 16) {
    const int fr = lane & 15, fq = lane >> 4;
    f32x4 acc[4][6];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    bf16x8 fa[6], fw[4];
#pragma unroll
    for (int t = 0; t < 6; ++t) fa[t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(As + lds_off(wr * 96 + t * 16 + fr, fq)));
#pragma unroll
    for (int t = 0; t < 4; ++t) fw[t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(Ws + lds_off(wc * 64 + t * 16 + fr, fq)));
    [[maybe_unused]] const uint32_t dma_lds = (uint32_t)(uintptr_t)smem + (192 + 256) * 128 + wave * 7 * 1024;
    // this wave's 7 x 8 "rows" of 4 KB.  SHARED = 1: the same 1.8 MB for every workgroup (L2-resident in every XCD, as the product's operand tiles are: a W tile is read by
    // the 3-5 workgroups of its tile column, an A tile by up to 70); SHARED = 0: 448 KB of its own per workgroup (115 MB touched: out of the Infinity Cache / HBM, 7 TB/s over the chip)
    [[maybe_unused]] const char* wsrc = gsrc + ((size_t)(SHARED ? 0 : blockIdx.x) * 8 + wave) * (7 * 8 * 4096);
    [[maybe_unused]] const char* pp[7];
    [[maybe_unused]] uint32_t voff[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) { pp[i] = CONTIG ? wsrc + (size_t)i * 8 * 4096 + lane * 16 : wsrc + (size_t)(i * 8 + (lane >> 3)) * 4096 + (lane & 7) * 16; voff[i] = (uint32_t)((i * 8 + (lane >> 3)) * 4096 + (lane & 7) * 16) - (uint32_t)(i & 3) * 1024u; }
    for (int s = 0; s < steps; ++s) {
      if constexpr (DMA != 0) { if constexpr (FLY == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else if constexpr (FLY == 1) asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); }
      if constexpr (BARRIER) __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        if constexpr (DMA == 1) {
          if (ks == 1) {
            const size_t ko = (size_t)(s & 7) * (CONTIG ? 1024 : 128);
#pragma unroll
            for (int i = 0; i < 7; ++i) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(pp[i] + ko), "s"(__builtin_amdgcn_readfirstlane(dma_lds + i * 1024)) : "memory");
          }
        }
        if constexpr (DMA == 2) {
          if (ks == 1) {
            const char* sb = wsrc + (size_t)(s & 7) * 128;
            asm volatile("s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %5\n\tglobal_load_lds_dwordx4 %1, %5 offset:1024\n\tglobal_load_lds_dwordx4 %2, %5 offset:2048\n\tglobal_load_lds_dwordx4 %3, %5 offset:3072"
                         : : "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "v"(voff[3]), "s"(__builtin_amdgcn_readfirstlane(dma_lds)), "s"(sb) : "memory");
            asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %4\n\tglobal_load_lds_dwordx4 %1, %4 offset:1024\n\tglobal_load_lds_dwordx4 %2, %4 offset:2048"
                         : : "v"(voff[4]), "v"(voff[5]), "v"(voff[6]), "s"(__builtin_amdgcn_readfirstlane(dma_lds + 4096)), "s"(sb) : "memory");
          }
        }
        if constexpr (READS) {
#pragma unroll
          for (int t = 0; t < 6; ++t) fa[t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const volatile u32x4*>(As + lds_off(wr * 96 + t * 16 + fr, ks * 4 + fq)));
#pragma unroll
          for (int t = 0; t < 4; ++t) fw[t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const volatile u32x4*>(Ws + lds_off(wc * 64 + t * 16 + fr, ks * 4 + fq)));
        }
        if constexpr (MFMAS) {
#pragma unroll
          for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int mt = 0; mt < 6; ++mt) acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[nt], fa[mt], acc[nt][mt], 0, 0, 0);
        } else {
#pragma unroll
          for (int t = 0; t < 6; ++t) asm volatile("" ::"v"(fa[t]));
#pragma unroll
          for (int t = 0; t < 4; ++t) asm volatile("" ::"v"(fw[t]));
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j) acc_sum += acc[i][j][0] + acc[i][j][3];
  } else {
    const int r32 = lane & 31, kh = lane >> 5;
    f32x16 acc[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    bf16x8 fa[3], fw[2];
#pragma unroll
    for (int t = 0; t < 3; ++t) fa[t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(As + lds_off(wr * 96 + t * 32 + r32, kh)));
#pragma unroll
    for (int t = 0; t < 2; ++t) fw[t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(Ws + lds_off(wc * 64 + t * 32 + r32, kh)));
    for (int s = 0; s < steps; ++s) {
      if constexpr (BARRIER) __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        if constexpr (READS) {
#pragma unroll
          for (int t = 0; t < 3; ++t) fa[t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const volatile u32x4*>(As + lds_off(wr * 96 + t * 32 + r32, kk * 2 + kh)));
#pragma unroll
          for (int t = 0; t < 2; ++t) fw[t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const volatile u32x4*>(Ws + lds_off(wc * 64 + t * 32 + r32, kk * 2 + kh)));
        }
        if constexpr (MFMAS) {
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int mt = 0; mt < 3; ++mt) acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[nt], fa[mt], acc[nt][mt], 0, 0, 0);
        } else {
#pragma unroll
          for (int t = 0; t < 3; ++t) asm volatile("" ::"v"(fa[t]));
#pragma unroll
          for (int t = 0; t < 2; ++t) asm volatile("" ::"v"(fw[t]));
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) acc_sum += acc[i][j][0] + acc[i][j][15];
  }
  if constexpr (DMA != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc_sum == 12345.678f) sink[blockIdx.x] = acc_sum;
}

// probe: does the immediate offset of global_load_lds move the LDS address, the global address, or both?  src[i] = i (dwords); one wave
__global__ void probe_kernel(const uint32_t* src, uint32_t* out) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[2048];
  for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = 0xdeadbeefu;
  __syncthreads();
  const uint32_t voff = threadIdx.x * 16, base = (uint32_t)(uintptr_t)lds;
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024\n\ts_waitcnt vmcnt(0)" : : "v"(voff), "s"(src), "s"(base) : "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 2048; i += 64) out[i] = lds[i];
}

template <int SHAPE, bool READS, bool MFMAS, bool BARRIER, int DMA = 0, int FLY = 0, bool CONTIG = false, bool SHARED = false>
static void run(const char* what, float* sink, const char* gsrc = nullptr) {
  const int steps = 4800;       // = 200 launches' worth of the 24-step K loop
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((lab_kernel<SHAPE, READS, MFMAS, BARRIER, DMA, FLY, CONTIG, SHARED>), dim3(256), dim3(512), 0, 0, 240, sink, gsrc);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((lab_kernel<SHAPE, READS, MFMAS, BARRIER, DMA, FLY, CONTIG, SHARED>), dim3(256), dim3(512), 0, 0, steps, sink, gsrc);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double us_step = ms * 1e3 / steps, tflops = 256.0 * 192 * 256 * 64 * 2 / (us_step * 1e-6) / 1e12;
  printf("| %dx%d | %s | %s | %.3f | %s |\n", SHAPE, SHAPE, what, BARRIER ? "yes" : "no", us_step, MFMAS ? (char*)([&] { static char b[32]; snprintf(b, 32, "%.0f", tflops); return b; }()) : "");
}

int main(int argc, char** argv) {
  float* sink;
  hipMalloc((void**)&sink, 256 * 4);
  const bool random_fill = argc > 1 && !strcmp(argv[1], "random");
  if (random_fill) {
    int one = 1;
    hipMemcpyToSymbol(HIP_SYMBOL(g_random_fill), &one, sizeof(int));
    printf("operands: full-range random bf16 in [-1, 1) (accumulators grow like a random walk: no overflow in 4 800 steps)\n\n");
  } else printf("operands: the near-constant pattern of r05 (1.0 +- a few ulp)\n\n");
  printf("| MFMA shape | per K-step (192x256x64 tile, 8 waves) | barrier per K-step | us per K-step | TFLOP/s (256 CUs) |\n|---|---|---|---|---|\n");
  run<16, true, true, true>("fragment reads + MFMAs (the product loop's structure)", sink);
  run<32, true, true, true>("fragment reads + MFMAs", sink);
  run<16, true, true, false>("fragment reads + MFMAs", sink);
  run<32, true, true, false>("fragment reads + MFMAs", sink);
  run<16, false, true, false>("MFMAs only", sink);
  run<32, false, true, false>("MFMAs only", sink);
  run<16, true, false, false>("fragment reads only", sink);
  run<32, true, false, false>("fragment reads only", sink);
  if (random_fill) return 0;
  // the immediate offset's meaning
  uint32_t *psrc, *pout;
  hipMalloc((void**)&psrc, 8192 * 4); hipMalloc((void**)&pout, 2048 * 4);
  { uint32_t h[8192]; for (int i = 0; i < 8192; ++i) h[i] = i; hipMemcpy(psrc, h, sizeof(h), hipMemcpyHostToDevice); }
  hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, 0, psrc, pout);
  { uint32_t h[2048]; hipMemcpy(h, pout, sizeof(h), hipMemcpyDeviceToHost);
    int first = -1; for (int i = 0; i < 2048; ++i) if (h[i] != 0xdeadbeefu) { first = i; break; }
    printf("\nprobe: global_load_lds_dwordx4 v_off, s[src] offset:1024 with m0 = LDS base: first written LDS dword %d (256 = the offset moves the LDS address), its value %u (256 = it moves the global address too)\n", first, first >= 0 ? h[first] : 0u); }
  // piece issue inside the loop
  char* gsrc;
  hipMalloc((void**)&gsrc, (size_t)256 * 8 * 7 * 8 * 4096 + 4096);
  hipMemset(gsrc, 1, (size_t)256 * 8 * 7 * 8 * 4096 + 4096);
  printf("\n| 16x16 loop with its barrier + 7 LDS-DMA pieces per wave and K-step | us per K-step |\n|---|---|\n");
  run<16, true, true, true, 0>("no pieces", sink, gsrc);
  run<16, true, true, true, 1>("form 1: 64-bit per-lane pointers, m0 per piece (the product's)", sink, gsrc);
  run<16, true, true, true, 2>("form 2: scalar base + 32-bit offsets, m0 per 4 pieces, immediate offsets", sink, gsrc);
  run<16, true, true, true, 1, 1>("form 1, one refill may fly over the barrier (1.5 steps to land)", sink, gsrc);
  run<16, true, true, true, 2, 1>("form 2, one refill may fly", sink, gsrc);
  run<16, true, true, true, 1, 2>("form 1, two refills may fly (2.5 steps to land)", sink, gsrc);
  run<16, true, true, true, 2, 2>("form 2, two refills may fly", sink, gsrc);
  run<16, true, true, true, 1, 0, true>("form 1, every piece 1 KiB CONTIGUOUS in memory (pre-tiled operand)", sink, gsrc);
  run<16, true, true, true, 1, 1, true>("form 1, contiguous pieces, one refill may fly", sink, gsrc);
  run<16, false, false, true, 1, 0, false>("pieces only (no reads, no MFMAs), 8 rows x 128 B", sink, gsrc);
  run<16, false, false, true, 1, 0, true>("pieces only, contiguous", sink, gsrc);
  run<16, false, false, true, 1, 2, false>("pieces only, 8 rows x 128 B, two refills may fly", sink, gsrc);
  printf("\n| the same with an L2-RESIDENT source (every workgroup reads the same 1.8 MB) | | | us per K-step | |\n|---|---|---|---|---|\n");
  run<16, false, false, true, 1, 0, false, true>("pieces only, 8 rows x 128 B", sink, gsrc);
  run<16, false, false, true, 1, 2, false, true>("pieces only, two refills may fly", sink, gsrc);
  run<16, false, false, true, 1, 0, true, true>("pieces only, contiguous", sink, gsrc);
  run<16, true, true, true, 1, 0, false, true>("loop + pieces (form 1)", sink, gsrc);
  run<16, true, true, true, 1, 1, false, true>("loop + pieces, one refill may fly", sink, gsrc);
  run<16, true, true, true, 1, 2, false, true>("loop + pieces, two refills may fly", sink, gsrc);
  run<16, true, true, true, 2, 2, false, true>("loop + pieces (form 2), two refills may fly", sink, gsrc);
  return 0;
}
