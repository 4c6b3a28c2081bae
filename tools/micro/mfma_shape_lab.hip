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
#include <cstdint>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

__device__ __forceinline__ int lds_off(int row, int slot) { return row * 128 + ((slot ^ ((row >> 1) & 7)) << 4); }      // conflict-free for 16 consecutive rows at one k-slot

template <int SHAPE, bool READS, bool MFMAS, bool BARRIER>
__global__ __launch_bounds__(512) void lab_kernel(int steps, float* sink) {
  __shared__ __attribute__((aligned(16))) char smem[(192 + 256) * 128];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 2, wc = wave & 3;
  for (int i = tid; i < (192 + 256) * 8; i += 512) reinterpret_cast<u32x4*>(smem)[i] = u32x4{0x3c003c00u + i, 0x3c003c00u, 0x3c013c00u, 0x3c003c02u};
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
    for (int s = 0; s < steps; ++s) {
      if constexpr (BARRIER) __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
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
  if (acc_sum == 12345.678f) sink[blockIdx.x] = acc_sum;
}

template <int SHAPE, bool READS, bool MFMAS, bool BARRIER>
static void run(const char* what, float* sink) {
  const int steps = 4800;       // = 200 launches' worth of the 24-step K loop
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((lab_kernel<SHAPE, READS, MFMAS, BARRIER>), dim3(256), dim3(512), 0, 0, 240, sink);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((lab_kernel<SHAPE, READS, MFMAS, BARRIER>), dim3(256), dim3(512), 0, 0, steps, sink);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double us_step = ms * 1e3 / steps, tflops = 256.0 * 192 * 256 * 64 * 2 / (us_step * 1e-6) / 1e12;
  printf("| %dx%d | %s | %s | %.3f | %s |\n", SHAPE, SHAPE, what, BARRIER ? "yes" : "no", us_step, MFMAS ? (char*)([&] { static char b[32]; snprintf(b, 32, "%.0f", tflops); return b; }()) : "");
}

int main() {
  float* sink;
  hipMalloc((void**)&sink, 256 * 4);
  printf("| MFMA shape | per K-step (192x256x64 tile, 8 waves) | barrier per K-step | us per K-step | TFLOP/s (256 CUs) |\n|---|---|---|---|---|\n");
  run<16, true, true, true>("fragment reads + MFMAs (the product loop's structure)", sink);
  run<32, true, true, true>("fragment reads + MFMAs", sink);
  run<16, true, true, false>("fragment reads + MFMAs", sink);
  run<32, true, true, false>("fragment reads + MFMAs", sink);
  run<16, false, true, false>("MFMAs only", sink);
  run<32, false, true, false>("MFMAs only", sink);
  run<16, true, false, false>("fragment reads only", sink);
  run<32, true, false, false>("fragment reads only", sink);
  return 0;
}
