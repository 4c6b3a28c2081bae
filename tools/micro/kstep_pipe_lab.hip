// Lab (r06): what a K-step of the LDS-DMA GEMM costs ON RANDOM DATA, and whether software-pipelining the fragment reads removes the serial part.
// r05's mfma_shape_lab priced the 192x256 loop over LDS-resident operands at 0.90 us per K-step -- on a near-constant operand pattern.  On full-range random bf16 the same
// loop takes 1.10 us (clock: guide 5.4 rule 25), MFMAs alone 0.81, fragment reads alone 0.28: in the product's structure [barrier | reads h0 | MFMAs h0 | reads h1 | MFMAs h1]
// the two waves of a SIMD read together and compute together, so reads and MFMAs ADD (0.81 + 0.28).  Variants here, all with one barrier per K-step, operands resident in LDS:
//   0  the product's order
//   1  pipelined: [barrier | reads(kt, h0) -> set A | MFMAs(kt-1, h1) from set B | reads(kt, h1) -> set B | MFMAs(kt, h0) from set A] -- every read block is issued in front of
//      an MFMA block that does not depend on it; same accumulation order per accumulator (bit-identical in the product)
//   2  variant 1 without scheduling fences (the compiler may interleave reads and MFMAs)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form tools/micro/kstep_pipe_lab.hip -o tools/micro/kstep_pipe_lab && tools/micro/kstep_pipe_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

__device__ __forceinline__ int lds_off(int row, int slot) { return row * 128 + ((slot ^ (row & 7)) << 4); }
__device__ __forceinline__ uint32_t rnd_pair(uint32_t i, int random) {
  if (!random) return 0x3c003c00u + (i & 3);
  uint32_t h = i * 2654435761u ^ 0x9e3779b9u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
  const float a = ((h & 0xffff) / 32768.0f - 1.0f), b = ((h >> 16) / 32768.0f - 1.0f);
  return (__float_as_uint(a) >> 16) | (__float_as_uint(b) & 0xffff0000u);
}

template <int BM, int VAR>
__global__ __launch_bounds__(512) void lab_kernel(int steps, float* sink, int random) {
  constexpr int BNT = 256, WTM = BM / 2, MT = WTM / 16, NT = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wr = wave >> 2, wc = wave & 3, fr = lane & 15, fq = lane >> 4;
  for (int i = tid; i < (BM + BNT) * 8; i += 512)
    reinterpret_cast<u32x4*>(smem)[i] = u32x4{rnd_pair(4 * i, random), rnd_pair(4 * i + 1, random), rnd_pair(4 * i + 2, random), rnd_pair(4 * i + 3, random)};
  __syncthreads();
  const char* As = smem;
  const char* Ws = smem + BM * 128;
  f32x4 acc[NT][MT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
  auto reads = [&](bf16x8 (&fa)[MT], bf16x8 (&fw)[NT], int ks) {
#pragma unroll
    for (int t = 0; t < MT; ++t) fa[t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const volatile u32x4*>(As + lds_off(wr * WTM + t * 16 + fr, ks * 4 + fq)));
#pragma unroll
    for (int t = 0; t < NT; ++t) fw[t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const volatile u32x4*>(Ws + lds_off(wc * 64 + t * 16 + fr, ks * 4 + fq)));
  };
  auto mfmas = [&](bf16x8 (&fa)[MT], bf16x8 (&fw)[NT]) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[nt], fa[mt], acc[nt][mt], 0, 0, 0);
  };
  bf16x8 faA[MT], fwA[NT], faB[MT], fwB[NT];
  if constexpr (VAR == 0) {
    for (int s = 0; s < steps; ++s) {
      __builtin_amdgcn_s_barrier();
      reads(faA, fwA, 0);
      mfmas(faA, fwA);
      reads(faA, fwA, 1);
      mfmas(faA, fwA);
    }
  } else {
    reads(faB, fwB, 1);
    for (int s = 0; s < steps; ++s) {
      __builtin_amdgcn_s_barrier();
      if constexpr (VAR == 1) __builtin_amdgcn_sched_barrier(0);
      reads(faA, fwA, 0);
      if constexpr (VAR == 1) __builtin_amdgcn_sched_barrier(0);
      mfmas(faB, fwB);
      if constexpr (VAR == 1) __builtin_amdgcn_sched_barrier(0);
      reads(faB, fwB, 1);
      if constexpr (VAR == 1) __builtin_amdgcn_sched_barrier(0);
      mfmas(faA, fwA);
      if constexpr (VAR == 1) __builtin_amdgcn_sched_barrier(0);
    }
    mfmas(faB, fwB);
  }
  float acc_sum = 0.f;
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc_sum += acc[i][j][0] + acc[i][j][3];
  if (acc_sum == 12345.678f) sink[blockIdx.x] = acc_sum;
}

// `kstep_pipe_lab power`: the resident loop (no memory traffic of its own) beside a second stream that streams HBM -- does the K-step stretch although the two share no unit?
__global__ __launch_bounds__(256) void stream_kernel(const u32x4* __restrict__ src, size_t n16, int rounds, float* sink) {
  u32x4 acc = {0, 0, 0, 0};
  for (int r = 0; r < rounds; ++r)
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
      const u32x4 v = __builtin_nontemporal_load(src + i);
      acc[0] ^= v[0]; acc[1] ^= v[1]; acc[2] ^= v[2]; acc[3] ^= v[3];
    }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[blockIdx.x] = 1.f;
}

template <int BM, int VAR>
static void run(const char* what, float* sink, int random) {
  const int steps = 4800, lds = (BM + 256) * 128;
  hipFuncSetAttribute((const void*)lab_kernel<BM, VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((lab_kernel<BM, VAR>), dim3(256), dim3(512), lds, 0, 240, sink, random);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((lab_kernel<BM, VAR>), dim3(256), dim3(512), lds, 0, steps, sink, random);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double us_step = ms * 1e3 / steps;
  printf("| %dx256 | %s | %s | %.3f | %.0f |\n", BM, random ? "random" : "near-constant", what, us_step, 256.0 * BM * 256 * 64 * 2 / (us_step * 1e-6) / 1e12);
  fflush(stdout);
}

int main(int argc, char** argv) {
  float* sink;
  hipMalloc((void**)&sink, 256 * 4);
  if (argc > 1 && !strcmp(argv[1], "power")) {
    const size_t bytes = (size_t)2 << 30;
    u32x4* buf; hipMalloc((void**)&buf, bytes); hipMemset(buf, 1, bytes);
    hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
    const int steps = 4800, lds = (192 + 256) * 128;
    hipFuncSetAttribute((const void*)lab_kernel<192, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    printf("| 192x256 resident K loop (product order, random operands, no memory traffic of its own) beside ... | us per K-step | background GB/s |\n|---|---|---|\n");
    for (int wgs : {0, 32, 64, 128, 256, 512, 0}) {
      hipEvent_t e0, e1, b0, b1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&b0); hipEventCreate(&b1);
      hipLaunchKernelGGL((lab_kernel<192, 0>), dim3(256), dim3(512), lds, s1, 240, sink, 1);
      hipDeviceSynchronize();
      const int rounds = 12;
      if (wgs) { hipEventRecord(b0, s2); hipLaunchKernelGGL(stream_kernel, dim3(wgs), dim3(256), 0, s2, buf, bytes / 16, rounds, sink); hipEventRecord(b1, s2); }
      hipEventRecord(e0, s1);
      hipLaunchKernelGGL((lab_kernel<192, 0>), dim3(256), dim3(512), lds, s1, steps, sink, 1);
      hipEventRecord(e1, s1);
      hipDeviceSynchronize();
      float ms = 0, bms = 0; hipEventElapsedTime(&ms, e0, e1); if (wgs) hipEventElapsedTime(&bms, b0, b1);
      char w[96]; snprintf(w, sizeof w, wgs ? "%d streaming workgroups (nt loads over 2 GiB)" : "nothing", wgs);
      printf("| %s | %.3f | %s |\n", w, ms * 1e3 / steps, wgs ? ([&] { static char b[32]; snprintf(b, 32, "%.0f (stream ran %.1f ms, loop %.1f ms)", (double)bytes * rounds / (bms * 1e-3) / 1e9, bms, ms); return b; }()) : "");
      fflush(stdout);
    }
    return 0;
  }
  printf("| tile | operands | K-step structure (one barrier per step, operands resident in LDS) | us per K-step | TFLOP/s (256 CUs) |\n|---|---|---|---|---|\n");
  for (int random = 1; random >= 0; --random) {
    run<192, 0>("product order: reads h0, MFMAs h0, reads h1, MFMAs h1", sink, random);
    run<192, 1>("pipelined: reads(kt,h0), MFMAs(kt-1,h1), reads(kt,h1), MFMAs(kt,h0) -- fenced", sink, random);
    run<192, 2>("pipelined, compiler-scheduled", sink, random);
    run<128, 0>("product order", sink, random);
    run<128, 1>("pipelined, fenced", sink, random);
    run<128, 2>("pipelined, compiler-scheduled", sink, random);
    run<256, 0>("product order", sink, random);
    run<256, 1>("pipelined, fenced", sink, random);
    run<256, 2>("pipelined, compiler-scheduled", sink, random);
  }
  return 0;
}
