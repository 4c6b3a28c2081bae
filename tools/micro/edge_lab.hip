// Microbenchmark for the persistent expert-stack kernel: cost of ONE in-launch all-to-all hand-off ("edge") on MI355X.
// 256 resident workgroups (one per CU) run NITER dependent edges; in each edge the first P workgroups publish `piece` bytes
// each (write-through sc1 stores -> vmcnt(0) -> barrier -> one sc1 flag store), every workgroup polls the P flags (one wave,
// relaxed agent loads) and then gathers `gather` bytes of the published payload with sc1 loads (no acquire fence), verifying
// every dword.  Optional background: every wave keeps `bg` 1-KiB non-temporal weight loads in flight per edge (in-order
// return per wave: shows what a gather queued behind a weight prefetch costs).
// Build: hipcc --offload-arch=gfx950 -O3 edge_lab.hip -o edge_lab ; run: ./edge_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(4))) unsigned int u4;
typedef __attribute__((address_space(1))) unsigned int gu32;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

__device__ __forceinline__ unsigned word_of(int e, unsigned idx) { return (unsigned)(e + 1) * 0x9E3779B1u + idx; }

template <int BG, bool GATHER_WAVE0>
__global__ __launch_bounds__(512) void edge_kernel(char* buf, unsigned* flags, unsigned* err, const u4* bgw, int P, int piece, int gather, int niter,
                                                   size_t layer_stride, unsigned* sink) {
  const int wg = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __shared__ unsigned s_abort;
  if (tid == 0) s_abort = 0;
  __syncthreads();
  unsigned acc = 0, bad = 0;
  const size_t total = (size_t)P * piece;
  for (int e = 0; e < niter; ++e) {
    char* pay = buf + (size_t)e * layer_stride;           // layer-indexed payload: written once per launch, no WAR hazards
    unsigned* fl = flags + (size_t)e * 256;
    u4 bgv[BG > 0 ? BG : 1];
    if constexpr (BG > 0) {
#pragma unroll
      for (int i = 0; i < BG; ++i) bgv[i] = __builtin_nontemporal_load(bgw + ((size_t)((e * 256 + wg) * 8 + wave) * BG + i) * 64 + lane);
    }
    if (wg < P) {
      auto rs = __builtin_amdgcn_make_buffer_rsrc(pay + (size_t)wg * piece, 0, piece, 0x00020000);
      for (int off = tid * 16; off < piece; off += 512 * 16) {
        const unsigned i0 = (unsigned)((size_t)wg * piece + off) >> 2;
        u4 v = {word_of(e, i0), word_of(e, i0 + 1), word_of(e, i0 + 2), word_of(e, i0 + 3)};
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 16);      // aux 16 = sc1 (write-through)
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (wg < P && tid == 0) __hip_atomic_store(fl + wg, 1u, RLX_AGENT);
    // ---- consume
    if (wave == 0) {
      unsigned spins = 0;
      for (;;) {
        bool ok = true;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int p = lane + 64 * k;
          const unsigned v = p < P ? __hip_atomic_load(fl + p, RLX_AGENT) : 1u;
          ok &= v != 0;
        }
        if (__all(ok)) break;
        if (++spins > 4000000u) { if (lane == 0) { s_abort = 1; atomicAdd(err + 1, 1u); } break; }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
    if (s_abort) break;
    {
      auto rs = __builtin_amdgcn_make_buffer_rsrc(pay, 0, (unsigned)total, 0x00020000);
      const int nthr = GATHER_WAVE0 ? 64 : 512;
      if (!GATHER_WAVE0 || wave == 0) {
        // each workgroup gathers `gather` bytes starting at a workgroup-dependent offset (wraps): the consumers of a real edge
        // read different slices
        const int start = (int)(((size_t)wg * 4096) % total) & ~15;
        for (int off = (GATHER_WAVE0 ? lane : tid) * 16; off < gather; off += nthr * 16) {
          int o = start + off; if (o >= (int)total) o -= (int)total;
          const u4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, o, 0, 16);   // sc1 load, no acquire fence
          const unsigned i0 = (unsigned)o >> 2;
          bad += (v.x != word_of(e, i0)) + (v.y != word_of(e, i0 + 1)) + (v.z != word_of(e, i0 + 2)) + (v.w != word_of(e, i0 + 3));
          acc ^= v.x;
        }
      }
    }
    if constexpr (BG > 0) {
#pragma unroll
      for (int i = 0; i < BG; ++i) acc ^= bgv[i].x ^ bgv[i].w;
    }
    __syncthreads();
  }
  if (bad) atomicAdd(err, bad);
  if (acc == 0x1234567u) sink[0] = acc;
}

template <int BG, bool G0>
static void run(const char* name, char* buf, unsigned* flags, unsigned* err, const u4* bgw, int P, int piece, int gather, int niter, size_t stride,
                unsigned* sink, hipStream_t s) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f; unsigned herr[2] = {0, 0};
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipMemsetAsync(flags, 0, (size_t)niter * 256 * 4, s));
    CK(hipMemsetAsync(err, 0, 8, s));
    CK(hipEventRecord(e0, s));
    hipLaunchKernelGGL((edge_kernel<BG, G0>), dim3(256), dim3(512), 0, s, buf, flags, err, bgw, P, piece, gather, niter, stride, sink);
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep > 0 && ms < best) best = ms;
    unsigned h[2]; CK(hipMemcpy(h, err, 8, hipMemcpyDeviceToHost)); herr[0] += h[0]; herr[1] += h[1];
  }
  printf("%-34s P=%3d piece=%5d gather=%6d bg=%d : %.2f us/edge  (stale words %u, timeouts %u)\n", name, P, piece, gather, BG, best * 1e3f / niter, herr[0], herr[1]);
}

int main() {
  hipStream_t s; CK(hipStreamCreate(&s));
  const int niter = 140;
  const size_t stride = 1 << 20;
  char* buf; CK(hipMalloc(&buf, stride * niter)); CK(hipMemset(buf, 0, stride * niter));
  unsigned *flags, *err, *sink; CK(hipMalloc(&flags, (size_t)niter * 256 * 4)); CK(hipMalloc(&err, 8)); CK(hipMalloc(&sink, 4));
  const size_t bgbytes = (size_t)niter * 256 * 8 * 8 * 1024;
  u4* bgw; CK(hipMalloc(&bgw, bgbytes)); CK(hipMemset(bgw, 1, bgbytes));
  // (producers, bytes per producer, bytes gathered per consumer)
  run<0, false>("h0: 48 x 128 B -> 6 KB", buf, flags, err, bgw, 48, 128, 6144, niter, stride, sink, s);
  run<0, true>("h0, wave 0 gathers", buf, flags, err, bgw, 48, 128, 6144, niter, stride, sink, s);
  run<0, false>("attn out: 2 x 6 KB -> 12 KB", buf, flags, err, bgw, 2, 6144, 12288, niter, stride, sink, s);
  run<0, false>("act: 254 x 280 B -> 14 KB", buf, flags, err, bgw, 254, 288, 14336, niter, stride, sink, s);
  run<0, false>("slab: 4 x 256 B -> 1 KB", buf, flags, err, bgw, 4, 256, 1024, niter, stride, sink, s);
  run<0, false>("5 fp32 slabs: 240 x 256 -> 60 KB", buf, flags, err, bgw, 240, 256, 61440, niter, stride, sink, s);
  run<0, false>("256 x 128 B -> 32 KB", buf, flags, err, bgw, 256, 128, 32768, niter, stride, sink, s);
  run<0, false>("flags only: 256 x 16 B -> 16 B", buf, flags, err, bgw, 256, 16, 16, niter, stride, sink, s);
  run<4, false>("h0 + 4 KB/wave weight loads", buf, flags, err, bgw, 48, 128, 6144, niter, stride, sink, s);
  run<4, false>("act + 4 KB/wave weight loads", buf, flags, err, bgw, 254, 288, 14336, niter, stride, sink, s);
  run<8, false>("act + 8 KB/wave weight loads", buf, flags, err, bgw, 254, 288, 14336, niter, stride, sink, s);
  run<4, true>("h0 (wave 0) + 4 KB/wave weights", buf, flags, err, bgw, 48, 128, 6144, niter, stride, sink, s);
  return 0;
}
