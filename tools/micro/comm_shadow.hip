// Lab helper (tools/micro/rccl_shadow_lab.py): a stand-in for RCCL's resident channel workgroups on ONE GPU -- C workgroups of 256 threads, each copying its slice
// of a buffer to another (device-local; on an 8-GPU node the peer would sit behind xGMI) `rounds` times.  What it answers: what a backward GEMM whose grid was
// sized for 256 free CUs costs while C CUs are held by a streaming kernel on another stream.  Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC comm_shadow.hip -o tools/micro/libcomm_shadow.so
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
__global__ __launch_bounds__(256) void comm_shadow_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16, int rounds, const int* stop, int mode) {
  if (mode == 1) {                // spin: hold the CU slot without touching memory beyond the flag (separates 'a second queue is busy' from 'CUs / HBM are taken')
    for (int r = 0; r < rounds * 4096; ++r) {
      if (stop && __hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
      __builtin_amdgcn_s_sleep(64);
    }
    return;
  }
  const size_t per = n16 / gridDim.x, lo = (size_t)blockIdx.x * per;
  for (int r = 0; r < rounds; ++r) {
    if (stop && __hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;      // the host raises the flag (stream-ordered memset) when the timed loop is over
    for (size_t i = threadIdx.x; i + 768 < per; i += 1024) {
      const u32x4 a = __builtin_nontemporal_load(src + lo + i), b = __builtin_nontemporal_load(src + lo + i + 256), c = __builtin_nontemporal_load(src + lo + i + 512),
                  d = __builtin_nontemporal_load(src + lo + i + 768);
      if (mode == 2) {              // read only: the loads must stay live
        if ((a.x ^ b.x ^ c.x ^ d.x) == 0x9e3779b9u) dst[lo + i] = a;
        continue;
      }
      __builtin_nontemporal_store(a, dst + lo + i); __builtin_nontemporal_store(b, dst + lo + i + 256); __builtin_nontemporal_store(c, dst + lo + i + 512);
      __builtin_nontemporal_store(d, dst + lo + i + 768);
    }
  }
}
extern "C" int comm_shadow(const void* src, void* dst, size_t bytes, int channels, int rounds, const int* stop, void* stream, int mode) {
  hipLaunchKernelGGL(comm_shadow_kernel, dim3(channels), dim3(256), 0, (hipStream_t)stream, (const u32x4*)src, (u32x4*)dst, bytes / 16, rounds, stop, mode);
  return (int)hipGetLastError();
}
// A stream whose kernels run only on CUs [first, first + n) of the 256-bit CU mask (KFD spreads consecutive mask bits round-robin over the 8 XCDs: the top 8 bits are one CU
// of each XCD).  The lab masks the COMPUTE stream (what a torch.distributed user can do: ProcessGroupNCCL owns RCCL's stream) and / or the shadow's.
extern "C" void* comm_shadow_masked_stream(int first, int n) {
  uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = first; i < first + n && i < 256; ++i) mask[i >> 5] |= 1u << (i & 31);
  hipStream_t s = nullptr;
  if (hipExtStreamCreateWithCUMask(&s, 8, mask) != hipSuccess) return nullptr;
  return (void*)s;
}
