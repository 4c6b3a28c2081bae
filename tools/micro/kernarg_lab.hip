// Kernel-argument fetch latency on MI355X (r04): how long after its start does a wave have (a) the first 64-byte line, (b) a later line of its
// kernarg segment, eager and inside a replayed HIP graph behind a producer kernel; and the same for a dependent load from a DEVICE buffer.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/kernarg_lab.hip -o tools/micro/kernarg_lab && tools/micro/kernarg_lab
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Big { unsigned long long* out; const uint32_t* dev; uint32_t pad[60]; };      // 256 bytes = 4 lines

__global__ void filler(float* x) { x[threadIdx.x + blockIdx.x * 64] += 1.0f; }

__global__ __launch_bounds__(64) void probe(Big b) {
  const unsigned long long t0 = wall_clock64();
  uint32_t a0, a3, d0;
  const uint64_t kp = (uint64_t)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("s_load_dword %0, %1, 0x10\n\ts_waitcnt lgkmcnt(0)" : "=s"(a0) : "s"(kp) : "memory");       // line 0 (holds out / dev too)
  const unsigned long long t1 = wall_clock64();
  asm volatile("s_load_dword %0, %1, 0xf0\n\ts_waitcnt lgkmcnt(0)" : "=s"(a3) : "s"(kp) : "memory");       // line 3
  const unsigned long long t2 = wall_clock64();
  const uint64_t dp = (uint64_t)b.dev;
  asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(d0) : "s"(dp) : "memory");        // device buffer, scalar path
  const unsigned long long t3 = wall_clock64();
  const uint32_t dv = b.dev[64 + threadIdx.x];                                                             // device buffer, vector path
  asm volatile("s_waitcnt vmcnt(0)" ::"v"(dv) : "memory");
  const unsigned long long t4 = wall_clock64();
  if (threadIdx.x == 0) {
    unsigned long long* o = b.out + blockIdx.x * 8;
    o[0] = t0; o[1] = t1; o[2] = t2; o[3] = t3; o[4] = t4; o[5] = a0 + a3 + d0 + dv;
  }
}

int main() {
  const int NB = 256, REP = 200;
  unsigned long long* out; uint32_t* dev; float* x;
  CK(hipMalloc(&out, NB * 8 * 8 * REP)); CK(hipMalloc(&dev, 4096)); CK(hipMalloc(&x, 64 * 256 * 4));
  CK(hipMemset(dev, 0, 4096)); CK(hipMemset(x, 0, 64 * 256 * 4));
  hipStream_t s; CK(hipStreamCreate(&s));
  std::vector<unsigned long long> h(NB * 8 * REP);
  auto report = [&](const char* name) {
    CK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> d1, d2, d3, d4;
    for (int r = REP / 2; r < REP; ++r)
      for (int b = 0; b < NB; ++b) {
        const unsigned long long* o = &h[(size_t)(r * NB + b) * 8];
        d1.push_back((o[1] - o[0]) * 10.0); d2.push_back((o[2] - o[1]) * 10.0); d3.push_back((o[3] - o[2]) * 10.0); d4.push_back((o[4] - o[3]) * 10.0);
      }
    auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    auto p90 = [](std::vector<double>& v) { return v[v.size() * 9 / 10]; };
    printf("%-28s kernarg line 0: %5.0f ns (p90 %5.0f) | line 3: %5.0f (%5.0f) | device buffer s_load: %5.0f (%5.0f) | device buffer vector load: %5.0f (%5.0f)\n", name,
           med(d1), p90(d1), med(d2), p90(d2), med(d3), p90(d3), med(d4), p90(d4));
    return 0;
  };
  // eager
  for (int r = 0; r < REP; ++r) {
    Big b{}; b.out = out + (size_t)r * NB * 8; b.dev = dev;
    hipLaunchKernelGGL(filler, dim3(256), dim3(64), 0, s, x);
    hipLaunchKernelGGL(probe, dim3(NB), dim3(64), 0, s, b);
  }
  CK(hipStreamSynchronize(s));
  if (report("eager")) return 1;
  // graph: REP (filler, probe) pairs captured once, replayed
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  for (int r = 0; r < REP; ++r) {
    Big b{}; b.out = out + (size_t)r * NB * 8; b.dev = dev;
    hipLaunchKernelGGL(filler, dim3(256), dim3(64), 0, s, x);
    hipLaunchKernelGGL(probe, dim3(NB), dim3(64), 0, s, b);
  }
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, s));
  CK(hipStreamSynchronize(s));
  if (report("graph replay")) return 1;
  return 0;
}
