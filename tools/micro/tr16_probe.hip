// Probe of ds_read_b64_tr_b16 (gfx950 transposing LDS read): LDS holds lds[i] = i; each lane passes an element address
// and prints the 4 values it receives, for a few address patterns.  hipcc --offload-arch=gfx950 tr16_probe.hip -o tr16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void k(uint16_t* out, const int* addr) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + addr[threadIdx.x]));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (uint16_t)v[j];
}
int main() {
  int* d_addr; uint16_t* d_out;
  hipMalloc(&d_addr, 64 * 4); hipMalloc(&d_out, 256 * 2);
  auto run = [&](const char* name, auto f) {
    std::vector<int> a(64); for (int l = 0; l < 64; ++l) a[l] = f(l);
    hipMemcpy(d_addr, a.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_out, d_addr);
    std::vector<uint16_t> o(256); hipMemcpy(o.data(), d_out, 512, hipMemcpyDeviceToHost);
    printf("== %s\n", name);
    for (int l = 0; l < 64; ++l) { printf("l%2d a=%4d: %4d %4d %4d %4d%s", l, a[l], o[l*4], o[l*4+1], o[l*4+2], o[l*4+3], (l % 4 == 3) ? "\n" : " | "); }
  };
  run("addr = lane*4 (consecutive 8-byte pieces)", [](int l) { return l * 4; });
  run("row-major [k][n] tile, ld=64: addr = (l>>4)*4*64*0 + ... lane i of group g -> row (i>>2)?", [](int l) { return (l & 15) * 64 + (l >> 4) * 4; });
  run("addr = (l&3)*64 + (l>>2)*4  (4 rows x 16 col-quads)", [](int l) { return (l & 3) * 64 + ((l >> 2) & 3) * 4 + (l >> 4) * 16; });
  run("addr = 0 for all", [](int) { return 0; });
  return 0;
}
