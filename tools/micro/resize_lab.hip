// Lab (r06): where the horizontal resampling pass (csrc/image.hip) spends its time on the 12-megapixel frame -> 1792 columns: the product kernel with its stores / its
// staging loads / its tap loop switched off one at a time.   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/resize_lab.hip -o tools/micro/resize_lab
#define VL_RS_LAB 1
#include "../../vlaser_amd/csrc/image.hip"
#include <cstdio>
#include <vector>
static thread_local char g_err[512];
void vlaser_set_error(const char* fmt, ...) { (void)fmt; }
int main() {
  const int H = 3024, W = 4032, w = 1792;
  const int ks = vlaser_resample_ksize(W, w);
  std::vector<int> bounds(2 * w), kk((size_t)ks * w);
  vlaser_resample_coeffs(W, w, bounds.data(), kk.data());
  uint8_t *src, *dst; int *db, *dk;
  hipMalloc((void**)&src, (size_t)H * W * 3); hipMalloc((void**)&dst, (size_t)H * w * 3); hipMalloc((void**)&db, bounds.size() * 4); hipMalloc((void**)&dk, kk.size() * 4);
  hipMemset(src, 77, (size_t)H * W * 3);
  hipMemcpy(db, bounds.data(), bounds.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dk, kk.data(), kk.size() * 4, hipMemcpyHostToDevice);
  printf("| variant of the horizontal pass, 4032x3024 -> 1792x3024 | us per launch |\n|---|---|\n");
  const int labs[] = {0, 1, 2, 4, 8, 11};
  const char* names[] = {"product", "no stores", "no staging loads", "one tap instead of 11", "no weight loads (constant weights)", "constant weights, no stores, no staging"};
  for (int li = 0; li < 6; ++li) {
    const int lab = labs[li];
    ResampleP p;
    p.src = src; p.dst = dst; p.bounds = db; p.kk = dk; p.ld_in = (long long)W * 3; p.ld_out = (long long)w * 3; p.out_n = w; p.ksize = ks; p.rows = H; p.row_bytes = w * 3; p.xb = 256; p.lab = lab;
    const double fs = (double)W / w;
    const long long npx = (long long)(256 * fs + ks + 2);
    const int lds = (int)((((npx * 3 + 26 + 15) & ~15LL) + npx * 4) * VL_RS_ROWS);
    hipFuncSetAttribute((const void*)resample_h_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(resample_h_kernel, dim3((w + 255) / 256, (H + VL_RS_ROWS - 1) / VL_RS_ROWS), dim3(256), lds, 0, p);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(resample_h_kernel, dim3((w + 255) / 256, (H + VL_RS_ROWS - 1) / VL_RS_ROWS), dim3(256), lds, 0, p);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("| %s | %.1f |\n", names[li], ms * 1e3 / 20);
  }
  return 0;
}
