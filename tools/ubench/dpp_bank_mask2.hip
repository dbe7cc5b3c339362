#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(double* out) {
  const double v = 100.0 + threadIdx.x;
  const double two = 2.0;
  double a1 = 0.5, a2 = 0.5, a3 = 0.5, a3b = 0.5, a4 = 0.5, a5 = 0.5, a6 = 0.5;
  asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0x3\n\ts_nop 1\n\t"
               "v_fmac_f64_dpp %0, %1, %2 row_newbcast:11 row_mask:0xf bank_mask:0xc" : "+v"(a1) : "v"(v), "v"(two));
  asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0x3\n\ts_nop 7\n\t"
               "v_fmac_f64_dpp %0, %1, %2 row_newbcast:11 row_mask:0xf bank_mask:0xc" : "+v"(a2) : "v"(v), "v"(two));
  asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %2, %3 row_newbcast:3 row_mask:0xf bank_mask:0x3\n\t"
               "v_fmac_f64_dpp %1, %2, %3 row_newbcast:11 row_mask:0xf bank_mask:0xc" : "+v"(a3), "+v"(a3b) : "v"(v), "v"(two));
  asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0x3" : "+v"(a4) : "v"(v), "v"(two));
  asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
               "v_fmac_f64_dpp %0, %1, %2 row_newbcast:11 row_mask:0xf bank_mask:0xf" : "+v"(a5) : "v"(v), "v"(two));
  // bound_ctrl variant
  asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xc\n\t"
               "v_fmac_f64_dpp %0, %1, %2 row_newbcast:11 row_mask:0xf bank_mask:0x3" : "+v"(a6) : "v"(v), "v"(two));
  double* o = out + threadIdx.x;
  o[0] = a1; o[64] = a2; o[128] = a3; o[192] = a3b; o[256] = a4; o[320] = a5; o[384] = a6;
}
int main() {
  double* d; (void)hipMalloc(&d, 448 * sizeof(double));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  std::vector<double> h(448);
  (void)hipMemcpy(h.data(), d, 448 * sizeof(double), hipMemcpyDeviceToHost);
  const char* names[7] = {"same acc, nop1 between", "same acc, nop7 between", "two accs: first (0x3)", "two accs: second (0xc)", "single 0x3", "full masks 3 then 11", "same acc 0xc(3) then 0x3(11)"};
  for (int t = 0; t < 7; ++t) {
    printf("%-30s:", names[t]);
    for (int l = 16; l < 32; ++l) printf(" %g", h[64 * t + l]);
    printf("\n");
  }
  return 0;
}
