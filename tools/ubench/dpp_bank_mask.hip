// Half-row (8-lane) groups on 64-bit DPP: does `bank_mask` work with the DP forms (v_mov_b64_dpp / v_fmac_f64_dpp
// row_newbcast)?  A bank is 4 lanes of a row: bank_mask 0x3 = lanes 0..7, 0xC = lanes 8..15.  Two problems per row then
// broadcast "lane L of my half" with two instructions:  row_newbcast:L bank_mask:0x3 ; row_newbcast:L+8 bank_mask:0xC.
// Prints what every lane of the first row receives.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k(double* out) {
  const double v = 100.0 + threadIdx.x;
  double mv = -1.0, fm = 0.5, both = -1.0, fboth = 0.5;
  const double two = 2.0;
  asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0x3" : "+v"(mv) : "v"(v));
  asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xc" : "+v"(fm) : "v"(v), "v"(two));
  asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0x3\n\t"
               "v_mov_b64_dpp %0, %1 row_newbcast:11 row_mask:0xf bank_mask:0xc" : "+v"(both) : "v"(v));
  asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0x3\n\t"
               "v_fmac_f64_dpp %0, %1, %2 row_newbcast:11 row_mask:0xf bank_mask:0xc" : "+v"(fboth) : "v"(v), "v"(two));
  out[threadIdx.x] = mv; out[64 + threadIdx.x] = fm; out[128 + threadIdx.x] = both; out[192 + threadIdx.x] = fboth;
}

int main() {
  double* d; (void)hipMalloc(&d, 256 * sizeof(double));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  std::vector<double> h(256);
  (void)hipMemcpy(h.data(), d, 256 * sizeof(double), hipMemcpyDeviceToHost);
  const char* names[4] = {"mov   bcast:3  bank 0x3 (old -1)", "fmac  bcast:3  bank 0xc (old .5)", "mov   bcast:3|11 half rows     ", "fmac  bcast:3|11 half rows     "};
  int ok = 1;
  for (int t = 0; t < 4; ++t) {
    printf("%s:", names[t]);
    for (int l = 16; l < 32; ++l) printf(" %g", h[64 * t + l]);
    printf("\n");
  }
  for (int l = 0; l < 64; ++l) {
    const int row = l & ~15, hi = (l & 8) != 0;
    ok &= h[l] == (hi ? -1.0 : 100.0 + row + 3);
    ok &= h[64 + l] == (hi ? 0.5 + 2.0 * (100.0 + row + 3) : 0.5);
    ok &= h[128 + l] == 100.0 + row + (hi ? 11 : 3);
    ok &= h[192 + l] == 0.5 + 2.0 * (100.0 + row + (hi ? 11 : 3));
  }
  printf("bank_mask on 64-bit DPP: %s\n", ok ? "works as documented (half-row broadcasts in two instructions)" : "NOT as expected");
  return 0;
}
