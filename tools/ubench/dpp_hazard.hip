// Is the "VALU writes a VGPR -> next instruction reads it through DPP" hazard interlocked on gfx950, or must software
// insert wait states (LLVM puts `s_nop 1` there)?  Dependent chains of 64-bit DPP instructions WITHOUT nops, verified
// bit for bit against a host emulation.  (tools/ubench/lane_comm.hip measured the nop at ~3 ns per link.)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

template <int NOP> __global__ void k_chain(double* io, int steps) {
  double a = io[threadIdx.x];
  const double m = 0.75;
  for (int s = 0; s < steps; ++s) {
    // 1: fmac reading the previous fmac's result through DPP, no nop
    if (NOP) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(m));
    else asm volatile("v_fmac_f64_dpp %0, %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(m));
    // 2: plain VALU producer immediately followed by a DPP mov of its result, then consumer
    double p, q;
    if (NOP) asm volatile("v_mul_f64 %0, %2, %3\n\ts_nop 1\n\tv_mov_b64_dpp %1, %0 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "=&v"(p), "=v"(q) : "v"(a), "v"(m));
    else asm volatile("v_mul_f64 %0, %2, %3\n\tv_mov_b64_dpp %1, %0 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "=&v"(p), "=v"(q) : "v"(a), "v"(m));
    a = q + 1.0;
  }
  io[threadIdx.x] = a;
}

int main() {
  const int steps = 200;
  std::vector<double> h(64), ref(64);
  for (int i = 0; i < 64; ++i) h[i] = ref[i] = 1.0 + 0.001 * i;
  for (int s = 0; s < steps; ++s) {
    double t[64];
    for (int i = 0; i < 64; ++i) t[i] = std::fma(ref[(i & ~15) | 3], 0.75, ref[i]);
    double p[64];
    for (int i = 0; i < 64; ++i) p[i] = t[i] * 0.75;
    for (int i = 0; i < 64; ++i) ref[i] = p[(i & ~15) | 5] + 1.0;
  }
  double* d; (void)hipMalloc(&d, 64 * sizeof(double));
  for (int nop = 0; nop < 2; ++nop) {
  int bad = 0;
  for (int rep = 0; rep < 50; ++rep) {
    (void)hipMemcpy(d, h.data(), 64 * sizeof(double), hipMemcpyHostToDevice);
    if (nop) hipLaunchKernelGGL(k_chain<1>, dim3(1), dim3(64), 0, 0, d, steps);
    else hipLaunchKernelGGL(k_chain<0>, dim3(1), dim3(64), 0, 0, d, steps);
    std::vector<double> o(64);
    (void)hipMemcpy(o.data(), d, 64 * sizeof(double), hipMemcpyDeviceToHost);
    for (int i = 0; i < 64; ++i) if (std::memcmp(&o[i], &ref[i], 8)) ++bad;
    if (rep == 0) printf("lane 0: device %.17g host %.17g\n", o[0], ref[0]);
  }
  printf("DPP read-after-write %s: %d mismatching lanes over 50 runs\n", nop ? "with s_nop 1" : "without wait states", bad);
  }
  return 0;
}
