// Micro-benchmark: v_mfma_f64_16x16x4_f64 issue interval on MI355X -- dependent accumulation chains (C <- A B + C) per wavefront,
// 1 / 2 / 4 wavefronts per SIMD (what bounds k_ilqr_backward_mfma).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int CH> __global__ __launch_bounds__(1024) void k_mfma(double* out, int iters) {
  d4 acc[CH];
#pragma unroll
  for (int i = 0; i < CH; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
  const double a = 1.0 + 1e-9 * threadIdx.x, b = 1e-3;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int i = 0; i < CH; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
  }
  double r = 0;
#pragma unroll
  for (int i = 0; i < CH; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int CH> void run(int waves_per_simd) {
  double* out;
  (void)hipMalloc(&out, sizeof(double) * 256 * 1024);
  const int iters = 2000, blocks = 256, threads = 256 * waves_per_simd;    // one workgroup per CU, waves_per_simd wavefronts on each SIMD
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k_mfma<CH>), dim3(blocks), dim3(threads), 0, 0, out, 10);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k_mfma<CH>), dim3(blocks), dim3(threads), 0, 0, out, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double n = (double)iters * 8 * CH;                 // MFMAs per wavefront
  const double per_simd_ns = ms * 1e6 / (n * waves_per_simd);
  printf("chains %d, %d wavefront(s) per SIMD: %.3f ms, %.1f ns per MFMA per wavefront, %.1f ns per MFMA per SIMD, %.1f TFLOP/s\n", CH, waves_per_simd, ms,
         ms * 1e6 / n, per_simd_ns, 2048.0 * n * waves_per_simd * 1024 / (ms * 1e-3) / 1e12);
  (void)hipFree(out);
}

int main() {
  for (int w : {1, 2, 4}) { run<1>(w); run<2>(w); run<4>(w); }
  return 0;
}
