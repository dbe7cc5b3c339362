// Micro-benchmark: issue interval of fp64 FMAs for one wavefront per SIMD as a function of the number of
// independent dependency chains (why the solve pass keeps the VALU busy only ~71 % of the time).
#include <hip/hip_runtime.h>
#include <cstdio>

template <int CH> __global__ __launch_bounds__(256, 1) void k_chain(double* out, int iters, int lanes) {
  if ((int)(threadIdx.x & 63) >= lanes) return;
  double a[CH];
#pragma unroll
  for (int i = 0; i < CH; ++i) a[i] = 1.0 + 1e-9 * threadIdx.x + i;
  const double m = 0.999999, c = 1e-7;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
      for (int i = 0; i < CH; ++i) a[i] = __builtin_fma(a[i], m, c);
    }
  }
  double r = 0;
#pragma unroll
  for (int i = 0; i < CH; ++i) r += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int CH> void run(int lanes) {
  double* out;
  (void)hipMalloc(&out, sizeof(double) * 1024 * 256);
  const int iters = 4000, blocks = 256;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k_chain<CH>), dim3(blocks), dim3(256), 0, 0, out, 10, lanes);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k_chain<CH>), dim3(blocks), dim3(256), 0, 0, out, iters, lanes);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double n = (double)iters * 16 * CH;
  printf("chains %2d lanes %2d: %.3f ms, %.2f ns per FMA per wavefront\n", CH, lanes, ms, ms * 1e6 / n);
  (void)hipFree(out);
}

// MI355X, 1024 wavefronts (one per SIMD): 1 chain 2.93 ns per FMA, 2: 2.62, 3: 2.53, 4: 2.03, 8: 2.28, 16: 2.20 (16 or 64 lanes
// alike): a dependent fp64 FMA issues every ~1.45 quad-cycles, independent ones every 1.0.
int main() {
  for (int lanes : {16, 64}) { run<1>(lanes); run<2>(lanes); run<3>(lanes); run<4>(lanes); run<8>(lanes); run<16>(lanes); }
  return 0;
}
