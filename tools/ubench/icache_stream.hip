// Micro-benchmark: straight-line (fully unrolled) fp64 code of NOPS instructions executed repeatedly by
// W wavefronts.  Does instruction delivery limit scaling when the code does not loop tightly?
#include <hip/hip_runtime.h>
#include <cstdio>

template <int NOPS> __global__ __launch_bounds__(256, 1) void k_stream(double* out, int reps) {
  if ((threadIdx.x & 63) >= 16) return;
  double a0 = 1.0 + 1e-9 * threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int i = 0; i < NOPS / 8; ++i) {       // 8 independent chains -> issue-bound, not latency-bound
      a0 = __builtin_fma(a0, 0.999999, 1e-7 * (i + 1));
      a1 = __builtin_fma(a1, 0.999998, 2e-7 * (i + 1));
      a2 = __builtin_fma(a2, 0.999997, 3e-7 * (i + 1));
      a3 = __builtin_fma(a3, 0.999996, 4e-7 * (i + 1));
      a4 = __builtin_fma(a4, 0.999995, 5e-7 * (i + 1));
      a5 = __builtin_fma(a5, 0.999994, 6e-7 * (i + 1));
      a6 = __builtin_fma(a6, 0.999993, 7e-7 * (i + 1));
      a7 = __builtin_fma(a7, 0.999992, 8e-7 * (i + 1));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int NOPS> void run() {
  double* out;
  hipMalloc(&out, sizeof(double) * 4096 * 256);
  const int reps = 200;
  for (int waves : {64, 256, 512, 1024}) {
    const int blocks = waves / 4;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_stream<NOPS>), dim3(blocks), dim3(256), 0, 0, out, 2);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_stream<NOPS>), dim3(blocks), dim3(256), 0, 0, out, reps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("unrolled %5d fma (~%d KB): waves %5d: %.3f ms  %.1f G wave-instr/s  %.2f cycles/instr/wave@2.4GHz\n", NOPS, NOPS * 16 / 1024,
           waves, ms, (double)waves * reps * NOPS / ms / 1e6, ms * 1e-3 * 2.4e9 / ((double)reps * NOPS));
  }
  hipFree(out);
}

int main() {
  run<256>();
  run<2048>();
  run<4096>();
  run<8192>();
  return 0;
}
