// Micro-benchmark: how does a register-heavy, dependent fp64 FMA stream scale with the number of
// resident wavefronts on MI355X?  (the solve pass stops scaling at ~256 wavefronts.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int NREG> __global__ __launch_bounds__(256, 1) void k_fma(double* out, int iters, int active_lanes) {
  if ((int)(threadIdx.x & 63) >= active_lanes) return;
  double a[NREG];
  const double s = 1.0 + 1e-9 * threadIdx.x;
#pragma unroll
  for (int i = 0; i < NREG; ++i) a[i] = s + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NREG; ++i) a[i] = __builtin_fma(a[i], 0.999999, a[(i + 1) % NREG] * 1e-7);
  }
  double r = 0;
#pragma unroll
  for (int i = 0; i < NREG; ++i) r += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int NREG> void run(const char* name) {
  double* out;
  hipMalloc(&out, sizeof(double) * 4096 * 256);
  const int iters = 2000;
  for (int wpb : {1, 4}) {
    for (int waves : {64, 128, 256, 512, 1024, 2048}) {
      const int blocks = waves / wpb;
      hipEvent_t e0, e1;
      hipEventCreate(&e0); hipEventCreate(&e1);
      hipLaunchKernelGGL((k_fma<NREG>), dim3(blocks), dim3(64 * wpb), 0, 0, out, 10, 16);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL((k_fma<NREG>), dim3(blocks), dim3(64 * wpb), 0, 0, out, iters, 16);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double instr = (double)waves * iters * NREG * 2;   // fma + mul per element
      printf("%s wpb %d waves %5d: %.3f ms  %.1f G wave-instr/s  (%.2f cycles/instr/wave at 2.4GHz)\n", name, wpb, waves, ms,
             instr / ms / 1e6, ms * 1e-3 * 2.4e9 / (iters * NREG * 2.0));
    }
  }
  hipFree(out);
}

int main() {
  run<32>("regs~64+ ");
  run<100>("regs~200+");
  run<200>("regs~400+");
  return 0;
}
