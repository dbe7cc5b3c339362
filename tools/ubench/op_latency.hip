// Issue cost of the instructions a lone wavefront per SIMD executes in the solve loop: cycles per instruction of dependent chains
// (one wavefront, 64 lanes), s_memtime around 4096 instructions of each kind.
//   hipcc --offload-arch=gfx950 -O3 -o op_latency op_latency.hip && ./op_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
template <int KIND> __global__ void k(double* out, long long* cyc, double seed) {
  double a = seed + threadIdx.x * 1e-3, b = 1.0000001, c = 0.999, d = a + 1.0;
  float f = (float)a;
  long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < 64; ++r) {
    if constexpr (KIND == 0) { REP64(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));) }
    if constexpr (KIND == 1) { REP64(asm volatile("v_rcp_f64 %0, %0" : "+v"(a));) }
    if constexpr (KIND == 2) { REP64(asm volatile("v_rsq_f64 %0, %0" : "+v"(a));) }
    if constexpr (KIND == 3) { REP64(asm volatile("v_rcp_f32 %0, %0" : "+v"(f));) }
    if constexpr (KIND == 4) { REP64(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(f) : "v"(f));) }
    if constexpr (KIND == 5) { REP64(asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(d), "v"(b));) }
    if constexpr (KIND == 6) { REP64(asm volatile("v_fma_f64 %0, %0, %2, %3\n\tv_fma_f64 %1, %1, %2, %3" : "+v"(a), "+v"(d) : "v"(b), "v"(c));) }   // two independent chains
    if constexpr (KIND == 7) { REP64(asm volatile("v_rcp_f64 %0, %0\n\tv_fma_f64 %1, %1, %2, %3\n\tv_fma_f64 %1, %1, %2, %3\n\tv_fma_f64 %1, %1, %2, %3" : "+v"(a), "+v"(d) : "v"(b), "v"(c));) }  // rcp + 3 independent fma
    if constexpr (KIND == 8) { REP64(asm volatile("v_mov_b64 %0, %1" : "+v"(a) : "v"(d));) }
    if constexpr (KIND == 9) { REP64(asm volatile("v_cvt_f32_f64 %1, %0\n\tv_rcp_f32 %1, %1\n\tv_cvt_f64_f32 %0, %1" : "+v"(a), "+v"(f));) }
    if constexpr (KIND == 10) { REP64(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(b));) }
    if constexpr (KIND == 11) { REP64(asm volatile("v_max_f64 %0, %0, %1" : "+v"(a) : "v"(b));) }
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
  out[threadIdx.x] = a + d + f;
}

int main() {
  double* d; long long* c;
  (void)hipMalloc(&d, 64 * sizeof(double)); (void)hipMalloc(&c, sizeof(long long));
  const char* names[] = {"v_fma_f64 dependent", "v_rcp_f64 dependent", "v_rsq_f64 dependent", "v_rcp_f32 dependent", "v_cndmask_b32 dependent",
                         "s_nop 1 + v_fmac_f64_dpp dependent (2 instr)", "v_fma_f64 x2 independent chains (2 instr)", "v_rcp_f64 + 3 independent v_fma_f64 (4 instr)",
                         "v_mov_b64", "cvt + v_rcp_f32 + cvt (3 instr)", "v_mul_f64 dependent", "v_max_f64 dependent"};
  long long h;
#define RUN(K_) hipLaunchKernelGGL(k<K_>, dim3(1), dim3(64), 0, 0, d, c, 1.2345); (void)hipDeviceSynchronize(); hipLaunchKernelGGL(k<K_>, dim3(1), dim3(64), 0, 0, d, c, 1.2345); \
  (void)hipMemcpy(&h, c, sizeof(h), hipMemcpyDeviceToHost); printf("%-52s %7.2f counter ticks per group (4096 groups)\n", names[K_], (double)h / 4096.0);
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11)
  printf("(readcyclecounter = s_memtime; one wavefront on an otherwise idle chip)\n");
  return 0;
}
