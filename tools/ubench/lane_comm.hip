// Micro-benchmark + semantics check of the cross-lane primitives the cooperative (lane = KKT row) solver uses
// on gfx950: 64-bit DPP row_newbcast (v_mov_b64_dpp / folded into v_fma_f64_dpp), 32-bit DPP pairs emitted by
// the compiler builtin, DPP row rotations, ds_bpermute, ds_swizzle, an LDS write->read hand-over.
// One wavefront per SIMD (256 workgroups x 256 threads), dependent chains: ns per link of the chain.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>

template <int L> __device__ __forceinline__ double bcast_mov64(double v) {       // hand-written, explicit hazard nop
  double r;
  asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(L));
  return r;
}
template <int L> __device__ __forceinline__ double bcast_mov64_nonop(double v) {  // no nop: is the hazard real?
  double r;
  asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(L));
  return r;
}
// VOP3 has no DPP on gfx9, so v_fma_f64 cannot take a DPP operand; the VOP2 form v_fmac_f64 (gfx90a+) can:
template <int L> __device__ __forceinline__ double fma_bcast(double a, double b, double c) {   // a[lane L of the row] * b + c
  asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(c) : "v"(a), "v"(b), "n"(L));
  return c;
}
template <int L> __device__ __forceinline__ double fma_bcast_nonop(double a, double b, double c) {
  asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(c) : "v"(a), "v"(b), "n"(L));
  return c;
}
template <int L> __device__ __forceinline__ double bcast_b32x2(double v) {        // compiler builtin: hazards handled by hipcc
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x150 + L, 0xF, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x150 + L, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
template <int R> __device__ __forceinline__ double row_ror(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x120 + R, 0xF, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x120 + R, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double bperm(double v, int lane) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_ds_bpermute(lane << 2, lo);
  hi = __builtin_amdgcn_ds_bpermute(lane << 2, hi);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double swz_bcast3(double v) {   // ds_swizzle bitmask mode: lane' = (lane & 0x10) | 3
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_ds_swizzle(lo, (0x10) | (3 << 5) | (0 << 10));
  hi = __builtin_amdgcn_ds_swizzle(hi, (0x10) | (3 << 5) | (0 << 10));
  return __hiloint2double(hi, lo);
}

enum { K_FMA = 0, K_FMA_DPP, K_FMA_DPP_NONOP, K_MOV64, K_MOV64_NONOP, K_B32X2, K_ROR_MIN, K_BPERM, K_SWZ, K_LDS, K_RCP, NK };
static const char* NAMES[NK] = {"fma (plain dependent chain)", "v_fmac_f64_dpp row_newbcast (+s_nop 1)", "v_fmac_f64_dpp row_newbcast (no nop)",
                                "v_mov_b64_dpp + fma (+s_nop 1)", "v_mov_b64_dpp + fma (no nop)", "2x v_mov_b32_dpp (builtin) + fma",
                                "row min over 16 lanes (4x row_ror + min)", "ds_bpermute x2 + fma", "ds_swizzle x2 + fma", "LDS write -> read (other lane) + fma",
                                "rcp + 3 Newton steps"};

template <int K> __global__ __launch_bounds__(256, 1) void k_chain(double* out, int iters) {
  __shared__ double lds[256];
  const int lane = threadIdx.x & 63;
  double a = 1.0 + 1e-3 * lane;
  const double m = 0.999999, c = 1e-7;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      if constexpr (K == K_FMA) a = __builtin_fma(a, m, c);
      if constexpr (K == K_FMA_DPP) a = fma_bcast<3>(a, m, c);
      if constexpr (K == K_FMA_DPP_NONOP) a = fma_bcast_nonop<3>(a, m, c);
      if constexpr (K == K_MOV64) a = __builtin_fma(bcast_mov64<3>(a), m, c);
      if constexpr (K == K_MOV64_NONOP) a = __builtin_fma(bcast_mov64_nonop<3>(a), m, c);
      if constexpr (K == K_B32X2) a = __builtin_fma(bcast_b32x2<3>(a), m, c);
      if constexpr (K == K_ROR_MIN) {
        a = fmin(a, row_ror<8>(a)); a = fmin(a, row_ror<4>(a)); a = fmin(a, row_ror<2>(a)); a = fmin(a, row_ror<1>(a));
        a = __builtin_fma(a, m, c + 1e-9 * lane);
      }
      if constexpr (K == K_BPERM) a = __builtin_fma(bperm(a, (lane & 48) | 3), m, c);
      if constexpr (K == K_SWZ) a = __builtin_fma(swz_bcast3(a), m, c);
      if constexpr (K == K_LDS) {
        lds[threadIdx.x] = a;
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
        a = __builtin_fma(lds[(threadIdx.x & ~15) | 3], m, c);
      }
      if constexpr (K == K_RCP) {
        double x = a, rr = __builtin_amdgcn_rcp(x);
        rr = __builtin_fma(__builtin_fma(-x, rr, 1.0), rr, rr);
        rr = __builtin_fma(__builtin_fma(-x, rr, 1.0), rr, rr);
        rr = __builtin_fma(__builtin_fma(-x, rr, 1.0), rr, rr);
        a = rr;
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}

template <int K> void run(double* out) {
  const int iters = 20000, blocks = 256;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k_chain<K>), dim3(blocks), dim3(256), 0, 0, out, 10);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k_chain<K>), dim3(blocks), dim3(256), 0, 0, out, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-48s %.2f ns per link\n", NAMES[K], ms * 1e6 / ((double)iters * 8));
}

// semantics: every lane holds v = 100*lane + 7; after the broadcast from lane 3 of its row it must hold 100*((lane&~15)|3) + 7,
// also when the source was written by the immediately preceding VALU instruction (the DPP read-after-write hazard)
__global__ void k_check(double* out, int* bad) {
  const int lane = threadIdx.x & 63;
  const double base = 100.0 * lane + 7.0;
  const double want = 100.0 * ((lane & ~15) | 3) + 7.0;
  int b = 0;
  { double v = base * 1.0; v = bcast_mov64<3>(v); if (v != want) b |= 1; }
  { double v = base + 0.0; v = bcast_mov64_nonop<3>(v); if (v != want) b |= 2; }
  { double v = base * 1.0; v = fma_bcast<3>(v, 1.0, 0.0); if (v != want) b |= 4; }
  { double v = base + 0.0; v = fma_bcast_nonop<3>(v, 1.0, 0.0); if (v != want) b |= 8; }
  { double v = base * 1.0; v = bcast_b32x2<3>(v); if (v != want) b |= 16; }
  { double v = base; v = swz_bcast3(v); if (v != 100.0 * ((lane & ~15 & 31) | (lane & 32) | 3) + 7.0) b |= 32; }
  { double v = base; v = bperm(v, (lane & 48) | 3); if (v != want) b |= 64; }
  // partial exec: only even lanes execute; the broadcast source lane 3 is inactive -> what do the active lanes read?
  double pv = -1.0;
  if ((lane & 1) == 0) { double v = base * 1.0; pv = bcast_mov64<3>(v); }
  out[threadIdx.x] = pv;
  bad[threadIdx.x] = b;
}

int main() {
  double* out; int* bad;
  (void)hipMalloc(&out, sizeof(double) * 256 * 256);
  (void)hipMalloc(&bad, sizeof(int) * 64);
  hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, out, bad);
  (void)hipDeviceSynchronize();
  std::vector<int> hb(64); std::vector<double> ho(64);
  (void)hipMemcpy(hb.data(), bad, 64 * sizeof(int), hipMemcpyDeviceToHost);
  (void)hipMemcpy(ho.data(), out, 64 * sizeof(double), hipMemcpyDeviceToHost);
  int orall = 0; for (int i = 0; i < 64; ++i) orall |= hb[i];
  printf("semantics check: failure bits (1 mov64, 2 mov64-nonop, 4 fma_dpp, 8 fma_dpp-nonop, 16 b32x2, 32 swizzle, 64 bpermute) = %d\n", orall);
  printf("partial exec (source lane 3 inactive), lanes 0,2,16,18 read: %g %g %g %g (active source would give 307, 307, 1907, 1907)\n", ho[0], ho[2], ho[16], ho[18]);
  run<K_FMA>(out); run<K_FMA_DPP>(out); run<K_FMA_DPP_NONOP>(out); run<K_MOV64>(out); run<K_MOV64_NONOP>(out); run<K_B32X2>(out);
  run<K_ROR_MIN>(out); run<K_BPERM>(out); run<K_SWZ>(out); run<K_LDS>(out); run<K_RCP>(out);
  return 0;
}
