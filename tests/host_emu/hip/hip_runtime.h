// TEST HARNESS ONLY: a minimal stand-in for <hip/hip_runtime.h> that lets the product's C-ABI
// sources (optimization_dynamics_amd/csrc/*.hip) be compiled with g++ and run their per-lane
// kernels as plain loops.  It exists so that the solver logic and the host-side argument handling
// can be checked against the oracle in the CPU-only test tier; it is built only by
// tests/host_emu/Makefile into tests/host_emu/libod_emu.so and is never loaded by the product.
#pragma once
#include <cstdlib>
#include <cstring>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ thread_local

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
typedef int hipError_t;
typedef void* hipStream_t;
enum { hipSuccess = 0 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };

extern thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;

inline const char* hipGetErrorString(hipError_t) { return "emulated"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
inline hipError_t hipMalloc(void** p, size_t n) { *p = std::malloc(n); return hipSuccess; }
inline hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { std::memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...)                 \
  do {                                                                              \
    dim3 g_ = (grid), b_ = (block);                                                 \
    gridDim = g_; blockDim = b_;                                                    \
    _Pragma("omp parallel for schedule(dynamic, 1)")                                \
    for (long bx_ = 0; bx_ < (long)g_.x; ++bx_) {                                   \
      gridDim = g_; blockDim = b_;                                                  \
      for (unsigned tx_ = 0; tx_ < b_.x; ++tx_) {                                   \
        blockIdx = dim3((unsigned)bx_); threadIdx = dim3(tx_);                      \
        kernel(__VA_ARGS__);                                                        \
      }                                                                             \
    }                                                                               \
  } while (0)
