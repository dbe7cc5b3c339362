// TEST HARNESS ONLY: a minimal stand-in for <hip/hip_runtime.h> that lets the product's C-ABI
// sources (optimization_dynamics_amd/csrc/*.hip) be compiled with g++ and run their per-lane
// kernels as plain loops.  It exists so that the solver logic and the host-side argument handling
// can be checked against the oracle in the CPU-only test tier; it is built only by
// tests/host_emu/Makefile into tests/host_emu/libod_emu.so and is never loaded by the product.
#pragma once
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

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
// devices: od_emu_set_device_count(n) makes n emulated devices visible; the current device is per thread, as in HIP.  Every
// kernel launch records the device that was current (od_emu_last_launch_device) so that the tests can see the handle's device
// guard (od_capi.hip::OnDevice) at work; streams are plain pointers whose device the tests register (od_emu_register_stream).
extern int od_emu_ndev;
extern thread_local int od_emu_cur_dev;
extern int od_emu_launch_dev;
int od_emu_stream_device(void* s);
inline hipError_t hipGetDeviceCount(int* n) { *n = od_emu_ndev; return hipSuccess; }
inline hipError_t hipGetDevice(int* d) { *d = od_emu_cur_dev; return hipSuccess; }
inline hipError_t hipSetDevice(int d) { if (d < 0 || d >= od_emu_ndev) return 101; od_emu_cur_dev = d; return hipSuccess; }
inline hipError_t hipStreamGetDevice(void* s, int* d) { *d = od_emu_stream_device(s); return hipSuccess; }
inline hipError_t hipMalloc(void** p, size_t n) { *p = std::malloc(n); return hipSuccess; }
inline hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
enum { hipHostMallocMapped = 2, hipHostMallocCoherent = 0x40000000 };
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = std::malloc(n); return hipSuccess; }
inline hipError_t hipHostGetDevicePointer(void** d, void* h, unsigned) { *d = h; return hipSuccess; }
inline hipError_t hipHostFree(void* p) { std::free(p); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { std::memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { std::memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { std::memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemset(void* d, int v, size_t n) { std::memset(d, v, n); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
typedef void* hipEvent_t;
enum { hipEventDisableTiming = 2 };
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = (void*)1; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline void __syncthreads() {}            // (host build: kernels that use it are launched with one thread per workgroup)
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }

// ---- lockstep rows (od_emu_set_lockstep(1)): the 16 threads of a DPP row run as 16 host threads that meet at every
// cross-lane operation, so the lane cooperation of od_solver.h (row rotations between the copies of a problem: shared
// step-length tests, parallel line search) executes in the CPU test tier exactly as written for the device.
struct OdEmuRowBus {
  std::mutex m;
  std::condition_variable cv;
  int n = 0, waiting = 0;
  unsigned long gen = 0;
  uint64_t slot[16];
  bool present[16];
  void barrier_locked(std::unique_lock<std::mutex>& lk) {
    const unsigned long g = gen;
    if (++waiting == n) { waiting = 0; ++gen; cv.notify_all(); }
    else cv.wait(lk, [&] { return gen != g; });
  }
  // lane `lane` contributes v and receives the contribution of lane `src`; a lane that has left the kernel reads as
  // the receiver's own value (the device's DPP `old` operand under an inactive source lane)
  uint64_t exchange(int lane, int src, uint64_t v) {
    std::unique_lock<std::mutex> lk(m);
    slot[lane] = v;
    barrier_locked(lk);
    const uint64_t r = present[src] ? slot[src] : v;
    barrier_locked(lk);
    return r;
  }
  void leave(int lane) {
    std::unique_lock<std::mutex> lk(m);
    present[lane] = false;
    --n;
    if (n > 0 && waiting == n) { waiting = 0; ++gen; cv.notify_all(); }
  }
};
extern int od_emu_lockstep;
extern thread_local OdEmuRowBus* od_emu_bus;
#define OD_HOST_EMU_LOCKSTEP 1
// row_ror:R -- lane l receives lane (l + 16 - R) % 16 of its row (identity when rows do not run in lockstep)
inline uint64_t od_emu_row_ror_bits(uint64_t v, int R) {
  if (!od_emu_bus) return v;
  const int lane = (int)(threadIdx.x & 15);
  return od_emu_bus->exchange(lane, (lane + 16 - R) & 15, v);
}

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...)                 \
  do {                                                                              \
    dim3 g_ = (grid), b_ = (block);                                                 \
    od_emu_launch_dev = od_emu_cur_dev;                                             \
    gridDim = g_; blockDim = b_;                                                    \
    if (od_emu_lockstep) {                                                          \
      for (long bx_ = 0; bx_ < (long)g_.x; ++bx_) {                                 \
        for (unsigned r0_ = 0; r0_ < b_.x; r0_ += 16) {                             \
          OdEmuRowBus bus_;                                                         \
          const unsigned nl_ = b_.x - r0_ < 16 ? b_.x - r0_ : 16;                   \
          bus_.n = (int)nl_;                                                        \
          for (unsigned l_ = 0; l_ < 16; ++l_) bus_.present[l_] = l_ < nl_;         \
          std::vector<std::thread> ts_;                                             \
          for (unsigned l_ = 0; l_ < nl_; ++l_)                                     \
            ts_.emplace_back([&, l_]() {                                            \
              gridDim = g_; blockDim = b_;                                          \
              blockIdx = dim3((unsigned)bx_); threadIdx = dim3(r0_ + l_);           \
              od_emu_bus = &bus_;                                                   \
              kernel(__VA_ARGS__);                                                  \
              od_emu_bus = nullptr;                                                 \
              bus_.leave((int)l_);                                                  \
            });                                                                     \
          for (auto& t_ : ts_) t_.join();                                           \
        }                                                                           \
      }                                                                             \
      break;                                                                        \
    }                                                                               \
    _Pragma("omp parallel for schedule(dynamic, 1)")                                \
    for (long bx_ = 0; bx_ < (long)g_.x; ++bx_) {                                   \
      gridDim = g_; blockDim = b_;                                                  \
      for (unsigned tx_ = 0; tx_ < b_.x; ++tx_) {                                   \
        blockIdx = dim3((unsigned)bx_); threadIdx = dim3(tx_);                      \
        kernel(__VA_ARGS__);                                                        \
      }                                                                             \
    }                                                                               \
  } while (0)
