#include <hip/hip_runtime.h>
thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;
int od_trace_flag = 0;
int od_emu_lockstep = 0;
thread_local OdEmuRowBus* od_emu_bus = nullptr;
extern "C" void od_emu_set_lockstep(int v) { od_emu_lockstep = v; }
extern "C" void od_emu_set_trace(int v) { od_trace_flag = v; }
thread_local double od_lds[160 * 1024 / 8];   // emulated per-workgroup LDS (one workgroup per OpenMP thread at a time)

// unit-test hooks for the scalar helpers of od_math.h (tests/test_models.py)
#include "od_math.h"
extern "C" void od_emu_sincos(const double* x, long n, double* s, double* c) {
  for (long i = 0; i < n; ++i) od::od_sincos(x[i], s[i], c[i]);
}
extern "C" int od_emu_lu6(const double* A, const double* b, double* x) {
  double a[36]; int piv[6];
  for (int i = 0; i < 36; ++i) a[i] = A[i];
  for (int i = 0; i < 6; ++i) x[i] = b[i];
  const bool ok = od::od_lu_factor<double, 6>(a, piv);
  od::od_lu_solve<double, 6>(a, piv, x);
  return ok ? 1 : 0;
}

// emulated devices (hip/hip_runtime.h): count, per-thread current device, device of the last kernel launch, stream registry
#include <map>
int od_emu_ndev = 1;
thread_local int od_emu_cur_dev = 0;
int od_emu_launch_dev = -1;
static std::map<void*, int> od_emu_streams;
int od_emu_stream_device(void* s) { auto it = od_emu_streams.find(s); return it == od_emu_streams.end() ? od_emu_cur_dev : it->second; }
extern "C" void od_emu_set_device_count(int n) { od_emu_ndev = n > 0 ? n : 1; if (od_emu_cur_dev >= od_emu_ndev) od_emu_cur_dev = 0; }
extern "C" int od_emu_set_device(int d) { return hipSetDevice(d); }
extern "C" int od_emu_get_device(void) { return od_emu_cur_dev; }
extern "C" int od_emu_last_launch_device(void) { return od_emu_launch_dev; }
extern "C" void od_emu_register_stream(void* s, int dev) { od_emu_streams[s] = dev; }
