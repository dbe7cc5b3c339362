#include <hip/hip_runtime.h>
thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;
int od_trace_flag = 0;
extern "C" void od_emu_set_trace(int v) { od_trace_flag = v; }
thread_local double od_lds[160 * 1024 / 8];   // emulated per-workgroup LDS (one workgroup per OpenMP thread at a time)
