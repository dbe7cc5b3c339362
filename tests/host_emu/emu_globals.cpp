#include <hip/hip_runtime.h>
thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;
int od_trace_flag = 0;
extern "C" void od_emu_set_trace(int v) { od_trace_flag = v; }
