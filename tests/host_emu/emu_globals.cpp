#include <hip/hip_runtime.h>
thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;
