// Fused rocket kernels: thrust-cone SOCP projection -> implicit-midpoint dynamics solve ->
// implicit gradients and the 12x3 * 3x3 chain product, one lane per knot
// (src/models/rocket/dynamics.jl:101-268).
#include "od_vtable.h"
#include "gen/rocket_dynamics.h"
#include "od_rocket_proj_direct.h"

namespace od {

template <class T> __global__ __launch_bounds__(OD_BLOCK) void k_rocket(RocketArgs<T> a, LaneMap lm) {
  if (a.skip && *a.skip) return;
  const long b = lm.problem(blockIdx.x, threadIdx.x);
  if (lm.active(threadIdx.x) && b < a.B && (!a.live || a.live[b % a.live_mod])) unit_rocket<Model_rocket_dynamics, Model_rocket_projection_direct, T>(a, b);
}

hipError_t launch_rocket64(const RocketArgs<double>& a, int ppw, hipStream_t s) {
  hipLaunchKernelGGL((k_rocket<double>), od_grid(a.B, ppw), dim3(OD_BLOCK), 0, s, a, LaneMap{ppw});
  return hipGetLastError();
}
hipError_t launch_rocket32(const RocketArgs<float>& a, int ppw, hipStream_t s) {
  hipLaunchKernelGGL((k_rocket<float>), od_grid(a.B, ppw), dim3(OD_BLOCK), 0, s, a, LaneMap{ppw});
  return hipGetLastError();
}

template <class T> __global__ __launch_bounds__(OD_BLOCK) void k_soc_project(SocProjectArgs<T> a, LaneMap lm) {
  const long b = lm.problem(blockIdx.x, threadIdx.x);
  if (lm.active(threadIdx.x) && b < a.a.B) unit_soc_project<Model_rocket_projection_direct, T>(a, b);
}
hipError_t launch_soc_project64(const SocProjectArgs<double>& a, int ppw, hipStream_t s) {
  hipLaunchKernelGGL((k_soc_project<double>), od_grid(a.a.B, ppw), dim3(OD_BLOCK), 0, s, a, LaneMap{ppw});
  return hipGetLastError();
}
hipError_t launch_soc_project32(const SocProjectArgs<float>& a, int ppw, hipStream_t s) {
  hipLaunchKernelGGL((k_soc_project<float>), od_grid(a.a.B, ppw), dim3(OD_BLOCK), 0, s, a, LaneMap{ppw});
  return hipGetLastError();
}

template <class T> __global__ __launch_bounds__(OD_BLOCK) void k_rocket_rollout(RocketRolloutArgs<T> a, LaneMap lm) {
  if (a.a.skip && *a.a.skip) return;
  const long p = lm.problem(blockIdx.x, threadIdx.x);
  if (lm.active(threadIdx.x) && p < a.a.B && (!a.a.live || a.a.live[p % a.a.live_mod])) unit_rocket_rollout<Model_rocket_dynamics, Model_rocket_projection_direct, T>(a, p);
}
hipError_t launch_rocket_rollout64(const RocketRolloutArgs<double>& a, int ppw, hipStream_t s) {
  hipLaunchKernelGGL((k_rocket_rollout<double>), od_grid(a.a.B, ppw), dim3(OD_BLOCK), 0, s, a, LaneMap{ppw});
  return hipGetLastError();
}
hipError_t launch_rocket_rollout32(const RocketRolloutArgs<float>& a, int ppw, hipStream_t s) {
  hipLaunchKernelGGL((k_rocket_rollout<float>), od_grid(a.a.B, ppw), dim3(OD_BLOCK), 0, s, a, LaneMap{ppw});
  return hipGetLastError();
}

}  // namespace od
