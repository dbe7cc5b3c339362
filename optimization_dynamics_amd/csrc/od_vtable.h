// Launch table: one entry per model, filled by the per-model translation units (od_model_*.hip)
// so that the heavy template instantiations compile in parallel.
#pragma once
#include <hip/hip_runtime.h>
#include "od_units.h"

namespace od {

struct ModelVT {
  int id, kind;
  const char* name;
  int nq, nu, nz, nth, nfric, nzq, ngc;
  double r_tol, kappa_eval, kappa_grad, eps_min, kappa_reg, gamma_reg, undercut;
  int max_iter, max_ls;
  double fric_default[4];
  hipError_t (*step)(const StepArgs<double>&, hipStream_t);          // mech models
  hipError_t (*rollout)(const RolloutArgs<double>&, hipStream_t);    // mech models
  hipError_t (*bundle)(const BundleArgs<double>&, long, hipStream_t);
  hipError_t (*raw64)(const RawArgs<double>&, hipStream_t);
  hipError_t (*raw32)(const RawArgs<float>&, hipStream_t);
};

const ModelVT* vt_acrobot_impact();
const ModelVT* vt_acrobot_nominal();
const ModelVT* vt_cartpole_friction();
const ModelVT* vt_cartpole_frictionless();
const ModelVT* vt_planar_push();
const ModelVT* vt_rocket_dynamics();
const ModelVT* vt_rocket_projection();
const ModelVT* vt_hopper();

hipError_t launch_rocket64(const RocketArgs<double>&, hipStream_t);
hipError_t launch_rocket32(const RocketArgs<float>&, hipStream_t);

constexpr int OD_BLOCK = 64;   // one wavefront per workgroup: units are independent, no LDS

inline dim3 od_grid(long n) { return dim3((unsigned)((n + OD_BLOCK - 1) / OD_BLOCK)); }

}  // namespace od
