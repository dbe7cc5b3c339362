// Launch table: one entry per model, filled by the per-model translation units (od_model_*.hip)
// so that the heavy template instantiations compile in parallel.
#pragma once
#include <array>
#include <hip/hip_runtime.h>
#include "od_units.h"

namespace od {

struct LaunchCfg { int ppw, wpb, coop; };     // problems per wavefront (pow2 <= 64), wavefronts per workgroup (1 | 4), cooperative kernel: 0 none, 1 = 16 lanes per problem (od_coop.h), 2 = 8 lanes (od_coop3.h)

struct ModelVT {
  int id, kind;
  const char* name;
  int nq, nu, nz, nth, nfric, nzq, ngc, nfact;
  double r_tol, kappa_eval, kappa_grad, eps_min, kappa_reg, gamma_reg, undercut;
  int max_iter, max_ls;
  double fric_default[4];
  int has_coop;                                // cooperative state kernels, 16 lanes per problem: 0 none, 1 on request, 2 automatic for small batches
  long coop_auto_max;                          // ... up to this many problems
  int has_coop8;                               // the same for the 8-lanes-per-problem form (cones up to dimension 3)
  long coop8_auto_max;
  int ngam, nbfr;                              // z indices of the impact / friction impulses
  std::array<int, 12> gam, bfr;
  hipError_t (*step_state)(const StepArgs<double>&, LaunchCfg, hipStream_t, LiveArgs);   // pass 1, independent knots
  hipError_t (*rollout_state)(const RolloutArgs<double>&, LaunchCfg, hipStream_t);  // pass 1, rollouts
  hipError_t (*grad_knots)(const StepArgs<double>&, hipStream_t, LiveArgs);         // pass 2 (a.B knots)
  hipError_t (*rollout_policy)(const PolicyArgs<double>&, LaunchCfg, hipStream_t);  // closed-loop rollouts
  hipError_t (*bundle)(const BundleArgs<double>&, long, LaunchCfg, hipStream_t);
  hipError_t (*step_full)(const FullArgs<double>&, int ppw, hipStream_t);           // z and dz, every row
  hipError_t (*raw64)(const RawArgs<double>&, int ppw, hipStream_t);
  hipError_t (*raw32)(const RawArgs<float>&, int ppw, hipStream_t);
};

// first n entries of a generated index table, zero-padded
template <int N> constexpr std::array<int, 12> od_pad12(const int (&a)[N], int n) {
  std::array<int, 12> r{};
  for (int i = 0; i < n && i < 12 && i < N; ++i) r[i] = a[i];
  return r;
}

// one launch table per model of the generated registry (gen/model_list.h, written by the generator: the eight models
// of the reference plus whatever `python -m optimization_dynamics_amd.codegen --add spec.py` registered)
#include "gen/model_list.h"
#define OD_DECLARE_VT(name, id) const ModelVT* vt_##name();
OD_FOR_EACH_MODEL(OD_DECLARE_VT)
#undef OD_DECLARE_VT

hipError_t launch_rocket64(const RocketArgs<double>&, int ppw, hipStream_t);
hipError_t launch_rocket32(const RocketArgs<float>&, int ppw, hipStream_t);
hipError_t launch_soc_project64(const SocProjectArgs<double>&, int ppw, hipStream_t);
hipError_t launch_soc_project32(const SocProjectArgs<float>&, int ppw, hipStream_t);
hipError_t launch_rocket_rollout64(const RocketRolloutArgs<double>&, int ppw, hipStream_t);
hipError_t launch_rocket_rollout32(const RocketRolloutArgs<float>&, int ppw, hipStream_t);

constexpr int OD_BLOCK = 64;   // one wavefront per workgroup: units are independent, no LDS

// Lane mapping.  Every lane runs its own interior-point loop, so a wavefront is as slow as its slowest
// lane (iteration counts differ with the contact mode).  When the batch is too small to fill the
// chip (256 CUs x 4 SIMDs), fewer problems per wavefront (ppw < 64, remaining lanes idle) spread
// the work over more SIMDs and shorten each wave's critical path; large batches use all 64 lanes.
struct LaneMap {
  int ppw;                                  // problems per wavefront, power of two in [1, 64]
  // thread t of a block: wavefront t/64, lane t%64 (blocks are 1 or 4 wavefronts)
  __host__ __device__ long problem(unsigned block, unsigned thread, unsigned block_threads = 64) const {
    return ((long)block * (block_threads / 64) + thread / 64) * ppw + ((thread & 63) % ppw);
  }
  // At least 16 lanes execute: with ppw < 16 the lanes ppw..15 repeat the problems of lanes 0..ppw-1 (same
  // loads, same arithmetic, identical stores).  Measured on MI355X (profiles/): this kernel runs 2.5x SLOWER
  // per wavefront with 8 or 4 active lanes than with 16 (SQ_WAIT_INST_ANY 56 % of wave cycles), so fewer
  // problems per wavefront only pay off if a full 16-lane group keeps executing.
  __host__ __device__ bool active(unsigned thread) const { return (int)(thread & 63) < (ppw < 16 ? 16 : ppw); }
};

inline int od_auto_ppw(long n) {
  // A wavefront is as slow as its slowest lane, so spread the batch: one wavefront per SIMD
  // (256 CUs x 4 = 1024 wavefronts) before packing more problems into a wavefront.  (Two 512-register
  // wavefronts cannot share a SIMD; measured: 2048 wavefronts take 1.7x longer than 1024.)
  long p = 1;
  while (p < 64 && n / p > 1024) p <<= 1;
  return (int)p;
}
inline dim3 od_grid(long n, int ppw) { return dim3((unsigned)((n + ppw - 1) / ppw)); }

}  // namespace od
