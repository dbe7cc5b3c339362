// Per-problem interior-point step + implicit gradient, one problem per GPU lane.
//
// Replaces the reference's per-timestep inner solve
//   RoboDojo.step!(sim, q2, v1, u, 1)                       (src/dynamics.jl:88,103,123)
//   interior_point_solve!(ip)                               (src/models/rocket/dynamics.jl:109..262)
// for all three of the reference's per-knot calls (f at kappa_eval; fx and fu at kappa_grad) in ONE
// pass: with undercut = Inf (src/dynamics.jl:26) the centering target sigma*mu does not depend on
// kappa_tol, so the kappa_grad solve's iterates are a prefix of the kappa_eval solve's.  The loop
// snapshots the implicit gradient dz = -rz^{-1} rtheta at the first iterate satisfying
// (r_tol, kappa_grad) and the state at the first iterate satisfying (r_tol, kappa_eval).
//
// All model structure (index sets, sparse KKT elimination) is compile-time (csrc/gen/*.h), every
// loop below is fully unrolled and every array lives in registers.
#pragma once
#include <type_traits>
#include "od_math.h"

#ifdef OD_TRACE   // test-harness builds only (tests/host_emu): per-iteration trace to stdout
#include <cstdio>
extern int od_trace_flag;
#define OD_TRACE_IT(...) do { if (od_trace_flag) std::printf(__VA_ARGS__); } while (0)
#else
#define OD_TRACE_IT(...)
#endif

// guards of the cone step in single precision (measured, 4096 thrust projections: 1e-14 / 1e-25 as in double: 17 run into
// max_iter with a cone variable rounded onto the boundary; 1e-8 / 1e-14: 1; 1e-7 / 1e-12: 0)
#ifndef OD_SOC_EPS_F32
#define OD_SOC_EPS_F32 1e-7
#define OD_SOC_FLOOR_F32 1e-12
#endif

namespace od {

template <class T> struct Opts {
  T r_tol, kappa_eval, kappa_grad, eps_min, kappa_reg, gamma_reg, undercut_inv;
  int max_iter, max_ls;
  // launch-side hint, not a solver option: when a wavefront carries ppw = 1, 2 or 4 problems, lanes
  // 0..15 hold 16/ppw copies of each (LaneMap, od_vtable.h); `coop` = ppw lets the copies share out
  // independent pieces of work and combine them across lanes.  0 = every lane does everything.
  int coop;
};

enum : int { OD_ST_EVAL_OK = 1, OD_ST_GRAD_OK = 2, OD_ST_FACTOR_OK = 4 };

// max |r_i| over the equality / complementarity rows, NaN-sticky (a NaN residual must fail every
// convergence and line-search test): hardware max drops NaNs, so a running sum carries them instead
template <class M, class T> OD_HD T viol_eq(const T* r) {
  T v = T(0), s = T(0);
#pragma unroll
  for (int i = 0; i < M::NEQ; ++i) { const T a = od_abs(r[M::EQUR[i]]); v = od_fmax(v, a); s += a; }
  return (s != s) ? s : v;
}
template <class M, class T> OD_HD T viol_bil(const T* r) {
  T v = T(0), s = T(0);
  if constexpr (M::NBIL > 0) {
#pragma unroll
    for (int i = 0; i < M::NBIL; ++i) { const T a = od_abs(r[M::BIL[i]]); v = od_fmax(v, a); s += a; }
  }
  return (s != s) ? s : v;
}

// CVXOPT sec. 8.2 step to the boundary of a second-order cone for lam + alpha*dlt
template <int N, class T> OD_HD T soc_step_one(const T* lam, const T* dlt, T tau) {
  // (guards of the double-precision formula; in single precision -- the rocket models only, the reference has none --
  // they sit at the resolution of float instead of 1e7 below it)
  const T eps = sizeof(T) == 4 ? T(OD_SOC_EPS_F32) : T(1e-14);
  const T l0 = lam[0];
  T ll = l0 * l0, ld = l0 * dlt[0];
#pragma unroll
  for (int i = 1; i < N; ++i) { ll -= lam[i] * lam[i]; ld -= lam[i] * dlt[i]; }
  ll = od_max(ll, sizeof(T) == 4 ? T(OD_SOC_FLOOR_F32) : T(1e-25)) + eps;
  ld += eps;
  const T isq = od_rsqrt(ll), ill = isq * isq;
  const T rs = ld * ill;
  const T c = (ld * isq + dlt[0]) * od_rcp(l0 * isq + T(1));
  T nv = T(0);
  if constexpr (N == 2) {
    nv = od_abs(dlt[1] * isq - c * lam[1] * ill);     // the norm of a 1-vector: no square root
  } else {
#pragma unroll
    for (int i = 1; i < N; ++i) { const T rv = dlt[i] * isq - c * lam[i] * ill; nv += rv * rv; }
    nv = od_sqrt(nv);
  }
  T a = T(1);
  if (nv - rs > T(0)) a = od_min(a, tau * od_rcp(nv - rs));
  return a;
}

template <class M, int C, class T> OD_HD T soc_step_cone(const T* z, const T* D, T tau, T a) {
  constexpr int o = M::SOCOFF[C], n = M::SOCOFF[C + 1] - M::SOCOFF[C];
  T lam[n], dl[n];
#pragma unroll
  for (int i = 0; i < n; ++i) { lam[i] = z[M::SOC1[o + i]]; dl[i] = -D[M::SOC1[o + i]]; }
  a = od_min(a, soc_step_one<n>(lam, dl, tau));
#pragma unroll
  for (int i = 0; i < n; ++i) { lam[i] = z[M::SOC2[o + i]]; dl[i] = -D[M::SOC2[o + i]]; }
  a = od_min(a, soc_step_one<n>(lam, dl, tau));
  if constexpr (C + 1 < M::NSOC) return soc_step_cone<M, C + 1>(z, D, tau, a);
  else return a;
}

// ---- lane cooperation between the copies of one problem (Opts::coop) --------------------------------
// Lanes 0..15 of a wavefront are one DPP row; the copies of a problem sit ppw lanes apart, so rotating
// the row by ppw and 2*ppw reaches four of them.
#if defined(__HIP_DEVICE_COMPILE__) || defined(OD_HOST_EMU_LOCKSTEP)
#if defined(__HIP_DEVICE_COMPILE__)
template <int R> __device__ __forceinline__ double od_row_ror(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x120 + R, 0xF, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x120 + R, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
template <int R> __device__ __forceinline__ float od_row_ror(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x120 + R, 0xF, 0xF, false));
}
template <int R> __device__ __forceinline__ int od_row_ror(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x120 + R, 0xF, 0xF, false); }
#else   // host test build with lockstep rows (tests/host_emu/hip/hip_runtime.h): the same rotations between host threads
template <int R> inline double od_row_ror(double v) { uint64_t b; __builtin_memcpy(&b, &v, 8); b = od_emu_row_ror_bits(b, R); __builtin_memcpy(&v, &b, 8); return v; }
template <int R> inline float od_row_ror(float v) { uint32_t b; __builtin_memcpy(&b, &v, 4); b = (uint32_t)od_emu_row_ror_bits(b, R); __builtin_memcpy(&v, &b, 4); return v; }
template <int R> inline int od_row_ror(int v) { return (int)(uint32_t)od_emu_row_ror_bits((uint32_t)v, R); }
#endif
template <class T> OD_HD T coop_min4(T v, int ppw) {
  if (ppw == 4) { v = od_min(v, od_row_ror<4>(v)); v = od_min(v, od_row_ror<8>(v)); }
  else if (ppw == 2) { v = od_min(v, od_row_ror<2>(v)); v = od_min(v, od_row_ror<4>(v)); }
  else { v = od_min(v, od_row_ror<1>(v)); v = od_min(v, od_row_ror<2>(v)); }
  return v;
}
OD_HD int coop_group(int ppw) { return (((int)threadIdx.x & 63) / ppw) & 3; }
// bitwise OR over ALL 16/ppw copies of a problem, and the index of a lane among them
OD_HD int coop_or_all(int v, int ppw) {
  if (ppw <= 1) v |= od_row_ror<1>(v);
  if (ppw <= 2) v |= od_row_ror<2>(v);
  v |= od_row_ror<4>(v);
  v |= od_row_ror<8>(v);
  return v;
}
OD_HD int coop_copy(int ppw) { return ((int)threadIdx.x & 15) / ppw; }
#else   // host builds without lockstep rows: lanes run one after the other, no cooperation (Opts::coop = 0)
template <class T> OD_HD T coop_min4(T v, int) { return v; }
OD_HD int coop_group(int) { return 0; }
OD_HD int coop_or_all(int v, int) { return v; }
OD_HD int coop_copy(int) { return 0; }
#endif

template <class M> constexpr bool soc_uniform() {
  if constexpr (M::NSOC > 0) {          // (SOCOFF has NSOC + 1 entries; a model without cones has a one-entry placeholder)
    for (int c = 0; c < M::NSOC; ++c)
      if (M::SOCOFF[c + 1] - M::SOCOFF[c] != M::SOCOFF[1] - M::SOCOFF[0]) return false;
    return true;
  } else {
    return false;
  }
}
template <class T> OD_HD T coop_pick(int g, T v0, T v1, T v2, T v3) {
  const T lo = (g & 1) ? v1 : v0, hi = (g & 1) ? v3 : v2;
  return (g & 2) ? hi : lo;
}
// The 2*NSOC cone steps (primal and dual variable of every cone) shared out over the four copies:
// copy g takes steps g, g+4, ...; requires cones of one dimension.
template <class M, class T> OD_HD T soc_step_coop(const T* z, const T* D, T tau, int ppw) {
  constexpr int n = M::SOCOFF[1] - M::SOCOFF[0], NU = 2 * M::NSOC;
  const int g = coop_group(ppw);
  T a = T(1);
#pragma unroll
  for (int r0 = 0; r0 < NU; r0 += 4) {
    T lam[n], dl[n];
#pragma unroll
    for (int i = 0; i < n; ++i) {
      // step u = 2*cone + (0 primal | 1 dual); steps past the end repeat the last one (harmless under min)
      auto idx = [](int u, int i_) constexpr { u = u < NU ? u : NU - 1; return ((u & 1) ? M::SOC2 : M::SOC1)[M::SOCOFF[u >> 1] + i_]; };
      const int k0 = idx(r0, i), k1 = idx(r0 + 1, i), k2 = idx(r0 + 2, i), k3 = idx(r0 + 3, i);
      lam[i] = coop_pick(g, z[k0], z[k1], z[k2], z[k3]);
      dl[i] = -coop_pick(g, D[k0], D[k1], D[k2], D[k3]);
    }
    a = od_min(a, soc_step_one<n>(lam, dl, tau));
  }
  return a;                    // this copy's share: the caller combines the four copies
}

// The 2*NORT orthant ratio tests shared out the same way: copy g takes tests g, g+4, ... and returns its own minimum.
template <class M, class T> OD_HD T ort_step_coop(const T* z, const T* D, T tau, int ppw) {
  constexpr int NT = 2 * M::NORT;
  const int g = coop_group(ppw);
  T num = T(1), den = T(1);
#pragma unroll
  for (int r0 = 0; r0 < NT; r0 += 4) {
    auto idx = [](int t) constexpr { t = t < NT ? t : NT - 1; return (t & 1) ? M::ORT2[t >> 1] : M::ORT1[t >> 1]; };   // past the end: repeat
    const int k0 = idx(r0), k1 = idx(r0 + 1), k2 = idx(r0 + 2), k3 = idx(r0 + 3);
    const T zk = coop_pick(g, z[k0], z[k1], z[k2], z[k3]), dk = coop_pick(g, D[k0], D[k1], D[k2], D[k3]);
    const T n1 = tau * zk;
    if (dk > T(0) && n1 * den < num * dk) { num = n1; den = dk; }
  }
  return num * od_rcp(den);
}

// largest alpha in (0,1] keeping z - alpha*D inside the cones (fractions tau_ort / tau_soc)
template <class M, class T> OD_HD T step_length(const T* z, const T* D, T tau_ort, T tau_soc, int coop) {
  constexpr bool COOP_OK = (M::NSOC == 0) || soc_uniform<M>();
  if constexpr (COOP_OK && (M::NORT + M::NSOC) > 0) {
    if (coop) {
      // every copy of the problem does a quarter of the tests; one combine for orthant and cone tests together
      T a = T(1);
      if constexpr (M::NORT > 0) a = ort_step_coop<M>(z, D, tau_ort, coop);
      if constexpr (M::NSOC > 0) a = od_min(a, soc_step_coop<M>(z, D, tau_soc, coop));
      return coop_min4(a, coop);
    }
  }
  T a = T(1);
  if constexpr (M::NORT > 0) {
    // min over the ratio tests tau*z_k/D_k (D_k > 0) kept as a fraction: one reciprocal at the end
    T num = T(1), den = T(1);
#pragma unroll
    for (int i = 0; i < M::NORT; ++i) {
      const int k1 = M::ORT1[i], k2 = M::ORT2[i];
      const T n1 = tau_ort * z[k1], n2 = tau_ort * z[k2];
      if (D[k1] > T(0) && n1 * den < num * D[k1]) { num = n1; den = D[k1]; }
      if (D[k2] > T(0) && n2 * den < num * D[k2]) { num = n2; den = D[k2]; }
    }
    a = num * od_rcp(den);
  }
  if constexpr (M::NSOC > 0) a = soc_step_cone<M, 0>(z, D, tau_soc, a);
  return a;
}

// CVXOPT sec. 5.1.3 centering: mu = <primal,dual>/ncones ; sigma = clamp(mu_aff/mu, 0, 1)^3
template <class M, class T> OD_HD T centering_kappa(const T* z, const T* Da, T aaff) {
  constexpr int n = M::NORT + M::NSOC;
  T s = T(0), sa = T(0);
  if constexpr (M::NORT > 0) {
#pragma unroll
    for (int i = 0; i < M::NORT; ++i) {
      const int a = M::ORT1[i], b = M::ORT2[i];
      s += z[a] * z[b];
      sa += (z[a] - aaff * Da[a]) * (z[b] - aaff * Da[b]);
    }
  }
  if constexpr (M::NSOC > 0) {
#pragma unroll
    for (int k = 0; k < M::SOCOFF[M::NSOC]; ++k) {
      const int a = M::SOC1[k], b = M::SOC2[k];
      s += z[a] * z[b];
      sa += (z[a] - aaff * Da[a]) * (z[b] - aaff * Da[b]);
    }
  }
  const T mu = s * T(1.0 / n);
  T q = sa * od_rcp(s);
  q = od_max(q, T(0));
  q = od_min(q, T(1));
  return q * q * q * mu;
}

template <class M, int C, class T> OD_HD void correction_cone(T* r, const T* Da) {
  constexpr int o = M::SOCOFF[C], n = M::SOCOFF[C + 1] - M::SOCOFF[C];
  T dot = T(0);
#pragma unroll
  for (int i = 0; i < n; ++i) dot += Da[M::SOC1[o + i]] * Da[M::SOC2[o + i]];
  r[M::SOCR[o]] += dot;
#pragma unroll
  for (int i = 1; i < n; ++i)
    r[M::SOCR[o + i]] += Da[M::SOC1[o]] * Da[M::SOC2[o + i]] + Da[M::SOC2[o]] * Da[M::SOC1[o + i]];
  if constexpr (C + 1 < M::NSOC) correction_cone<M, C + 1>(r, Da);
}

// rz evaluated with the orthant variables clamped from below at reg (regularisation of
// rz!(ip, rz, z, theta; reg)), then factored.
// PIV = false: the tail is factored down its diagonal (models with M::STATIC_TAIL, interior-point iterations only)
// models with a hand-written closed-form elimination (od_rocket_proj_direct.h) declare `static constexpr bool DIRECT_FACTOR = true`
template <class M, class = void> struct model_direct_factor : std::false_type {};
template <class M> struct model_direct_factor<M, std::void_t<decltype(M::DIRECT_FACTOR)>> : std::bool_constant<M::DIRECT_FACTOR> {};

template <class M, bool PIV = true, class T, class F>
OD_HD bool eval_factor(const T* z, const T* th, const T* pre, const T* tr, T reg, F& f) {
  T zr[M::NZ];
#pragma unroll
  for (int i = 0; i < M::NZ; ++i) zr[i] = z[i];
  if constexpr (M::NORT > 0) {
#pragma unroll
    for (int i = 0; i < M::NORT; ++i) {
      zr[M::ORT1[i]] = od_max(zr[M::ORT1[i]], reg);
      zr[M::ORT2[i]] = od_max(zr[M::ORT2[i]], reg);
    }
  }
  if constexpr (model_direct_factor<M>::value) {
    return M::template direct_factor<PIV>(zr, f);
  } else {
    T a[M::NNZ];
    M::eval_rz(zr, th, pre, tr, a);
    return M::template factor<PIV>(a, f);
  }
}

// backtracking on z - alpha D until either violation does not increase (at most max_ls trials, the last one is kept
// regardless); leaves z at the accepted point, r = r(z; 0) and its violations.
// Models up to the hopper's size, when lanes carry copies (Opts::coop): a solve that jams spends most of its time here (acrobot at its joint
// limit: ~15 trials in each of its 100 iterations), so after two sequential trials the 16/ppw copies each try a step
// size, agree on the first accepted one (the same one the sequential loop would find) and re-evaluate it.
template <class M, class = void> struct model_no_lane_copies : std::false_type {};
template <class M> struct model_no_lane_copies<M, std::void_t<decltype(M::DIRECT_STEP)>> : std::true_type {};   // (the rocket kernels map one problem to one lane)
template <class M> constexpr bool parallel_line_search() { return M::NZ <= 20 && !model_no_lane_copies<M>::value; }

// models whose full step to an orthant boundary is completed exactly (od_rocket_proj_direct.h) declare `static constexpr bool SNAP_BLOCKING = true`
template <class M, class = void> struct model_snap_blocking : std::false_type {};
template <class M> struct model_snap_blocking<M, std::void_t<decltype(M::SNAP_BLOCKING)>> : std::bool_constant<M::SNAP_BLOCKING> {};

// models whose equality rows are linear in z declare `static constexpr bool LINEAR_EQ_ROWS = true`
template <class M, class = void> struct model_linear_eq : std::false_type {};
template <class M> struct model_linear_eq<M, std::void_t<decltype(M::LINEAR_EQ_ROWS)>> : std::bool_constant<M::LINEAR_EQ_ROWS> {};

// snap: index of the orthant variable that blocks the step at tau = 1 (the trial at that very step length only), or -1
template <class M, class T>
OD_HD bool ls_trial(const T* th, const T* pre, T* tr, const T* z, const T* D, T alpha, T r_vio, T k_vio, T* zc, T* r, T& r_c, T& k_c, int snap = -1) {
#pragma unroll
  for (int i = 0; i < M::NZ; ++i) zc[i] = z[i] - alpha * D[i];
  if constexpr (model_snap_blocking<M>::value) {
#pragma unroll
    for (int i = 0; i < M::NSNAP; ++i) zc[M::SNAP[i]] = (M::SNAP[i] == snap) ? T(0) : zc[M::SNAP[i]];
  }
  M::eval_r(zc, th, pre, tr, r);
  r_c = viol_eq<M>(r);
  k_c = viol_bil<M>(r);
  // Equality rows that are LINEAR in z (M::LINEAR_EQ_ROWS): r_eq(z - alpha D) = (1 - alpha) r_eq(z) for the Newton direction D, so the
  // first disjunct holds in exact arithmetic for every alpha in [0, 1] and the first trial is the accepted one.  Evaluated in floating
  // point, from the first full step on both sides of `r_c <= r_vio` are the rounding noise of a residual that is exactly zero, and the
  // test -- whenever the complementarity violation rises, which it does next to the apex of the cone -- is a coin toss that no two
  // implementations (this one, the oracle, the reference on another BLAS) throw alike.  The test is evaluated as exact arithmetic would.
  if constexpr (model_linear_eq<M>::value) return true;
  return r_c <= r_vio || k_c <= k_vio;
}

template <class M, class T>
OD_HD void line_search(const Opts<T>& o, const T* th, const T* pre, T* tr, T* z, const T* D, T& alpha, T* r, T& r_vio, T& k_vio, int snap = -1) {
  T zc[M::NZ];
#pragma unroll
  for (int i = 0; i < M::NZ; ++i) zc[i] = z[i];        // (max_ls < 1 is rejected at the API; never copy garbage back)
  T r_c = r_vio, k_c = k_vio;
  const bool par = parallel_line_search<M>() && o.coop != 0;
  const int nseq = par ? od_min(2, o.max_ls) : o.max_ls;
  bool done = false;
  int ls = 0;
  for (; ls < nseq; ++ls) {
    if (ls_trial<M>(th, pre, tr, z, D, alpha, r_vio, k_vio, zc, r, r_c, k_c, ls == 0 ? snap : -1)) { done = true; break; }
    if (ls + 1 < o.max_ls) alpha *= T(0.5);            // alpha stays at the last trial's value if all of them fail
  }
  if constexpr (parallel_line_search<M>()) {
    if (par && !done && ls < o.max_ls) {
      // alpha is the step of trial `ls`; the G = 16/ppw copies try trials j0 .. j0 + G - 1
      const int G = 16 / o.coop, g = coop_copy(o.coop);
      bool found = false;
      for (int j0 = ls; j0 < o.max_ls && !found; j0 += G) {
        const T aj = od_ldexp(alpha, -g);
        const bool acc = ls_trial<M>(th, pre, tr, z, D, aj, r_vio, k_vio, zc, r, r_c, k_c) && (j0 + g < o.max_ls);
        const int m = coop_or_all(acc ? (1 << g) : 0, o.coop);
        if (m != 0) {
          alpha = od_ldexp(alpha, -__builtin_ctz(m));                        // first accepted trial of the round
          found = true;
        } else {
          const int left = o.max_ls - 1 - j0;                                // trials after j0: move on by G, or to the last
          alpha = od_ldexp(alpha, -(left < G ? left : G));
          if (left < G) break;                                               // alpha is now the last trial's step
        }
      }
      ls_trial<M>(th, pre, tr, z, D, alpha, r_vio, k_vio, zc, r, r_c, k_c);   // every copy lands on the chosen trial
    }
  }
#pragma unroll
  for (int i = 0; i < M::NZ; ++i) z[i] = zc[i];
  r_vio = r_c;
  k_vio = k_c;
}

// One predictor-corrector iteration at z (r = r(z; 0), r_vio / k_vio its violations): factor, affine
// direction, centering, corrector direction, step length, backtracking line search.  Shared by the
// lockstep loop below and the decoupled rollout (od_units.h) so that both do identical arithmetic.
// models with hand-written step lengths (od_rocket_proj_direct.h) declare `static constexpr bool DIRECT_STEP = true`
template <class M, class = void> struct model_direct_step : std::false_type {};
template <class M> struct model_direct_step<M, std::void_t<decltype(M::DIRECT_STEP)>> : std::bool_constant<M::DIRECT_STEP> {};

template <class M, class T, class F>
OD_HD void ip_iteration(const Opts<T>& o, const T* th, const T* pre, T* tr, T* z, T* r, T& r_vio, T& k_vio, T& reg_prev,
                        int& status, int it, F& f, T* alpha_out = nullptr) {
  constexpr bool CONES = (M::NORT + M::NSOC) > 0;
  const T reg = (k_vio < o.kappa_reg) ? k_vio * o.gamma_reg : T(0);
  reg_prev = reg;
  constexpr bool PIV = !M::STATIC_TAIL;
  if (!eval_factor<M, PIV>(z, th, pre, tr, reg, f)) status &= ~OD_ST_FACTOR_OK;
  T D[M::NZ];
  M::template solve<PIV>(f, r, D);                     // affine (predictor) direction
  int blk = -1;                                         // (SNAP_BLOCKING models: the orthant variable that sets the last step length computed)
  auto steplen = [&](const T* D_, T tau_ort, T tau_soc, const auto& pre_) {
    if constexpr (model_snap_blocking<M>::value) return M::template direct_step_length<T>(pre_, z, D_, tau_ort, tau_soc, &blk);
    else if constexpr (model_direct_step<M>::value) return M::template direct_step_length<T>(pre_, z, D_, tau_ort, tau_soc);
    else return step_length<M>(z, D_, tau_ort, tau_soc, o.coop);
  };
  auto make_pre = [&]() {
    if constexpr (model_direct_step<M>::value) return M::template step_pre<T>(z);
    else return 0;
  };
  const auto spre = make_pre();                        // (what the two step lengths of an iteration share)
  if constexpr (CONES) {
    const T aaff = steplen(D, T(1), T(1), spre);
    T kap = centering_kappa<M>(z, D, aaff);
    kap = od_max(kap, o.kappa_eval * o.undercut_inv);
#pragma unroll
    for (int i = 0; i < M::NKROWS; ++i) r[M::KROWS[i]] -= kap;    // r(z; kappa) from r(z; 0)
    if constexpr (M::NORT > 0) {
#pragma unroll
      for (int i = 0; i < M::NORT; ++i) r[M::ORTR[i]] += D[M::ORT1[i]] * D[M::ORT2[i]];
    }
    if constexpr (M::NSOC > 0) correction_cone<M, 0>(r, D);
    M::template solve<PIV>(f, r, D);                   // corrector direction, factors reused
  }
  const T vio = od_max(r_vio, k_vio);
  const T tau = T(1) - od_min(o.eps_min, vio * vio);
  T alpha = steplen(D, tau, od_min(tau, T(0.99)), spre);
#ifdef OD_TRACE
  const T alpha0_ = alpha;
#endif
  if constexpr (model_snap_blocking<M>::value) {
    if (!(tau == T(1))) blk = -1;                      // (only a step that goes ALL the way to the boundary ends on it)
    line_search<M>(o, th, pre, tr, z, D, alpha, r, r_vio, k_vio, blk);
  } else {
    line_search<M>(o, th, pre, tr, z, D, alpha, r, r_vio, k_vio);
  }
  if (alpha_out) *alpha_out = alpha;
  OD_TRACE_IT("dev it %d alpha %.17g r_vio %.6e k_vio %.6e alpha0 %.6e\n", it + 1, (double)alpha, (double)r_vio, (double)k_vio, (double)alpha0_);
#ifdef OD_TRACE
  if (od_trace_flag > 1) {          // level 2: the iterate as well
    std::printf("dev z%d", it + 1);
    for (int k = 0; k < M::NZ; ++k) std::printf(" %.17g", (double)z[k]);
    std::printf("\n");
  }
#endif
}

// sinks that want every row of dz (not only the solution block ZQ) declare `static constexpr bool ALL_ROWS = true`
template <class S, class = void> struct sink_all_rows : std::false_type {};
template <class S> struct sink_all_rows<S, std::void_t<decltype(S::ALL_ROWS)>> : std::bool_constant<S::ALL_ROWS> {};

// Stall exit (models that declare STALL_ALPHA / STALL_ITERS: the thrust-cone projection, od_rocket_proj_direct.h).  With eps_min = 0
// the projection's iterates can run into the boundary of the cone away from the solution (~0.02 % of random controls, 0.2 % of
// the controls of config 5's first iterations, in the oracle alike): the step to the boundary shrinks 100x per iteration, the
// accepted step length ends at ~1e-13 -- the guards of the cone step keep it from being exactly zero -- and the solve creeps through
// its remaining ~85 iterations moving the iterate by less than 1e-11 in total, to be reported as not converged.  A lockstep
// wavefront waits for it.  When the accepted step length has been below STALL_ALPHA for STALL_ITERS consecutive iterations the
// solve is abandoned as if max_iter had been reached (same status bits, iteration count = max_iter, the iterate within
// (max_iter - it) * STALL_ALPHA * |D| of the one the full loop would return).  od_set_projection_stall_exit(h, 0) runs every iteration.
template <class M, class = void> struct model_stall : std::false_type {};
template <class M> struct model_stall<M, std::void_t<decltype(M::STALL_ITERS)>> : std::true_type {};

// Sink concept:  void grad(int i /*row in ZQ*/, int c /*grad column*/, T v)
//
// z: in = initial guess, out = iterate at (r_tol, kappa_eval) convergence (or the last iterate).
// Returns status bits; iters[0] = iterations to kappa_eval, iters[1] = iterations to kappa_grad.
template <class M, class T, class Sink, class F>
OD_HD int ip_step_grad(const Opts<T>& o, const T* th, T* z, bool want_state, bool want_grad, Sink& sink, int* iters, F& f, bool stall_exit = false) {
  // state snapshot at (r_tol, kappa_eval): the whole z for raw solves, only the next configuration otherwise
  constexpr int NSNAP = Sink::FULL_STATE ? M::NZ : M::NZQ;
  T r[M::NZ], zs[NSNAP], pre[M::NPRE], tr[M::NTR];
  M::eval_pre(th, pre);
  M::eval_r(z, th, pre, tr, r);
  T r_vio = viol_eq<M>(r), k_vio = viol_bil<M>(r);
  bool eval_done = !want_state, grad_done = !want_grad;
  int status = OD_ST_FACTOR_OK;
  T reg_prev = T(0);
  iters[0] = iters[1] = 0;
  int it = 0;
  int nstall = 0;
  for (;; ++it) {
    const bool req = r_vio < o.r_tol;
    const bool last = it >= o.max_iter;
    if (!grad_done && ((req && k_vio < o.kappa_grad) || last)) {
      // differentiate_solution!: dz = -rz(z*)^{-1} rtheta(z*), reg = max(reg_val, kappa_tol*gamma_reg)
      const T reg = od_max(reg_prev, o.kappa_grad * o.gamma_reg);
      if constexpr (Sink::DEFER_GRAD) {
        // the gradient is computed later by a separate, fully parallel pass (gradient_at): only
        // record where -- keeps rtheta / the right-hand sides out of this loop's register budget
        sink.defer(z, reg);
      } else {
      if (!eval_factor<M>(z, th, pre, tr, reg, f)) status &= ~OD_ST_FACTOR_OK;
      T g[M::NNZTH];
      M::eval_rth(z, th, pre, tr, g);
      for (int c = 0; c < M::NGC; ++c) {
        T b[M::NZ];
#pragma unroll
        for (int i = 0; i < M::NZ; ++i) b[i] = T(0);
#pragma unroll
        for (int k = 0; k < M::NNZTH; ++k) b[M::RTH_ROW[k]] = (M::RTH_COL[k] == c) ? g[k] : b[M::RTH_ROW[k]];
        M::solve(f, b, b);
        if constexpr (sink_all_rows<Sink>::value) {
#pragma unroll
          for (int i = 0; i < M::NZ; ++i) sink.grad(i, c, -b[i]);
        } else {
#pragma unroll
          for (int i = 0; i < M::NZQ; ++i) sink.grad(i, c, -b[M::ZQ[i]]);
        }
      }
      }
      grad_done = true;
      iters[1] = it;
      if (!last) status |= OD_ST_GRAD_OK;
    }
    if (!eval_done && ((req && k_vio < o.kappa_eval) || last)) {
#pragma unroll
      for (int i = 0; i < NSNAP; ++i) zs[i] = Sink::FULL_STATE ? z[i] : z[M::ZQ[i]];
      eval_done = true;
      iters[0] = it;
      if (!last) status |= OD_ST_EVAL_OK;
    }
    if (eval_done && grad_done) break;

    if constexpr (model_stall<M>::value) {
      T alpha;
      ip_iteration<M>(o, th, pre, tr, z, r, r_vio, k_vio, reg_prev, status, it, f, &alpha);
      nstall = (alpha < T(sizeof(T) == 4 ? M::STALL_ALPHA_F32 : M::STALL_ALPHA)) ? nstall + 1 : 0;
      if (stall_exit && nstall >= M::STALL_ITERS && it + 1 < o.max_iter) it = o.max_iter - 1;
    } else {
      ip_iteration<M>(o, th, pre, tr, z, r, r_vio, k_vio, reg_prev, status, it, f);
    }
  }
  if (want_state) {
#pragma unroll
    for (int i = 0; i < NSNAP; ++i) z[Sink::FULL_STATE ? i : M::ZQ[i]] = zs[i];
  }
  return status;
}

// Implicit gradient at a recorded iterate (second pass of the split rollout): dz = -rz(z)^{-1} rtheta(z)
// with the orthant clamp `reg` that differentiate_solution! used.  Returns false if a pivot vanished.
template <class M, class T, class Sink> OD_HD bool gradient_at(const T* th, const T* z, T reg, Sink& sink) {
  T pre[M::NPRE], tr[M::NTR], r[M::NZ];
  M::eval_pre(th, pre);
  M::eval_r(z, th, pre, tr, r);          // produces the shared trigonometric values at z
  typename M::template Fact<T> f;
  const bool ok = eval_factor<M>(z, th, pre, tr, reg, f);
  T g[M::NNZTH];
  M::eval_rth(z, th, pre, tr, g);
  for (int c = 0; c < M::NGC; ++c) {
    T b[M::NZ];
#pragma unroll
    for (int i = 0; i < M::NZ; ++i) b[i] = T(0);
#pragma unroll
    for (int k = 0; k < M::NNZTH; ++k) b[M::RTH_ROW[k]] = (M::RTH_COL[k] == c) ? g[k] : b[M::RTH_ROW[k]];
    M::solve(f, b, b);
#pragma unroll
    for (int i = 0; i < M::NZQ; ++i) sink.grad(i, c, -b[M::ZQ[i]]);
  }
  return ok;
}

// convenience: factor storage in registers
template <class M, class T, class Sink>
OD_HD int ip_step_grad(const Opts<T>& o, const T* th, T* z, bool want_state, bool want_grad, Sink& sink, int* iters, bool stall_exit = false) {
  typename M::template Fact<T> f;
  return ip_step_grad<M, T, Sink>(o, th, z, want_state, want_grad, sink, iters, f, stall_exit);
}

// theta = [q2 - h*v1 ; q2 ; u ; friction ; h] and z0 = initialize_z!(q2) for the mechanical models
// (RoboDojo.step! as called from src/dynamics.jl:82-88); v1 = (q2 - q1)/h.
template <class M, class T> OD_HD void mech_setup(const T* q1, const T* q2, const T* u, const T* fric, T h, T* th, T* z) {
  const T hinv = od_rcp(h);
#pragma unroll
  for (int i = 0; i < M::NQ; ++i) {
    const T v1 = (q2[i] - q1[i]) * hinv;
    th[i] = q2[i] - h * v1;
    th[M::NQ + i] = q2[i];
  }
#pragma unroll
  for (int i = 0; i < M::NU; ++i) th[2 * M::NQ + i] = u[i];
  if constexpr (M::NFRIC > 0) {
#pragma unroll
    for (int i = 0; i < M::NFRIC; ++i) th[2 * M::NQ + M::NU + i] = fric[i];
  }
  th[2 * M::NQ + M::NU + M::NFRIC] = h;
#pragma unroll
  for (int i = 0; i < M::NZ; ++i) z[i] = M::ZI_KIND[i] == 0 ? q2[M::ZI_IDX[i]] : T(M::ZI_VAL[i]);
}

}  // namespace od
