// The thrust-cone projection's KKT system in closed form (round 4).
//
// The reference factors the dense 10 x 10 Jacobian of the projection residual (src/models/rocket/codegen.jl:45-64) with partial
// pivoting in every interior-point iteration of soc_projection (src/models/rocket/dynamics.jl:168-186).  The generated
// elimination (gen/rocket_projection.h) takes the five structural -1 pivots statically and leaves a 5 x 5 tail with runtime
// partial pivoting -- selects for every row exchange, 78 % of a closed-loop rocket rollout.  The system
// has far more structure than that.  With  z = [u (3), p, s, w, y, v (3)],  a = u + v  and the right-hand side b:
//
//   rows 0-4 (pivots -1):   Dv01 = Du01 - b01,   Dv2 = Du2 - (Dy + Dp) - b2,   Ds = -Du2 - b3,   Dw = -Dy - b4
//   rows 8, 9 (cone tail):  a2 Du_i + a_i Du2 - u_i (Dy + Dp) = c_{8+i},  i = 0, 1      pivot a2 = u2 + v2 > 0 (u, v in the cone)
//   row 7 then:             E Du2 + G (Dy + Dp) = c7',   E = (a2^2 - a0^2 - a1^2) / a2 > 0,   G = (a0 u0 + a1 u1) / a2 - u2 <= 0
//   rows 5, 6 (orthants):   -w Du2 - s Dy = c5,    p Du2 + u2 Dp = b6
//
// The last three lines are a 3 x 3 system in (Du2, Dy, Dp) whose determinant  G (w u2 + s p) - s u2 E  is a sum of terms of
// one sign (E > 0 >= G; s, w, u2, p >= 0): it cannot cancel, vanishes only if the iterate is degenerate in both orthant pairs at
// once, and Cramer's rule needs no pivot choice at all -- in particular none of the divisions by s or u2 that break a static
// elimination when a full step (eps_min = 0, tau = 1) lands exactly on the boundary (codegen/models.py::rocket_projection).
// Same linear system, solved exactly in another order: iterates agree with the pivoted elimination to rounding, i.e. up to the
// line-search ties of this solve.
#pragma once
#include "gen/rocket_projection.h"

namespace od {

// sqrt(x) for the norm in the cone step: x * rsqrt(x) (the library's correctly rounded sqrt expands to ~28 instructions in single
// precision, ~20 in double; four of them per interior-point iteration); 0 for x = 0
template <class T> OD_HD T od_sqrt_fast(T x) {
#if (defined(__HIP_DEVICE_COMPILE__) && !defined(OD_EXACT_RCP)) || defined(OD_EMULATE_RCP)
  return x > T(0) ? x * od_rsqrt(x) : T(0);
#else
  return od_sqrt(x);
#endif
}

struct Model_rocket_projection_direct : Model_rocket_projection {
  static constexpr bool DIRECT_FACTOR = true;
  // stall exit (od_solver.h::model_stall): the accepted step length below STALL_ALPHA in STALL_ITERS consecutive iterations
  // (single precision: the guards of the cone step sit at 1e-7 and a stalled solve creeps with step lengths between 5e-7 and 7e-4:
  // of the 42 240 candidate controls of an iteration of config 5, 38 solves make more than 20 iterations; step length < 1e-5 x 4
  // abandons 14 of them, < 1e-3 x 4 abandons 29 -- every one that would run into max_iter -- and in both cases one that would have
  // converged late, tools/diag_config5_rollout.py)
  static constexpr double STALL_ALPHA = 1e-9, STALL_ALPHA_F32 = 1e-3;
  static constexpr int STALL_ITERS = 4;
  // step lengths of this model: hand-written (od_solver.h::model_direct_step) -- one 3-d cone, two orthant pairs
  static constexpr bool DIRECT_STEP = true;

  // The part of the CVXOPT sec. 8.2 cone step (od_solver.h::soc_step_one<3>) that depends on the cone variable lam alone: the
  // predictor and the corrector step length of an iteration share it.
  template <class T> struct ConePre { T l0, l1, l2, isq, ill, q; };
  template <class T> OD_HD static ConePre<T> cone_pre(T l0, T l1, T l2) {
    const T eps = sizeof(T) == 4 ? T(OD_SOC_EPS_F32) : T(1e-14);
    T ll = l0 * l0;
    ll -= l1 * l1;
    ll -= l2 * l2;
    ll = od_max(ll, sizeof(T) == 4 ? T(OD_SOC_FLOOR_F32) : T(1e-25)) + eps;
    ConePre<T> p;
    p.l0 = l0; p.l1 = l1; p.l2 = l2;
    p.isq = od_rsqrt(ll);
    p.ill = p.isq * p.isq;
    p.q = od_rcp(l0 * p.isq + T(1));
    return p;
  }
  // step to the boundary for lam - alpha * (D0, D1, D2)  (soc_step_one with dlt = -D)
  template <class T> OD_HD static T cone_step(const ConePre<T>& p, T D0, T D1, T D2, T tau) {
    const T eps = sizeof(T) == 4 ? T(OD_SOC_EPS_F32) : T(1e-14);
    const T d0 = -D0, d1 = -D1, d2 = -D2;
    T ld = p.l0 * d0;
    ld -= p.l1 * d1;
    ld -= p.l2 * d2;
    ld += eps;
    const T rs = ld * p.ill;
    const T c = (ld * p.isq + d0) * p.q;
    const T r1 = d1 * p.isq - c * p.l1 * p.ill, r2 = d2 * p.isq - c * p.l2 * p.ill;
    const T nv = od_sqrt_fast(r1 * r1 + r2 * r2);
    T a = T(1);
    if (nv - rs > T(0)) a = od_min(a, tau * od_rcp(nv - rs));
    return a;
  }
  template <class T> struct StepPre { ConePre<T> u, v; };
  template <class T> OD_HD static StepPre<T> step_pre(const T* z) {
    StepPre<T> p;
    p.u = cone_pre<T>(z[2], z[0], z[1]);       // SOC1 = {2, 0, 1}
    p.v = cone_pre<T>(z[9], z[7], z[8]);       // SOC2 = {9, 7, 8}
    return p;
  }
  // largest alpha in (0, 1] keeping z - alpha D inside the cones (od_solver.h::step_length for this model)
  // blk (optional): the orthant variable whose ratio test set the step (-1: a cone, or the full step)
  template <class T> OD_HD static T direct_step_length(const StepPre<T>& p, const T* z, const T* D, T tau_ort, T tau_soc, int* blk = nullptr) {
    // orthant pairs (s, w) and (u3, p): ORT1 = {4, 2}, ORT2 = {5, 3}; min of the ratio tests kept as a fraction
    T num = T(1), den = T(1);
    int kb = -1;
    constexpr int K[4] = {4, 5, 2, 3};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const T n1 = tau_ort * z[K[i]];
      if (D[K[i]] > T(0) && n1 * den < num * D[K[i]]) { num = n1; den = D[K[i]]; kb = K[i]; }
    }
    const T ao = num * od_rcp(den);
    T a = ao;
    a = od_min(a, cone_step<T>(p.u, D[2], D[0], D[1], tau_soc));
    a = od_min(a, cone_step<T>(p.v, D[9], D[7], D[8], tau_soc));
    if (blk) *blk = a < ao ? -1 : kb;
    return a;
  }
  // A full step to the boundary of an orthant (eps_min = 0: tau = 1, src/models/rocket/dynamics.jl:81) ends, in exact arithmetic, with
  // the blocking variable EXACTLY zero: z_k - (z_k / D_k) D_k.  In floating point it ends on the rounding residual of the division (a
  // fused multiply-add returns exactly that residual: +-1e-17, either sign), and the algorithm is discontinuous there -- the next
  // affine direction of a variable at zero is zero times something, the sign of its noise decides between "blocked at step length
  // 0" (sigma = 1) and "not blocking" (sigma ~ 0), and the iterates part at the 1e-3 level (tools/proj_paths_host.py).  The line search
  // sets the blocking variable of an accepted full step to its exact value (od_solver.h::ls_trial), and `solve` below returns the
  // exact component of a variable that sits at zero, so that the device follows the exact-arithmetic path through such a landing.
  static constexpr bool SNAP_BLOCKING = true;
  // rows 0-4 of the residual (src/models/rocket/codegen.jl:53-57) are linear in z: the line search accepts as exact arithmetic does
  // (od_solver.h::ls_trial)
  static constexpr bool LINEAR_EQ_ROWS = true;
  static constexpr int NSNAP = 4;
  static constexpr int SNAP[4] = {4, 5, 2, 3};
  // u0 u1 u2 | s w | a0 a1 1/a2 | inverse of the 3 x 3 block, row-major (9) | p
  template <class T> struct Fact { T v[18]; };

  // z: the iterate with its orthant variables clamped (eval_factor); the Jacobian does not depend on theta
  template <bool PIV = true, class T, class F> OD_HD static bool direct_factor(const T* z, F& f) {
    const T u0 = z[0], u1 = z[1], u2 = z[2], p = z[3], s = z[4], w = z[5], v0 = z[7], v1 = z[8], v2 = z[9];
    const T a0 = u0 + v0, a1 = u1 + v1, a2 = u2 + v2;
    const bool oka = a2 != T(0);
    const T ia2 = oka ? od_rcp(a2) : T(0);
    const T E = a2 - (a0 * a0 + a1 * a1) * ia2;
    const T G = (a0 * u0 + a1 * u1) * ia2 - u2;
    const T pg = p * G - u2 * E, wg = w * G - s * E;            // (each a sum of two terms of one sign)
    const T det = G * (w * u2 + s * p) - s * u2 * E;
    const bool okd = det != T(0) && det == det;
    const T id = okd ? od_rcp(det) : T(0);
    f.v[0] = u0; f.v[1] = u1; f.v[2] = u2; f.v[3] = s; f.v[4] = w; f.v[5] = a0; f.v[6] = a1; f.v[7] = ia2;
    // adj / det of  [-w -s 0; p 0 u2; E G G]
    f.v[8] = -u2 * G * id;  f.v[9] = s * G * id;   f.v[10] = -s * u2 * id;
    f.v[11] = -pg * id;     f.v[12] = -w * G * id; f.v[13] = w * u2 * id;
    f.v[14] = p * G * id;   f.v[15] = wg * id;     f.v[16] = s * p * id;
    f.v[17] = p;
    return oka && okd;
  }

  // x = J^{-1} b (x may alias b)
  template <bool PIV = true, class T, class F> OD_HD static void solve(const F& f, const T* b, T* x) {
    const T u0 = f.v[0], u1 = f.v[1], u2 = f.v[2], s = f.v[3], w = f.v[4], a0 = f.v[5], a1 = f.v[6], ia2 = f.v[7];
    const T b0 = b[0], b1 = b[1], b2 = b[2], b3 = b[3], b4 = b[4], b6 = b[6];
    const T c5 = b[5] + w * b3 + s * b4;
    const T c7 = b[7] + u0 * b0 + u1 * b1 + u2 * b2;
    const T c8 = b[8] + u2 * b0 + u0 * b2;
    const T c9 = b[9] + u2 * b1 + u1 * b2;
    const T c7p = c7 - (a0 * c8 + a1 * c9) * ia2;
    const T du2 = f.v[8] * c5 + f.v[9] * b6 + f.v[10] * c7p;
    const T dy = f.v[11] * c5 + f.v[12] * b6 + f.v[13] * c7p;
    const T dp = f.v[14] * c5 + f.v[15] * b6 + f.v[16] * c7p;
    const T q = dy + dp;
    const T du0 = (c8 - a0 * du2 + u0 * q) * ia2;
    const T du1 = (c9 - a1 * du2 + u1 * q) * ia2;
    T ds = -du2 - b3, dw = -dy - b4;
    // a variable of an orthant pair that sits EXACTLY at zero (the landing of a full step, SNAP_BLOCKING): its row of the system,
    // s Dw + w Ds = b5 or p Du2 + u2 Dp = b6, has one entry left and gives the component exactly -- zero for the affine direction,
    // -kappa / partner for the corrector -- where the elimination order above returns it as a difference of O(1) terms, i.e. as
    // rounding noise of either sign (a pivoted LU takes that row as it stands: same system, same solution, no noise)
    const T pp = f.v[17];
    T du2o = du2, dpo = dp;
    if (od_any_lane(s == T(0) || w == T(0) || u2 == T(0) || pp == T(0))) {
      if (w == T(0) && s != T(0)) dw = b[5] * od_rcp(s);
      if (s == T(0) && w != T(0)) ds = b[5] * od_rcp(w);
      if (pp == T(0) && u2 != T(0)) dpo = b6 * od_rcp(u2);
      if (u2 == T(0) && pp != T(0)) du2o = b6 * od_rcp(pp);
    }
    x[0] = du0; x[1] = du1; x[2] = du2o; x[3] = dpo;
    x[4] = ds;
    x[5] = dw;
    x[6] = dy;
    x[7] = du0 - b0; x[8] = du1 - b1; x[9] = du2 - q - b2;
  }
};

}  // namespace od
