// Scalar helpers shared by the generated model code and the interior-point solver.
// Device code for gfx950 (hipcc); the same header also compiles with a host C++ compiler so that
// tests/ can exercise the solver logic on CPU against the oracle (test harness only -- the shipped
// library has no CPU path).
#pragma once
#include <cmath>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define OD_HD __host__ __device__ __forceinline__
#else
#define OD_HD inline
#endif

// floor for static pivots on orthant slack variables (strictly positive, may underflow at convergence)
#define OD_PIVOT_FLOOR 1e-12
// largest dense tail that uses the branchy / guarded pivoting code (od_lu_factor)
#ifndef OD_LU_BRANCHY_MAX
#define OD_LU_BRANCHY_MAX 8
#endif

namespace od {

// fused multiply-add with one rounding on both the device and the host test build
OD_HD double od_fma(double a, double b, double c) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_fma(a, b, c);
#else
  return std::fma(a, b, c);
#endif
}

// a * b + c with c a polynomial coefficient: the three-operand instruction, stated.  Left to itself hipcc 7.2 keeps the
// coefficients in registers across the solve loop and evaluates a Horner step as a copy of the coefficient plus the two-operand
// v_fmac_f64 -- two issue slots where one does (the 10 steps of od_sincos are on every line-search trial).  Same rounding.
OD_HD double od_fma_coef(double a, double b, double c) {
#if defined(__HIP_DEVICE_COMPILE__)
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
#else
  return std::fma(a, b, c);
#endif
}
// (the first step: both coefficients are constants, the multiplier comes from a scalar register)
OD_HD double od_fma_coef2(double a, double kb, double c) {
#if defined(__HIP_DEVICE_COMPILE__)
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(kb), "v"(c));
  return d;
#else
  return std::fma(a, kb, c);
#endif
}

// sin and cos of a joint angle.  The library routines carry a Payne-Hanek path for huge arguments
// and cost ~200 instructions each on gfx950; angles on this path are O(1..1e3) rad, so: three-term
// Cody-Waite reduction by pi/2 with FMAs (the reduced argument is good to ~1 ulp for |x| < 2^30 and
// degrades gracefully beyond), then the classic degree-13/14 minimax kernels on [-pi/4, pi/4].
// Measured <= 1 ulp against a 200-bit reference for |x| <= 1e6 (tests/test_models.py).  od_sin(x) and
// od_cos(x) of the same x share everything but the last selects once inlined.
OD_HD void od_sincos(double x, double& sn, double& cs) {
  const double n = __builtin_rint(x * 6.36619772367581382433e-01);
  double r = od_fma(n, -1.5707963267948966, x);            // fl(pi/2)
  r = od_fma(n, -6.123233995736766e-17, r);                // next 53 bits
  r = od_fma(n, 1.4973849048591698e-33, r);                // and the next
  const double z = r * r;
  // sin(r) = r + r^3 (S1 + z S2 + ... + z^5 S6)
  double ps = od_fma_coef2(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
  ps = od_fma_coef(z, ps, 2.75573137070700676789e-06);
  ps = od_fma_coef(z, ps, -1.98412698298579493134e-04);
  ps = od_fma_coef(z, ps, 8.33333333332248946124e-03);
  ps = od_fma_coef(z, ps, -1.66666666666666324348e-01);
  const double s = od_fma(z * r, ps, r);
  // cos(r) = 1 - z/2 + z^2 (C1 + z C2 + ... + z^5 C6), summed so that the rounding of 1 - z/2 is recovered
  double pc = od_fma_coef2(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
  pc = od_fma_coef(z, pc, -2.75573143513906633035e-07);
  pc = od_fma_coef(z, pc, 2.48015872894767294178e-05);
  pc = od_fma_coef(z, pc, -1.38888888888741095749e-03);
  pc = od_fma_coef(z, pc, 4.16666666666666019037e-02);
  const double hz = 0.5 * z, w = 1.0 - hz;
  const double c = w + (((1.0 - w) - hz) + z * z * pc);
  const int q = (int)n;                                    // quadrant (saturates, harmlessly, for |x| > 3e9)
  const bool swp = q & 1;
  const double ss = swp ? c : s, cc = swp ? s : c;
  sn = (q & 2) ? -ss : ss;
  cs = ((q + 1) & 2) ? -cc : cc;
}
OD_HD double od_sin(double x) { double s, c; od_sincos(x, s, c); return s; }
OD_HD double od_cos(double x) { double s, c; od_sincos(x, s, c); return c; }
OD_HD float od_sin(float x);
OD_HD float od_cos(float x);
// how the generated residuals evaluate the sines and cosines of their N z-dependent angles: one after the other here;
// the cooperative solver substitutes a policy that computes two angles side by side (od_coop.h::TrigHalves)
struct TrigDirect {
  template <int N, class T> OD_HD static void sincos_n(const T* a, T* s, T* c) {
#pragma unroll
    for (int i = 0; i < N; ++i) { s[i] = od_sin(a[i]); c[i] = od_cos(a[i]); }
  }
};
OD_HD double od_sqrt(double x) { return sqrt(x); }
OD_HD double od_abs(double x) { return fabs(x); }
OD_HD double od_pow(double x, double y) { return pow(x, y); }
OD_HD float od_sin(float x) { return sinf(x); }
OD_HD float od_cos(float x) { return cosf(x); }
OD_HD float od_sqrt(float x) { return sqrtf(x); }
OD_HD float od_abs(float x) { return fabsf(x); }
OD_HD float od_pow(float x, float y) { return powf(x, y); }

// reciprocal: hardware seed (v_rcp_f64: 24 bits, tools/ubench/rcp_accuracy.hip) + one third-order step
// r (1 + e + e^2), e = 1 - x r, on the device -- within 0.5 ulp, and bit for bit what two or three Newton steps give
// (1 M random operands, profiles/r2_ubench_rcp_accuracy.txt) in half their dependent FMAs.  No denormal / inf
// special-casing: operands here are pivots, norms and step denominators.  Plain division on the host.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(OD_EXACT_RCP)
OD_HD double od_rcp(double x) {
  const double r = __builtin_amdgcn_rcp(x);
  const double e = __builtin_fma(-x, r, 1.0);
  return __builtin_fma(r, __builtin_fma(e, e, e), r);
}
OD_HD float od_rcp(float x) {
  float r = __builtin_amdgcn_rcpf(x);
  r = __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r);
  return r;
}
#elif defined(OD_EMULATE_RCP)   // test harness: mimic the device sequence (24-bit seed + 2 Newton steps)
OD_HD double od_rcp(double x) {
  double r = (double)(float)(1.0 / x);
  r = std::fma(std::fma(-x, r, 1.0), r, r);
  r = std::fma(std::fma(-x, r, 1.0), r, r);
  return r;
}
OD_HD float od_rcp(float x) { return 1.0f / x; }
#else
OD_HD double od_rcp(double x) { return 1.0 / x; }
OD_HD float od_rcp(float x) { return 1.0f / x; }
#endif

// reciprocal square root (seed + Newton steps on the device)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(OD_EXACT_RCP)
OD_HD double od_rsqrt(double x) {
  // v_rsq_f64 seed: 24 bits; two Newton steps: < 1 ulp (a third changes the last bit on 11 % of operands, not the bound)
  double y = __builtin_amdgcn_rsq(x);
  const double nhx = -0.5 * x;
#pragma unroll
  for (int i = 0; i < 2; ++i) y = __builtin_fma(__builtin_fma(nhx * y, y, 0.5), y, y);
  return y;
}
OD_HD float od_rsqrt(float x) {
  float y = __builtin_amdgcn_rsqf(x);
  return __builtin_fmaf(__builtin_fmaf(-0.5f * x * y, y, 0.5f), y, y);
}
#elif defined(OD_EMULATE_RCP)   // test harness: mimic the device sequence (24-bit seed + 2 Newton steps)
OD_HD double od_rsqrt(double x) {
  double y = (double)(float)(1.0 / sqrt(x));
  const double nhx = -0.5 * x;
  for (int i = 0; i < 2; ++i) y = std::fma(std::fma(nhx * y, y, 0.5), y, y);
  return y;
}
OD_HD float od_rsqrt(float x) { return 1.0f / sqrtf(x); }
#else
OD_HD double od_rsqrt(double x) { return 1.0 / sqrt(x); }
OD_HD float od_rsqrt(float x) { return 1.0f / sqrtf(x); }
#endif

// x^(-1/Q) for x > 0, Q >= 2 known at code-generation time (codegen/emit.py::rewrite_roots: the fractional powers of one base
// become integer powers of this root).  Device: seed 2^(-log2(x)/Q) from the single-precision transcendental units
// (v_log_f32 / v_exp_f32, ~2e-7 relative for the operands met here), then division-free Newton steps on
// f(u) = u^-Q - x:  u <- u + u (1 - x u^Q) / Q,  error e -> (Q + 1)/2 e^2: two steps reach the rounding floor (the
// residual 1 - x u^Q is one FMA).  Operands outside the range of the single-precision seed take the library's pow.
template <int N, class T> OD_HD T od_powi(T x);
#if defined(__HIP_DEVICE_COMPILE__) && !defined(OD_EXACT_RCP)
template <int Q> __device__ __forceinline__ double od_rootinv(double x) {
  if constexpr (Q == 2) {
    // (x = 0 and x = inf would turn the Newton steps into 0 * inf = NaN where 1 / sqrt(x) is +inf and 0: same answers as the host
    // build on degenerate geometry -- a planar-push distance that vanishes exactly)
    if (!(x > 1e-300 && x < 1e300)) return 1.0 / sqrt(x);
    double y = __builtin_amdgcn_rsq(x);
    const double nhx = -0.5 * x;
#pragma unroll
    for (int i = 0; i < 2; ++i) y = __builtin_fma(__builtin_fma(nhx * y, y, 0.5), y, y);
    return y;
  } else {
    const float xf = (float)x;
    if (!(xf > 1e-30f && xf < 1e30f)) return pow(x, -1.0 / Q);
    double u = (double)__builtin_amdgcn_exp2f(__builtin_amdgcn_logf(xf) * (-1.0f / Q));
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const double e = __builtin_fma(-x, od_powi<Q>(u), 1.0);
      u = __builtin_fma(u * (1.0 / Q), e, u);
    }
    return u;
  }
}
template <int Q> __device__ __forceinline__ float od_rootinv(float x) { return powf(x, -1.0f / Q); }
#else
template <int Q> OD_HD double od_rootinv(double x) { return Q == 2 ? 1.0 / sqrt(x) : pow(x, -1.0 / Q); }
template <int Q> OD_HD float od_rootinv(float x) { return Q == 2 ? 1.0f / sqrtf(x) : powf(x, -1.0f / Q); }
#endif

// true if the predicate holds in any active lane of the wavefront (a scalar branch on the device)
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ bool od_any_lane(bool pred) { return __builtin_amdgcn_ballot_w64(pred) != 0; }
#else
inline bool od_any_lane(bool pred) { return pred; }
#endif

// x * 2^e, exact
OD_HD double od_ldexp(double x, int e) { return __builtin_ldexp(x, e); }
OD_HD float od_ldexp(float x, int e) { return __builtin_ldexpf(x, e); }

template <class T> OD_HD T od_min(T a, T b) { return a < b ? a : b; }
template <class T> OD_HD T od_max(T a, T b) { return a > b ? a : b; }
// IEEE maxNum (one instruction on the device; a NaN operand is dropped)
OD_HD double od_fmax(double a, double b) { return __builtin_fmax(a, b); }
OD_HD float od_fmax(float a, float b) { return __builtin_fmaxf(a, b); }
OD_HD double od_fmin(double a, double b) { return __builtin_fmin(a, b); }
OD_HD float od_fmin(float a, float b) { return __builtin_fminf(a, b); }

// x^N by repeated squaring (N known at code-generation time)
template <int N, class T> OD_HD T od_powi(T x) {
  if constexpr (N == 0) return T(1);
  else if constexpr (N == 1) return x;
  else if constexpr (N % 2 == 0) { const T h = od_powi<N / 2>(x); return h * h; }
  else return x * od_powi<N - 1>(x);
}

// ---------------------------------------------------------------------------------------------
// Dense LU with partial pivoting for the small "tail" block left after the static elimination
// (N = nq for the mechanical models).  Fully unrolled; row exchanges are done with selects so the
// N*N block stays in registers (no dynamically indexed private array -> no scratch).
// A is column-major N x N, overwritten by L\U of P*A with the diagonal of U stored inverted.
// ---------------------------------------------------------------------------------------------
template <class T, int N> OD_HD bool od_lu_factor(T* A, int* piv) {
  bool ok = true;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    // partial pivoting, ties to the diagonal.  The generated tail puts the natural pivots on the
    // diagonal, and a whole exchange is skipped when no lane of the wavefront needs it.
    int p = k;
    T best = od_abs(A[k + N * k]);
#pragma unroll
    for (int i = k + 1; i < N; ++i) {
      const T v = od_abs(A[i + N * k]);
      if (v > best) { best = v; p = i; }
    }
    piv[k] = p;
    ok = ok && (best > T(0));
    // exchange rows k and p of the ACTIVE part only (columns >= k); the multipliers already stored
    // in columns < k stay with their physical rows, and od_lu_solve replays the exchanges in order.
    // Small tails skip an exchange no lane of the wavefront needs (a scalar branch); for the large ones
    // (planar push, 14 x 14, partly in scratch) the branch-free form is both faster and -- measured -- the
    // only one hipcc 7.2 compiles reliably: with either kind of branch here, a fraction of the planar-push
    // solves stopped converging depending on unrelated source details (a printf made them converge).
    bool exchange = true;
    if constexpr (N <= OD_LU_BRANCHY_MAX) exchange = od_any_lane(p != k);
    if (exchange) {
#pragma unroll
      for (int i = k + 1; i < N; ++i) {
        const bool sw = (p == i);
#pragma unroll
        for (int j = k; j < N; ++j) {
          const T u = A[k + N * j], w = A[i + N * j];
          A[k + N * j] = sw ? w : u;
          A[i + N * j] = sw ? u : w;
        }
      }
    }
    // a column that vanished entirely (a cone variable stepped exactly onto its boundary: tau rounds to
    // 1 once the violation is below 1e-8) makes the system singular; in the small tails its unknown is
    // dropped (x_k = 0) so that the outputs stay finite; the caller reports the knot through FACTOR_OK
    T inv;
    if constexpr (N > OD_LU_BRANCHY_MAX) inv = od_rcp(A[k + N * k]);
    else inv = best > T(0) ? od_rcp(A[k + N * k]) : T(0);
    A[k + N * k] = inv;                       // the diagonal holds 1/u_kk
#pragma unroll
    for (int i = k + 1; i < N; ++i) A[i + N * k] *= inv;
#pragma unroll
    for (int j = k + 1; j < N; ++j) {
      const T ukj = A[k + N * j];
#pragma unroll
      for (int i = k + 1; i < N; ++i) A[i + N * j] -= A[i + N * k] * ukj;
    }
  }
  return ok;
}

// The same factorisation with the pivots taken down the diagonal, in the generated order (no search, no exchange,
// nothing to replay in the solves).  For models whose tail needs no runtime pivoting in the interior-point
// iterations (M::STATIC_TAIL: measured per model against the pivoted factorisation -- the hopper's iterates are
// identical, its converged-point gradient solve is NOT and keeps od_lu_factor).
template <class T, int N> OD_HD bool od_lu_factor_static(T* A) {
  bool ok = true;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    ok = ok && (od_abs(A[k + N * k]) > T(0));
    const T inv = od_rcp(A[k + N * k]);
    A[k + N * k] = inv;
#pragma unroll
    for (int i = k + 1; i < N; ++i) A[i + N * k] *= inv;
#pragma unroll
    for (int j = k + 1; j < N; ++j) {
      const T ukj = A[k + N * j];
#pragma unroll
      for (int i = k + 1; i < N; ++i) A[i + N * j] -= A[i + N * k] * ukj;
    }
  }
  return ok;
}

// read-only view of the tail block inside a factor store (registers or any other storage with v[i])
template <class T, int BASE, class F> struct TailView {
  const F& f;
  OD_HD T operator[](int i) const { return f.v[BASE + i]; }
};
template <class T, int BASE, class F> OD_HD TailView<T, BASE, F> od_tail_view(const F& f) { return TailView<T, BASE, F>{f}; }

template <class T, int N, class Mat> OD_HD void od_lu_solve(const Mat& A, const int* piv, T* b) {
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const int p = piv[k];
#pragma unroll
    for (int i = k + 1; i < N; ++i) {          // exchange k of the factorisation ...
      const bool sw = (p == i);
      const T u = b[k], w = b[i];
      b[k] = sw ? w : u;
      b[i] = sw ? u : w;
    }
#pragma unroll
    for (int i = k + 1; i < N; ++i) b[i] -= A[i + N * k] * b[k];   // ... then its elimination step
  }
#pragma unroll
  for (int k = N - 1; k >= 0; --k) {
    b[k] *= A[k + N * k];
#pragma unroll
    for (int i = 0; i < k; ++i) b[i] -= A[i + N * k] * b[k];
  }
}

template <class T, int N, class Mat> OD_HD void od_lu_solve_static(const Mat& A, T* b) {
#pragma unroll
  for (int k = 0; k < N; ++k) {
#pragma unroll
    for (int i = k + 1; i < N; ++i) b[i] -= A[i + N * k] * b[k];
  }
#pragma unroll
  for (int k = N - 1; k >= 0; --k) {
    b[k] *= A[k + N * k];
#pragma unroll
    for (int i = 0; i < k; ++i) b[i] -= A[i + N * k] * b[k];
  }
}

}  // namespace od
