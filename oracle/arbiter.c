/* TEST INFRASTRUCTURE -- extended-precision arbiter for the implicit gradient.
 *
 * The implicit-function gradient of the path (src/dynamics.jl:103-111,123-125: grad_sim step with diff_sol,
 * RoboDojo differentiate_solution!) is dz = -rz(z*; reg)^{-1} rtheta(z*).  At converged contact modes rz is numerically
 * singular (condition numbers ~1e18: inactive / sticking cones carry variables of size ~1e-23), so the oracle's dense
 * partial-pivot LU and the device's sparse elimination -- two correct double-precision solvers -- can disagree beyond
 * the 1e-4 bar.  This file decides who is right: given an iterate (z, reg, theta) it evaluates rz (with the orthant
 * clamp) and rtheta with the oracle's generated functions and solves in IEEE binary128 (__float128, 113-bit mantissa,
 * libquadmath) with partial pivoting and two steps of iterative refinement -- exact to double rounding for any
 * condition number below ~1e30.  It also returns the condition number ||A||_inf ||A^-1||_inf.
 * The tests call it with the device's recorded gradient iterate (od_get_grad_iterates) and with the oracle's own.
 * Included at the end of ip_oracle.c (same translation unit: it uses the model table). */
#include <quadmath.h>

typedef __float128 q128;

static int q_lu_factor(int n, q128* A, int* piv) {
  for (int k = 0; k < n; ++k) {
    int p = k;
    q128 best = fabsq(A[k + n * k]);
    for (int i = k + 1; i < n; ++i) {
      q128 v = fabsq(A[i + n * k]);
      if (v > best) { best = v; p = i; }
    }
    piv[k] = p;
    if (best == 0) return 0;
    if (p != k) for (int j = 0; j < n; ++j) { q128 t = A[k + n * j]; A[k + n * j] = A[p + n * j]; A[p + n * j] = t; }
    q128 inv = 1 / A[k + n * k];
    for (int i = k + 1; i < n; ++i) A[i + n * k] *= inv;
    for (int j = k + 1; j < n; ++j) {
      q128 u = A[k + n * j];
      if (u != 0) for (int i = k + 1; i < n; ++i) A[i + n * j] -= A[i + n * k] * u;
    }
  }
  return 1;
}

static void q_lu_solve(int n, const q128* A, const int* piv, q128* b) {
  /* whole rows were exchanged in the factorisation (LAPACK style): apply every interchange first, then L, then U */
  for (int k = 0; k < n; ++k) {
    int p = piv[k];
    if (p != k) { q128 t = b[k]; b[k] = b[p]; b[p] = t; }
  }
  for (int k = 0; k < n; ++k)
    for (int i = k + 1; i < n; ++i) b[i] -= A[i + n * k] * b[k];
  for (int k = n - 1; k >= 0; --k) {
    b[k] /= A[k + n * k];
    for (int i = 0; i < k; ++i) b[i] -= A[i + n * k] * b[k];
  }
}

/* dz (nz x nth col-major) = -rz(z; reg)^{-1} rtheta(z), solved in binary128; cond = ||A||_inf ||A^-1||_inf.
 * returns 1, or 0 if the matrix is exactly singular (dz, cond = NaN). */
int od_arbiter_gradient(int model_id, const double* z, const double* th, double reg, double* dz, double* cond) {
  const od_oracle_model* m = od_oracle_models[model_id];
  const int nz = m->nz, nth = m->nth;
  double rz[NZMAX * NZMAX], rth[NZMAX * NTHMAX];
  rz_reg(m, rz, z, th, reg);
  m->rth(z, th, rth);
  static __thread q128 A[NZMAX * NZMAX], LU[NZMAX * NZMAX], inv[NZMAX * NZMAX];
  q128 x[NZMAX], res[NZMAX];
  int piv[NZMAX];
  for (int i = 0; i < nz * nz; ++i) A[i] = LU[i] = (q128)rz[i];
  if (!q_lu_factor(nz, LU, piv)) {
    for (int i = 0; i < nz * nth; ++i) dz[i] = NAN;
    if (cond) *cond = NAN;
    return 0;
  }
  for (int c = 0; c < nth; ++c) {
    for (int i = 0; i < nz; ++i) x[i] = -(q128)rth[i + nz * c];
    q_lu_solve(nz, LU, piv, x);
    for (int rep = 0; rep < 2; ++rep) {                       /* iterative refinement in the same precision */
      for (int i = 0; i < nz; ++i) {
        q128 s = -(q128)rth[i + nz * c];
        for (int j = 0; j < nz; ++j) s -= A[i + nz * j] * x[j];
        res[i] = s;
      }
      q_lu_solve(nz, LU, piv, res);
      for (int i = 0; i < nz; ++i) x[i] += res[i];
    }
    for (int i = 0; i < nz; ++i) dz[i + nz * c] = (double)x[i];
  }
  if (cond) {
    q128 na = 0, ni = 0;
    for (int c = 0; c < nz; ++c) {
      for (int i = 0; i < nz; ++i) x[i] = (i == c) ? 1 : 0;
      q_lu_solve(nz, LU, piv, x);
      for (int i = 0; i < nz; ++i) inv[i + nz * c] = x[i];
    }
    for (int i = 0; i < nz; ++i) {
      q128 ra = 0, ri = 0;
      for (int j = 0; j < nz; ++j) { ra += fabsq(A[i + nz * j]); ri += fabsq(inv[i + nz * j]); }
      if (ra > na) na = ra;
      if (ri > ni) ni = ri;
    }
    *cond = (double)(na * ni);
  }
  return 1;
}

/* batched (OpenMP): Z (nz+1) x B with the clamp in row nz (the layout of the device's gradient hand-over workspace,
 * od_get_grad_iterates), X 2nq x B, U nu x B col-major -> G (nq*(2nq+nu)) x B = d q3 / d(q1, q2, u1) col-major, cond B */
int od_arbiter_dq3_batch(const od_oracle_sim* s, int B, const double* X, const double* U, const double* Zg, double* G, double* cond) {
  const od_oracle_model* m = od_oracle_models[s->model_id];
  const int nq = m->nq, nu = m->nu, nz = m->nz, n = 2 * nq, ngc = n + nu;
  int bad = 0;
#pragma omp parallel for reduction(+ : bad) schedule(dynamic, 8)
  for (int b = 0; b < B; ++b) {
    const double* x = X + (size_t)n * b;
    const double* u = U + (size_t)nu * b;
    const double* zg = Zg + (size_t)(nz + 1) * b;
    double th[NTHMAX], dz[NZMAX * NTHMAX];
    for (int i = 0; i < nq; ++i) {
      const double v1 = (x[nq + i] - x[i]) / s->h;
      th[i] = x[nq + i] - s->h * v1;
      th[nq + i] = x[nq + i];
    }
    for (int i = 0; i < nu; ++i) th[2 * nq + i] = u[i];
    for (int i = 0; i < m->nfric; ++i) th[2 * nq + nu + i] = s->fric[i];
    th[2 * nq + nu + m->nfric] = s->h;
    bad += !od_arbiter_gradient(s->model_id, zg, th, zg[nz], dz, cond ? cond + b : NULL);
    for (int c = 0; c < ngc; ++c)
      for (int i = 0; i < nq; ++i) G[(size_t)nq * ngc * b + i + nq * c] = dz[m->zq[i] + nz * c];
  }
  return bad;
}
