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

/* ------------------------------------------------------------------------------------------------------------------
 * The rocket's thrust-cone projection (src/models/rocket/dynamics.jl:168-186, options :77-86: eps_min = 0, gamma_reg = 0,
 * kappa_reg = 0, kappa_tol = 1e-4) solved by the SAME predictor-corrector loop (ip_solve_impl above) in
 * binary128 -- residual and Jacobian restated from gen/rocket_projection.h (src/models/rocket/codegen.jl:45-64).
 *
 * Why: with eps_min = 0 the step fraction is tau = 1, so from the first full step on the equality residual of this
 * problem is ROUNDING NOISE, and the line-search test `r_cand <= r_vio || k_cand <= k_vio` compares noise: two correct
 * double-precision implementations can accept different step lengths and end on different kappa_tol-accurate points.
 * In exact arithmetic there is no such freedom: the five equality rows are LINEAR in z, so r(z - a D) = (1 - a) r(z)
 * and |r_cand| <= |r_vio| holds for every a in [0, 1] -- the first trial is always accepted.  `exact_acceptance = 1`
 * applies that rule (and checks the linearity it rests on); `exact_acceptance = 0` runs the comparison as computed in
 * binary128 (noise at 1e-34 instead of 1e-16).  The tests compare the device's and the oracle's end points with the
 * exact-acceptance path and with the closed-form Euclidean projection.
 * ------------------------------------------------------------------------------------------------------------------ */
static void projq_r(const q128* z, const q128* th, q128 kappa, q128* r) {
  r[0] = -th[0] + z[0] - z[7];
  r[1] = -th[1] + z[1] - z[8];
  r[2] = -th[2] + z[2] - z[3] - z[6] - z[9];
  r[3] = th[3] - z[2] - z[4];
  r[4] = -z[5] - z[6];
  r[5] = -kappa + z[4] * z[5];
  r[6] = -kappa + z[2] * z[3];
  r[7] = -kappa + z[0] * z[7] + z[1] * z[8] + z[2] * z[9];
  r[8] = z[0] * z[9] + z[2] * z[7];
  r[9] = z[1] * z[9] + z[2] * z[8];
}
static void projq_rz(const q128* z, q128* rz) {
  for (int i = 0; i < 100; ++i) rz[i] = 0;
  rz[0] = 1; rz[70] = -1; rz[11] = 1; rz[81] = -1; rz[22] = 1; rz[32] = -1; rz[62] = -1; rz[92] = -1; rz[23] = -1; rz[43] = -1;
  rz[54] = -1; rz[64] = -1;
  rz[45] = z[5]; rz[55] = z[4]; rz[26] = z[3]; rz[36] = z[2];
  rz[7] = z[7]; rz[17] = z[8]; rz[27] = z[9]; rz[77] = z[0]; rz[87] = z[1]; rz[97] = z[2];
  rz[8] = z[9]; rz[28] = z[7]; rz[78] = z[2]; rz[98] = z[0];
  rz[19] = z[9]; rz[29] = z[8]; rz[89] = z[2]; rz[99] = z[1];
}
static q128 q_viol(const int* idx, int n, const q128* r) {
  q128 v = 0;
  for (int i = 0; i < n; ++i) { q128 a = fabsq(r[idx[i]]); if (a > v || a != a) v = a; }
  return v;
}
static q128 q_soc_step_one(int n, const q128* lam, const q128* dlt, q128 tau) {
  const q128 eps = 1e-14Q;
  q128 l0 = lam[0], ll = l0 * l0, ld = l0 * dlt[0];
  for (int i = 1; i < n; ++i) { ll -= lam[i] * lam[i]; ld -= lam[i] * dlt[i]; }
  if (ll < 1e-25Q) ll = 1e-25Q;
  ll += eps;
  ld += eps;
  q128 sq = sqrtq(ll), rs = ld / ll, c = (ld / sq + dlt[0]) / (l0 / sq + 1);
  q128 nv = 0;
  for (int i = 1; i < n; ++i) { q128 rv = dlt[i] / sq - c * lam[i] / ll; nv += rv * rv; }
  nv = sqrtq(nv);
  q128 a = 1;
  if (nv - rs > 0) { q128 cnd = tau / (nv - rs); if (cnd < a) a = cnd; }
  return a;
}
extern int od_oracle_trace;   /* ip_oracle.c */
/* blk (optional): the orthant variable whose ratio test set the step length, -1 if a cone did or the step is the full one */
static q128 q_step_length_blk(const od_oracle_model* m, const q128* z, const q128* D, q128 tau_ort, q128 tau_soc, int* blk) {
  q128 a = 1, lam[8], dl[8];
  int kb = -1;
  for (int s = 0; s < 2; ++s) {
    const int* idx = s == 0 ? m->ort1 : m->ort2;
    for (int i = 0; i < m->nort; ++i) { int k = idx[i]; if (D[k] > 0) { q128 c = tau_ort * z[k] / D[k]; if (c < a) { a = c; kb = k; } } }
  }
  const q128 ao = a;
  for (int c = 0; c < m->nsoc; ++c) {
    int o = m->socoff[c], n = m->socoff[c + 1] - o;
    for (int s = 0; s < 2; ++s) {
      const int* idx = s == 0 ? m->soc1 : m->soc2;
      for (int i = 0; i < n; ++i) { lam[i] = z[idx[o + i]]; dl[i] = -D[idx[o + i]]; }
      q128 cnd = q_soc_step_one(n, lam, dl, tau_soc);
      if (cnd < a) a = cnd;
    }
  }
  if (blk) *blk = a < ao ? -1 : kb;
  return a;
}
static q128 q_step_length(const od_oracle_model* m, const q128* z, const q128* D, q128 tau_ort, q128 tau_soc) {
  return q_step_length_blk(m, z, D, tau_ort, tau_soc, NULL);
}

/* An orthant variable that sits exactly at zero (the landing of a full step, below): its complementarity row z_a D_b + z_b D_a = rhs has
 * one entry left and gives D_a = rhs / z_b exactly -- zero for the affine direction.  The binary128 LU returns it with 1e-34 of noise of
 * either sign, and `D[k] > 0` of the ratio test would read that sign (blocked at step length 0, sigma = 1, or not blocking at all). */
static void q_exact_boundary_rows(const od_oracle_model* m, const q128* z, const q128* rhs, q128* x) {
  for (int i = 0; i < m->nort; ++i) {
    int a = m->ort1[i], b = m->ort2[i], row = m->ortr[i];
    if (z[a] == 0 && z[b] != 0) x[a] = rhs[row] / z[b];
    if (z[b] == 0 && z[a] != 0) x[b] = rhs[row] / z[a];
  }
}

/* u (3), u_max -> z (10, rounded to double), iterations, trials[it] = index of the accepted line-search trial of iteration
 * it (0 = the first), lin_err = largest violation of r(z - a D) = (1 - a) r(z) on the equality rows seen (relative).
 * returns 1 if converged to (r_tol, kappa_tol) */
int od_arbiter_soc_projection(double u_max, const double* u, int exact_acceptance, double* z_out, int* iters_out, int* trials, double* lin_err) {
  const od_oracle_model* m = od_oracle_models[ROCKET_PROJ];
  const od_oracle_opts* o = &m->opts;
  const int nz = 10, ncone = m->nort + m->nsoc;
  const q128 kappa_tol = o->kappa_tol;
  q128 z[10], th[4] = {u[0], u[1], u[2], u_max}, r[10], Da[10], D[10], zc[10], rz[100], rcand[10];
  int piv[10], iters = 0;
  for (int i = 0; i < nz; ++i) z[i] = m->zival[i];                 /* z .= 0.1; z[3] += 1; z[10] += 1; z[7] = 0 (1-based) */
  projq_r(z, th, 0, r);
  q128 r_vio = q_viol(m->equr, m->neq, r), k_vio = q_viol(m->bil, m->nbil, r);
  double lerr = 0.0;
  for (int j = 0; j < o->max_iter; ++j) {
    if (r_vio < (q128)o->r_tol && k_vio < kappa_tol) break;
    iters++;
    projq_rz(z, rz);                                               /* kappa_reg = 0: no clamp */
    if (!q_lu_factor(nz, rz, piv)) break;
    for (int i = 0; i < nz; ++i) Da[i] = r[i];
    q_lu_solve(nz, rz, piv, Da);
    q_exact_boundary_rows(m, z, r, Da);
    q128 aaff = q_step_length(m, z, Da, 1, 1);
    q128 s = 0, sa = 0;
    for (int i = 0; i < m->nort; ++i) { int a = m->ort1[i], b = m->ort2[i]; s += z[a] * z[b]; sa += (z[a] - aaff * Da[a]) * (z[b] - aaff * Da[b]); }
    for (int c = 0; c < m->nsoc; ++c)
      for (int k = m->socoff[c]; k < m->socoff[c + 1]; ++k) { int a = m->soc1[k], b = m->soc2[k]; s += z[a] * z[b]; sa += (z[a] - aaff * Da[a]) * (z[b] - aaff * Da[b]); }
    q128 mu = s / ncone, q = (sa / ncone) / mu;
    if (q < 0) q = 0;
    if (q > 1) q = 1;
    q128 kap = q * q * q * mu, floor_ = kappa_tol / (q128)o->undercut;
    if (floor_ > kap) kap = floor_;
    projq_r(z, th, kap, r);
    for (int i = 0; i < m->nort; ++i) r[m->ortr[i]] += Da[m->ort1[i]] * Da[m->ort2[i]];
    for (int c = 0; c < m->nsoc; ++c) {
      int o_ = m->socoff[c], n = m->socoff[c + 1] - o_;
      q128 dot = 0;
      for (int i = 0; i < n; ++i) dot += Da[m->soc1[o_ + i]] * Da[m->soc2[o_ + i]];
      r[m->socr[o_]] += dot;
      for (int i = 1; i < n; ++i) r[m->socr[o_ + i]] += Da[m->soc1[o_]] * Da[m->soc2[o_ + i]] + Da[m->soc2[o_]] * Da[m->soc1[o_ + i]];
    }
    for (int i = 0; i < nz; ++i) D[i] = r[i];
    q_lu_solve(nz, rz, piv, D);
    q_exact_boundary_rows(m, z, r, D);
    q128 vio = r_vio > k_vio ? r_vio : k_vio, eps = vio * vio;
    if ((q128)o->eps_min < eps) eps = o->eps_min;
    q128 tau = 1 - eps;
    int blk = -1;
    q128 alpha = q_step_length_blk(m, z, D, tau, tau < 0.99Q ? tau : 0.99Q, &blk);
    if (!exact_acceptance || tau != 1) blk = -1;
    q128 r0[10];
    projq_r(z, th, 0, r0);
    q128 r_c = 0, k_c = 0;
    int tr = 0;
    for (int i = 0; i < o->max_ls; ++i) {
      tr = i;
      for (int k = 0; k < nz; ++k) zc[k] = z[k] - alpha * D[k];
      /* exact arithmetic: a full step (tau = 1) to the boundary of an orthant leaves the blocking variable exactly at zero,
       * z_k - (z_k / D_k) D_k; binary128 leaves the rounding residual of the division (1e-34, either sign), and the algorithm is
       * discontinuous there (the sign of the next affine direction of that variable) -- completed exactly */
      if (i == 0 && blk >= 0) zc[blk] = 0;
      projq_r(zc, th, 0, rcand);
      r_c = q_viol(m->equr, m->neq, rcand);
      k_c = q_viol(m->bil, m->nbil, rcand);
      if (i == 0)                                                   /* the linearity the exact rule rests on */
        for (int e = 0; e < m->neq; ++e) {
          q128 want = (1 - alpha) * r0[m->equr[e]], sc = fabsq(r0[m->equr[e]]) + 1e-30Q;
          double d = (double)(fabsq(rcand[m->equr[e]] - want) / (sc + fabsq(z[0]) + fabsq(th[2]) + 1));
          if (d > lerr) lerr = d;
        }
      if (exact_acceptance || r_c <= r_vio || k_c <= k_vio) break;
      alpha *= 0.5Q;
    }
    if (trials) trials[iters - 1] = tr;
    for (int k = 0; k < nz; ++k) z[k] = zc[k];
    for (int k = 0; k < nz; ++k) r[k] = rcand[k];
    r_vio = r_c;
    k_vio = k_c;
    if (od_oracle_trace) {
      printf("arb it %d alpha %.17g r_vio %.6e k_vio %.6e trial %d z", iters, (double)alpha, (double)r_vio, (double)k_vio, tr);
      for (int k = 0; k < nz; ++k) printf(" %.17g", (double)z[k]);
      printf("\n");
    }
  }
  for (int i = 0; i < nz; ++i) z_out[i] = (double)z[i];
  if (iters_out) *iters_out = iters;
  if (lin_err) *lin_err = lerr;
  return (r_vio < (q128)o->r_tol && k_vio < kappa_tol) ? 1 : 0;
}

/* batched (OpenMP) forms.  od_arbiter_gradient_batch: Z nz x B, TH nth x B, reg B (NULL = 0) -> DZ (nz*nth) x B, cond B.
 * od_arbiter_soc_projection_batch: U 3 x B -> Z 10 x B, ok / iters B. */
int od_arbiter_gradient_batch(int model_id, int B, const double* Z, const double* TH, const double* reg, double* DZ, double* cond) {
  const od_oracle_model* m = od_oracle_models[model_id];
  const int nz = m->nz, nth = m->nth;
  int bad = 0;
#pragma omp parallel for reduction(+ : bad) schedule(dynamic, 16)
  for (int b = 0; b < B; ++b)
    bad += !od_arbiter_gradient(model_id, Z + (size_t)nz * b, TH + (size_t)nth * b, reg ? reg[b] : 0.0, DZ + (size_t)nz * nth * b, cond ? cond + b : NULL);
  return bad;
}
int od_arbiter_soc_projection_batch(double u_max, int B, const double* U, int exact_acceptance, double* Z, int* ok, int* iters) {
  int bad = 0;
#pragma omp parallel for reduction(+ : bad) schedule(dynamic, 8)
  for (int b = 0; b < B; ++b) {
    int it = 0;
    double le;
    const int o = od_arbiter_soc_projection(u_max, U + 3 * (size_t)b, exact_acceptance, Z + 10 * (size_t)b, &it, NULL, &le);
    if (ok) ok[b] = o;
    if (iters) iters[b] = it;
    bad += !o;
  }
  return bad;
}
