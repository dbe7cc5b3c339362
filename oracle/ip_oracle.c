/*
 * CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Never linked, imported or called by the product path
 * (optimization_dynamics_amd/); used by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg as the checker / reported baseline.
 *
 * PARITY UNPINNED: the reference (thowell/optimization_dynamics, Julia) cannot run in this
 * environment and ships no tests or golden vectors; the solver it calls lives in the un-vendored
 * package RoboDojo.jl (Project.toml:31, compat "0.1.2").  This file restates
 *   - the reference's own glue (src/dynamics.jl:81-128, src/gradient_bundle.jl:8-13,87-104,
 *     src/ls.jl:20-60, src/models/rocket/dynamics.jl:101-268) line by line, and
 *   - RoboDojo's interior-point loop from its published structure as recalled:
 *     Mehrotra predictor-corrector on the relaxed KKT system with a dense partial-pivot LU.
 * Every function cites what it follows.  Residuals / Jacobians come from oracle/gen/*.h, generated
 * from the symbolic statement in optimization_dynamics_amd/codegen/models.py and pinned against the
 * hand-written numpy restatement oracle/models_np.py by tests/test_models.py.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

int od_oracle_trace = 0;   /* tests can switch on a per-iteration trace */
void od_oracle_set_trace(int v) { od_oracle_trace = v; }

/* od_oracle_set_exact_boundary(1): the thrust-cone projection's loop (eps_min = 0: full steps to the boundary of an orthant, tau = 1)
 * with the three places completed as exact arithmetic has them, where the literal double-precision transcription (the default, 0) reads
 * rounding noise: (a) the blocking variable of an accepted full step is exactly zero; (b) the direction component of a variable that sits
 * at zero comes from its own complementarity row (one entry left: exact); (c) the line search's `r_c <= r_vio` on the LINEAR equality rows
 * holds for every step length, so the first trial is accepted.  This is the algorithm oracle/arbiter.c::od_arbiter_soc_projection runs in
 * binary128 with exact acceptance, in double and with the dense pivoted LU: a second implementation of the exact-arithmetic path, to which
 * the device's closed-form solve is compared at 1e-6 (tests/parity_checks.py::check_rocket_sweep).  The default stays the literal loop. */
int od_oracle_exact_boundary = 0;
void od_oracle_set_exact_boundary(int v) { od_oracle_exact_boundary = v; }

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

static inline double od_powi(double x, int n) {
  double r = 1.0;
  for (int i = 0; i < n; ++i) r *= x;
  return r;
}

typedef struct {
  double r_tol, kappa_tol, kappa_grad_tol;
  int max_iter, max_ls;
  double eps_min, kappa_reg, gamma_reg, undercut;
} od_oracle_opts;

typedef struct {
  const char* name;
  int id, kind, nq, nu, nz, nth, nfric;
  int nort; const int* ort1; const int* ort2; const int* ortr;
  int nsoc; const int* socoff; const int* soc1; const int* soc2; const int* socr;
  int neq; const int* equr; int nbil; const int* bil; int nzq; const int* zq;
  const int* zikind; const int* ziidx; const double* zival; const double* fric;
  od_oracle_opts opts;
  void (*r)(const double*, const double*, double, double*);
  void (*rz)(const double*, const double*, double*);
  void (*rth)(const double*, const double*, double*);
} od_oracle_model;

#include "gen/models_gen.h"

#define NZMAX 40
#define NTHMAX 20

/* ----------------------------------------------------------------------------------------------
 * dense LU with partial pivoting (RoboDojo lu_solver / linear_solve!, called at
 * src/ls.jl:52 and inside interior_point_solve!).  Column-major n x n, in place.
 * -------------------------------------------------------------------------------------------- */
static int lu_factor(int n, double* A, int* piv) {
  int ok = 1;
  for (int k = 0; k < n; ++k) {
    int p = k;
    double best = fabs(A[k + n * k]);
    for (int i = k + 1; i < n; ++i) {
      double v = fabs(A[i + n * k]);
      if (v > best) { best = v; p = i; }
    }
    piv[k] = p;
    if (best == 0.0 || best != best) ok = 0;
    if (p != k)
      for (int j = 0; j < n; ++j) { double t = A[k + n * j]; A[k + n * j] = A[p + n * j]; A[p + n * j] = t; }
    double inv = 1.0 / A[k + n * k];
    for (int i = k + 1; i < n; ++i) A[i + n * k] *= inv;
    for (int j = k + 1; j < n; ++j) {
      double ukj = A[k + n * j];
      for (int i = k + 1; i < n; ++i) A[i + n * j] -= A[i + n * k] * ukj;
    }
  }
  return ok;
}

static void lu_solve(int n, const double* A, const int* piv, double* b) {
  for (int k = 0; k < n; ++k) {          /* P b (rows of L were swapped along with A) */
    int p = piv[k];
    if (p != k) { double t = b[k]; b[k] = b[p]; b[p] = t; }
  }
  for (int k = 0; k < n; ++k)
    for (int i = k + 1; i < n; ++i) b[i] -= A[i + n * k] * b[k];
  for (int k = n - 1; k >= 0; --k) {
    b[k] /= A[k + n * k];
    for (int i = 0; i < k; ++i) b[i] -= A[i + n * k] * b[k];
  }
}

/* ----------------------------------------------------------------------------------------------
 * cone utilities (RoboDojo src/solver/cones.jl as recalled; cone_product pinned by
 * src/models/cartpole/model.jl:111-112)
 * -------------------------------------------------------------------------------------------- */
static double residual_violation(const od_oracle_model* m, const double* r) {
  double v = 0.0;
  for (int i = 0; i < m->neq; ++i) { double a = fabs(r[m->equr[i]]); if (a > v || a != a) v = a; }
  return v;
}
static double bilinear_violation(const od_oracle_model* m, const double* r) {
  double v = 0.0;
  for (int i = 0; i < m->nbil; ++i) { double a = fabs(r[m->bil[i]]); if (a > v || a != a) v = a; }
  return v;
}

/* maximum alpha in (0,1] keeping the orthant variables of z - alpha*D positive (fraction tau) */
static double ort_step_length(const od_oracle_model* m, const double* z, const double* D, double tau) {
  double a = 1.0;
  for (int s = 0; s < 2; ++s) {
    const int* idx = s == 0 ? m->ort1 : m->ort2;
    for (int i = 0; i < m->nort; ++i) {
      int k = idx[i];
      if (D[k] > 0.0) { double c = tau * z[k] / D[k]; if (c < a) a = c; }
    }
  }
  return a;
}

/* CVXOPT sec. 8.2 step to the boundary of the second-order cone for lam + alpha*dlt
 * (RoboDojo soc_step_length as recalled, eps guards 1e-14 / 1e-25) */
static double soc_step_one(int n, const double* lam, const double* dlt, double tau) {
  const double eps = 1e-14;
  double l0 = lam[0], ll = l0 * l0, ld = l0 * dlt[0];
  for (int i = 1; i < n; ++i) { ll -= lam[i] * lam[i]; ld -= lam[i] * dlt[i]; }
  if (ll < 1e-25) ll = 1e-25;
  ll += eps;
  ld += eps;
  double sq = sqrt(ll);
  double rs = ld / ll;
  double c = (ld / sq + dlt[0]) / (l0 / sq + 1.0);
  double nv = 0.0;
  for (int i = 1; i < n; ++i) { double rv = dlt[i] / sq - c * lam[i] / ll; nv += rv * rv; }
  nv = sqrt(nv);
  double a = 1.0;
  if (nv - rs > 0.0) { double cnd = tau / (nv - rs); if (cnd < a) a = cnd; }
  return a;
}

static double soc_step_length(const od_oracle_model* m, const double* z, const double* D, double tau) {
  double a = 1.0, lam[8], dl[8];
  for (int c = 0; c < m->nsoc; ++c) {
    int o = m->socoff[c], n = m->socoff[c + 1] - o;
    for (int s = 0; s < 2; ++s) {
      const int* idx = s == 0 ? m->soc1 : m->soc2;
      for (int i = 0; i < n; ++i) { lam[i] = z[idx[o + i]]; dl[i] = -D[idx[o + i]]; } /* step is z - alpha*D */
      double cnd = soc_step_one(n, lam, dl, tau);
      if (cnd < a) a = cnd;
    }
  }
  return a;
}

static double step_length(const od_oracle_model* m, const double* z, const double* D, double tau_ort, double tau_soc) {
  double a = ort_step_length(m, z, D, tau_ort);
  double b = soc_step_length(m, z, D, tau_soc);
  return a < b ? a : b;
}
/* the orthant variable whose ratio test sets step_length(), -1 if a cone does or the step is the full one */
static int blocking_variable(const od_oracle_model* m, const double* z, const double* D, double tau_ort, double tau_soc) {
  double a = 1.0;
  int kb = -1;
  for (int s = 0; s < 2; ++s) {
    const int* idx = s == 0 ? m->ort1 : m->ort2;
    for (int i = 0; i < m->nort; ++i) {
      int k = idx[i];
      if (D[k] > 0.0) { double c = tau_ort * z[k] / D[k]; if (c < a) { a = c; kb = k; } }
    }
  }
  return soc_step_length(m, z, D, tau_soc) < a ? -1 : kb;
}
/* (b) of od_oracle_set_exact_boundary: z_a D_b + z_b D_a = rhs with z_a = 0 exactly gives D_a = rhs / z_b */
static void exact_boundary_rows(const od_oracle_model* m, const double* z, const double* rhs, double* x) {
  for (int i = 0; i < m->nort; ++i) {
    int a = m->ort1[i], b = m->ort2[i], row = m->ortr[i];
    if (z[a] == 0.0 && z[b] != 0.0) x[a] = rhs[row] / z[b];
    if (z[b] == 0.0 && z[a] != 0.0) x[b] = rhs[row] / z[a];
  }
}

/* CVXOPT sec. 5.1.3: mu = <primal, dual> / (number of cones), sigma = clamp(mu_aff/mu,0,1)^3 */
static void centering(const od_oracle_model* m, const double* z, const double* Da, double aaff, double* mu, double* sigma) {
  int n = m->nort + m->nsoc;
  double s = 0.0, sa = 0.0;
  for (int i = 0; i < m->nort; ++i) {
    int a = m->ort1[i], b = m->ort2[i];
    s += z[a] * z[b];
    sa += (z[a] - aaff * Da[a]) * (z[b] - aaff * Da[b]);
  }
  for (int c = 0; c < m->nsoc; ++c)
    for (int k = m->socoff[c]; k < m->socoff[c + 1]; ++k) {
      int a = m->soc1[k], b = m->soc2[k];
      s += z[a] * z[b];
      sa += (z[a] - aaff * Da[a]) * (z[b] - aaff * Da[b]);
    }
  *mu = s / n;
  double q = (sa / n) / (*mu);
  if (q < 0.0) q = 0.0;
  if (q > 1.0) q = 1.0;
  *sigma = q * q * q;
}

/* general_correction_term! : r[bil] += cone_product(Daff_primal, Daff_dual) */
static void correction_term(const od_oracle_model* m, double* r, const double* Da) {
  for (int i = 0; i < m->nort; ++i) r[m->ortr[i]] += Da[m->ort1[i]] * Da[m->ort2[i]];
  for (int c = 0; c < m->nsoc; ++c) {
    int o = m->socoff[c], n = m->socoff[c + 1] - o;
    double dot = 0.0;
    for (int i = 0; i < n; ++i) dot += Da[m->soc1[o + i]] * Da[m->soc2[o + i]];
    r[m->socr[o]] += dot;
    for (int i = 1; i < n; ++i)
      r[m->socr[o + i]] += Da[m->soc1[o]] * Da[m->soc2[o + i]] + Da[m->soc2[o]] * Da[m->soc1[o + i]];
  }
}

/* rz! with the regularisation clamp on the orthant variables (ContactImplicitMPC/RoboDojo rz!(ip,...;reg)) */
static void rz_reg(const od_oracle_model* m, double* rz, const double* z, const double* th, double reg) {
  double zr[NZMAX];
  memcpy(zr, z, sizeof(double) * m->nz);
  for (int i = 0; i < m->nort; ++i) {
    if (zr[m->ort1[i]] < reg) zr[m->ort1[i]] = reg;
    if (zr[m->ort2[i]] < reg) zr[m->ort2[i]] = reg;
  }
  m->rz(zr, th, rz);
}

/* ----------------------------------------------------------------------------------------------
 * interior_point_solve! (RoboDojo src/solver/interior_point.jl, recalled).
 *   z in/out, theta in.  dz (nz x nth col-major) written when diff_sol.  returns status (1 ok).
 * -------------------------------------------------------------------------------------------- */
static int ip_solve_impl(int model_id, const od_oracle_opts* o, double kappa_tol, int diff_sol,
                         double* z, const double* th, double* dz, int* iters_out, double* reg_out) {
  const od_oracle_model* m = od_oracle_models[model_id];
  const int nz = m->nz, nth = m->nth, ncone = m->nort + m->nsoc;
  double r[NZMAX], Da[NZMAX], D[NZMAX], zc[NZMAX], rz[NZMAX * NZMAX];
  int piv[NZMAX];
  double reg_val = 0.0;
  int iters = 0;

  m->r(z, th, 0.0, r);
  double r_vio = residual_violation(m, r);
  double k_vio = bilinear_violation(m, r);

  for (int j = 0; j < o->max_iter; ++j) {
    if (r_vio < o->r_tol && k_vio < kappa_tol) break;
    iters++;
    reg_val = (k_vio < o->kappa_reg) ? k_vio * o->gamma_reg : 0.0;
    rz_reg(m, rz, z, th, reg_val);
    lu_factor(nz, rz, piv);
    const int exact = od_oracle_exact_boundary && model_id == 6 /* ROCKET_PROJ: linear equality rows, eps_min = 0 */;
    memcpy(Da, r, sizeof(double) * nz);
    lu_solve(nz, rz, piv, Da);                       /* affine direction */
    if (exact) exact_boundary_rows(m, z, r, Da);
    if (ncone > 0) {
      double aaff = step_length(m, z, Da, 1.0, 1.0);
      double mu, sigma;
      centering(m, z, Da, aaff, &mu, &sigma);
      double kap = sigma * mu;
      double floor_ = kappa_tol / o->undercut;       /* undercut = Inf -> 0 (src/dynamics.jl:26) */
      if (floor_ > kap) kap = floor_;
      m->r(z, th, kap, r);
      correction_term(m, r, Da);
      memcpy(D, r, sizeof(double) * nz);
      lu_solve(nz, rz, piv, D);                      /* corrector direction, factors reused */
      if (exact) exact_boundary_rows(m, z, r, D);
    } else {
      memcpy(D, Da, sizeof(double) * nz);            /* no cones: plain Newton */
    }
    double vio = r_vio > k_vio ? r_vio : k_vio;
    double eps = vio * vio;
    if (o->eps_min < eps) eps = o->eps_min;
    double tau = 1.0 - eps;                          /* progress!: tau = 1 - min(eps_min, vio^2) */
    double alpha = step_length(m, z, D, tau, tau < 0.99 ? tau : 0.99);
    const int blk = (exact && tau == 1.0) ? blocking_variable(m, z, D, tau, tau < 0.99 ? tau : 0.99) : -1;
    double r_c = 0.0, k_c = 0.0;
    for (int i = 0; i < o->max_ls; ++i) {
      for (int k = 0; k < nz; ++k) zc[k] = z[k] - alpha * D[k];
      if (i == 0 && blk >= 0) zc[blk] = 0.0;
      m->r(zc, th, 0.0, r);
      r_c = residual_violation(m, r);
      k_c = bilinear_violation(m, r);
      if (exact || r_c <= r_vio || k_c <= k_vio) break;
      alpha *= 0.5;
    }
    memcpy(z, zc, sizeof(double) * nz);
    r_vio = r_c;
    k_vio = k_c;
    if (od_oracle_trace) {
      printf("ora it %d alpha %.17g r_vio %.6e k_vio %.6e z", iters, alpha, r_vio, k_vio);
      for (int k = 0; k < nz; ++k) printf(" %.17g", z[k]);
      printf("\n");
    }
  }
  if (iters_out) *iters_out = iters;
  int status = (r_vio < o->r_tol && k_vio < kappa_tol) ? 1 : 0;   /* NaN -> 0 */
  if (diff_sol) {
    /* differentiate_solution!: dz = -rz(z*)^{-1} rtheta(z*), reg = max(reg_val, kappa_tol*gamma_reg) */
    double reg = kappa_tol * o->gamma_reg;
    if (reg_val > reg) reg = reg_val;
    if (reg_out) *reg_out = reg;
    rz_reg(m, rz, z, th, reg);
    m->rth(z, th, dz);
    lu_factor(nz, rz, piv);
    for (int c = 0; c < nth; ++c) {
      lu_solve(nz, rz, piv, dz + nz * c);
      for (int k = 0; k < nz; ++k) dz[k + nz * c] = -dz[k + nz * c];
    }
  }
  return status;
}

int od_oracle_ip_solve(int model_id, const od_oracle_opts* o, double kappa_tol, int diff_sol,
                       double* z, const double* th, double* dz, int* iters_out) {
  return ip_solve_impl(model_id, o, kappa_tol, diff_sol, z, th, dz, iters_out, NULL);
}

/* ----------------------------------------------------------------------------------------------
 * RoboDojo.step!(sim, q2, v1, u, 1) as used by src/dynamics.jl:88,103,123:
 *   z <- initialize_z!(q2), theta <- [q2 - h*v1; q2; u; w(empty); friction; h], solve, return q3.
 * -------------------------------------------------------------------------------------------- */
static void init_z(const od_oracle_model* m, const double* q, double* z) {
  for (int i = 0; i < m->nz; ++i) z[i] = m->zikind[i] == 0 ? q[m->ziidx[i]] : m->zival[i];
}

typedef struct {
  int model_id;
  od_oracle_opts opts;
  double h;
  double fric[4];
  double u_max;
} od_oracle_sim;

void od_oracle_default_sim(int model_id, double h, od_oracle_sim* s) {
  const od_oracle_model* m = od_oracle_models[model_id];
  s->model_id = model_id;
  s->opts = m->opts;
  s->h = h;
  for (int i = 0; i < 4; ++i) s->fric[i] = i < m->nfric ? m->fric[i] : 0.0;
  s->u_max = 12.5;
}

static int sim_step(const od_oracle_sim* s, double kappa_tol, int diff_sol,
                    const double* q2, const double* v1, const double* u,
                    double* z, double* dz, int* iters) {
  const od_oracle_model* m = od_oracle_models[s->model_id];
  double th[NTHMAX];
  int nq = m->nq, nu = m->nu;
  for (int i = 0; i < nq; ++i) { th[i] = q2[i] - s->h * v1[i]; th[nq + i] = q2[i]; }
  for (int i = 0; i < nu; ++i) th[2 * nq + i] = u[i];
  for (int i = 0; i < m->nfric; ++i) th[2 * nq + nu + i] = s->fric[i];
  th[2 * nq + nu + m->nfric] = s->h;
  init_z(m, q2, z);
  return od_oracle_ip_solve(s->model_id, &s->opts, kappa_tol, diff_sol, z, th, dz, iters);
}

/* f (src/dynamics.jl:81-94): d = [q2; q3] with the eval simulator (kappa_eval, no diff) */
int od_oracle_f(const od_oracle_sim* s, const double* x, const double* u, double* d, int* iters) {
  const od_oracle_model* m = od_oracle_models[s->model_id];
  int nq = m->nq;
  double v1[NZMAX], z[NZMAX];
  for (int i = 0; i < nq; ++i) v1[i] = (x[nq + i] - x[i]) / s->h;
  int st = sim_step(s, s->opts.kappa_tol, 0, x + nq, v1, u, z, NULL, iters);
  for (int i = 0; i < nq; ++i) { d[i] = x[nq + i]; d[nq + i] = z[m->zq[i]]; }
  return st;
}

/* fx (src/dynamics.jl:96-114): dx (2nq x 2nq col-major, caller pre-zeroed) */
int od_oracle_fx(const od_oracle_sim* s, const double* x, const double* u, double* dx, int* iters) {
  const od_oracle_model* m = od_oracle_models[s->model_id];
  int nq = m->nq, n = 2 * nq, nz = m->nz;
  double v1[NZMAX], z[NZMAX], dz[NZMAX * NTHMAX];
  for (int i = 0; i < nq; ++i) v1[i] = (x[nq + i] - x[i]) / s->h;
  int st = sim_step(s, s->opts.kappa_grad_tol, 1, x + nq, v1, u, z, dz, iters);
  for (int i = 0; i < nq; ++i) dx[i + n * (nq + i)] = 1.0;
  for (int c = 0; c < nq; ++c)
    for (int i = 0; i < nq; ++i) {
      dx[(nq + i) + n * c] = dz[m->zq[i] + nz * c];                 /* dq3/dq1 */
      dx[(nq + i) + n * (nq + c)] = dz[m->zq[i] + nz * (nq + c)];   /* dq3/dq2 */
    }
  return st;
}

/* fu (src/dynamics.jl:116-128): du (2nq x nu col-major) */
int od_oracle_fu(const od_oracle_sim* s, const double* x, const double* u, double* du, int* iters) {
  const od_oracle_model* m = od_oracle_models[s->model_id];
  int nq = m->nq, n = 2 * nq, nz = m->nz, nu = m->nu;
  double v1[NZMAX], z[NZMAX], dz[NZMAX * NTHMAX];
  for (int i = 0; i < nq; ++i) v1[i] = (x[nq + i] - x[i]) / s->h;
  int st = sim_step(s, s->opts.kappa_grad_tol, 1, x + nq, v1, u, z, dz, iters);
  for (int c = 0; c < nu; ++c)
    for (int i = 0; i < nq; ++i) du[(nq + i) + n * c] = dz[m->zq[i] + nz * (2 * nq + c)];
  return st;
}

/* full solution access for tests: z*, and dz (nz x nth) at a given tolerance */
int od_oracle_step_full(const od_oracle_sim* s, double kappa_tol, int diff_sol, const double* x, const double* u,
                        double* z, double* dz, int* iters) {
  const od_oracle_model* m = od_oracle_models[s->model_id];
  int nq = m->nq;
  double v1[NZMAX];
  for (int i = 0; i < nq; ++i) v1[i] = (x[nq + i] - x[i]) / s->h;
  return sim_step(s, kappa_tol, diff_sol, x + nq, v1, u, z, dz, iters);
}

/* the grad simulator's iterate and the clamp differentiate_solution! used, (nz+1) x B col-major like the device's
 * hand-over workspace, plus its dq3/d(q1,q2,u1) -- input of the extended-precision arbiter (arbiter.c) */
int od_oracle_grad_iterates(const od_oracle_sim* s, int B, const double* X, const double* U, double* Zg, double* G) {
  const od_oracle_model* m = od_oracle_models[s->model_id];
  const int nq = m->nq, nu = m->nu, nz = m->nz, n = 2 * nq, ngc = n + nu;
  int bad = 0;
#pragma omp parallel for reduction(+ : bad) schedule(dynamic, 16)
  for (int b = 0; b < B; ++b) {
    const double* x = X + (size_t)n * b;
    double v1[NZMAX], th[NTHMAX], z[NZMAX], dz[NZMAX * NTHMAX], reg = 0.0;
    for (int i = 0; i < nq; ++i) { v1[i] = (x[nq + i] - x[i]) / s->h; th[i] = x[nq + i] - s->h * v1[i]; th[nq + i] = x[nq + i]; }
    for (int i = 0; i < nu; ++i) th[2 * nq + i] = U[(size_t)nu * b + i];
    for (int i = 0; i < m->nfric; ++i) th[2 * nq + nu + i] = s->fric[i];
    th[2 * nq + nu + m->nfric] = s->h;
    init_z(m, x + nq, z);
    int it;
    bad += !ip_solve_impl(s->model_id, &s->opts, s->opts.kappa_grad_tol, 1, z, th, dz, &it, &reg);
    for (int i = 0; i < nz; ++i) Zg[(size_t)(nz + 1) * b + i] = z[i];
    Zg[(size_t)(nz + 1) * b + nz] = reg;
    for (int c = 0; c < ngc; ++c)
      for (int i = 0; i < nq; ++i) G[(size_t)nq * ngc * b + i + nq * c] = dz[m->zq[i] + nz * c];
  }
  return bad;
}

/* ----------------------------------------------------------------------------------------------
 * least squares (src/ls.jl:20-60) on cost sum_i |f_eta_i - f_z - M eta_i|^2, theta = vec(M)
 * (cost statement src/gradient_bundle.jl:35-39).  Newton with dense LU, tol 1e-8, <= 100 its.
 * theta (ny*nzb) in/out (warm start persists like ls.theta).
 * -------------------------------------------------------------------------------------------- */
int od_oracle_ls_update(int N, int ny, int nzb, const double* fz, const double* feta /* ny x N */,
                        const double* eta /* nzb x N */, double* theta) {
  int nt = ny * nzb, iter = 0;
  double* g = (double*)calloc(nt, sizeof(double));
  double* H = (double*)calloc((size_t)nt * nt, sizeof(double));
  int* piv = (int*)calloc(nt, sizeof(int));
  double res;
#define EVAL_GRAD()                                                                           \
  do {                                                                                        \
    memset(g, 0, sizeof(double) * nt);                                                        \
    for (int i = 0; i < N; ++i)                                                               \
      for (int a = 0; a < ny; ++a) {                                                          \
        double ra = feta[a + ny * i] - fz[a];                                                 \
        for (int b = 0; b < nzb; ++b) ra -= theta[a + ny * b] * eta[b + nzb * i];             \
        for (int b = 0; b < nzb; ++b) g[a + ny * b] += -2.0 * ra * eta[b + nzb * i];          \
      }                                                                                       \
    res = 0.0;                                                                                \
    for (int k = 0; k < nt; ++k) if (fabs(g[k]) > res) res = fabs(g[k]);                      \
  } while (0)
  EVAL_GRAD();
  while (res > 1e-8 && iter < 100) {
    memset(H, 0, sizeof(double) * nt * nt);
    for (int i = 0; i < N; ++i)
      for (int a = 0; a < ny; ++a)
        for (int b = 0; b < nzb; ++b)
          for (int c = 0; c < nzb; ++c)
            H[(a + ny * b) + nt * (a + ny * c)] += 2.0 * eta[b + nzb * i] * eta[c + nzb * i];
    lu_factor(nt, H, piv);
    lu_solve(nt, H, piv, g);                /* Delta */
    for (int k = 0; k < nt; ++k) theta[k] -= g[k];
    EVAL_GRAD();
    iter++;
  }
#undef EVAL_GRAD
  free(g); free(H); free(piv);
  return iter;
}

/* gradient! (src/gradient_bundle.jl:87-104): N+1 steps with the EVAL simulator + LS fit.
 * dzb: ny x nzb col-major (nzb = 2nq+nu).  theta: warm-start storage (ny*nzb). */
int od_oracle_gradient_bundle(const od_oracle_sim* s, int N, const double* eta /* nzb x N */,
                              const double* q1, const double* q2, const double* u1,
                              double* theta, double* dzb) {
  const od_oracle_model* m = od_oracle_models[s->model_id];
  int nq = m->nq, nu = m->nu, nzb = 2 * nq + nu;
  double* feta = (double*)calloc((size_t)nq * N, sizeof(double));
  double fz[NZMAX], x[2 * NZMAX], uu[NZMAX], d[2 * NZMAX];
  int it, ok = 1;
  for (int i = 0; i < nq; ++i) { x[i] = q1[i]; x[nq + i] = q2[i]; }
  ok &= od_oracle_f(s, x, u1, d, &it);      /* _step: v1=(q2-q1)/h; step!  (:8-13) */
  for (int i = 0; i < nq; ++i) fz[i] = d[nq + i];
  for (int k = 0; k < N; ++k) {
    const double* e = eta + nzb * k;
    for (int i = 0; i < nq; ++i) { x[i] = q1[i] + e[i]; x[nq + i] = q2[i] + e[nq + i]; }
    for (int i = 0; i < nu; ++i) uu[i] = u1[i] + e[2 * nq + i];
    ok &= od_oracle_f(s, x, uu, d, &it);
    for (int i = 0; i < nq; ++i) feta[i + nq * k] = d[nq + i];
  }
  od_oracle_ls_update(N, nq, nzb, fz, feta, eta, theta);
  memcpy(dzb, theta, sizeof(double) * nq * nzb);
  free(feta);
  return ok;
}

/* ----------------------------------------------------------------------------------------------
 * rocket (src/models/rocket/dynamics.jl)
 * -------------------------------------------------------------------------------------------- */
#define ROCKET_DYN 5
#define ROCKET_PROJ 6

/* f_rocket / fx_rocket / fu_rocket (:101-164): z0 = x, theta=[x;u;h]; dz = delta z (12 x 16) */
int od_oracle_rocket(double h, const double* x, const double* u, int diff_sol, double* y, double* dz, int* iters) {
  const od_oracle_model* m = od_oracle_models[ROCKET_DYN];
  double th[16];
  memcpy(th, x, 12 * sizeof(double));
  memcpy(th + 12, u, 3 * sizeof(double));
  th[15] = h;
  memcpy(y, x, 12 * sizeof(double));
  return od_oracle_ip_solve(ROCKET_DYN, &m->opts, m->opts.kappa_tol, diff_sol, y, th, dz, iters);
}

/* soc_projection / soc_projection_gradient (:168-210): returns z (10), dz (10 x 4) */
int od_oracle_soc_projection(double u_max, const double* u, int diff_sol, double* z, double* dz, int* iters) {
  const od_oracle_model* m = od_oracle_models[ROCKET_PROJ];
  double th[4] = {u[0], u[1], u[2], u_max};
  init_z(m, u, z);
  return od_oracle_ip_solve(ROCKET_PROJ, &m->opts, m->opts.kappa_tol, diff_sol, z, th, dz, iters);
}

/* f_rocket_proj / fx_rocket_proj / fu_rocket_proj (:215-268).
 * y (12), dx (12x12 col-major), du (12x3 col-major) = dz_dyn[:,u] * dproj[1:3,1:3] */
int od_oracle_rocket_proj(double h, double u_max, const double* x, const double* u,
                          double* y, double* dx, double* du) {
  double zp[10], dzp[40], up[3], dzd[12 * 16], yy[12];
  int it, ok = 1;
  ok &= od_oracle_soc_projection(u_max, u, 0, zp, NULL, &it);
  up[0] = zp[0]; up[1] = zp[1]; up[2] = zp[2];
  ok &= od_oracle_rocket(h, x, up, 1, yy, dzd, &it);
  if (y) memcpy(y, yy, sizeof(yy));
  if (dx) memcpy(dx, dzd, 144 * sizeof(double));
  if (du) {
    ok &= od_oracle_soc_projection(u_max, u, 1, zp, dzp, &it);
    for (int c = 0; c < 3; ++c)
      for (int i = 0; i < 12; ++i) {
        double a = 0.0;
        for (int k = 0; k < 3; ++k) a += dzd[i + 12 * (12 + k)] * dzp[k + 10 * c];
        du[i + 12 * c] = a;
      }
  }
  return ok;
}

/* ----------------------------------------------------------------------------------------------
 * raw access for model-pinning tests
 * -------------------------------------------------------------------------------------------- */
int od_oracle_num_models_(void) { return od_oracle_num_models; }
const char* od_oracle_model_name(int id) { return od_oracle_models[id]->name; }
void od_oracle_model_dims(int id, int* nq, int* nu, int* nz, int* nth, int* nfric) {
  const od_oracle_model* m = od_oracle_models[id];
  *nq = m->nq; *nu = m->nu; *nz = m->nz; *nth = m->nth; *nfric = m->nfric;
}
void od_oracle_eval_r(int id, const double* z, const double* th, double kappa, double* r) { od_oracle_models[id]->r(z, th, kappa, r); }
void od_oracle_eval_rz(int id, const double* z, const double* th, double* rz) { od_oracle_models[id]->rz(z, th, rz); }
void od_oracle_eval_rth(int id, const double* z, const double* th, double* rth) { od_oracle_models[id]->rth(z, th, rth); }

/* batched CPU baseline: B independent step+grad units (f at kappa_eval, fx and fu at kappa_grad:
 * the reference's three solves per knot), optionally OpenMP-parallel over the batch.
 * x: 2nq x B col-major, u: nu x B, d: 2nq x B, dx: (2nq*2nq) x B, du: (2nq*nu) x B */
int od_oracle_step_grad_batch(const od_oracle_sim* s, int B, const double* x, const double* u,
                              double* d, double* dx, double* du, int fused) {
  const od_oracle_model* m = od_oracle_models[s->model_id];
  int n = 2 * m->nq, nu = m->nu, bad = 0;
#pragma omp parallel for reduction(+ : bad) schedule(dynamic, 16)
  for (int b = 0; b < B; ++b) {
    int it;
    double* dxb = dx + (size_t)n * n * b;
    double* dub = du + (size_t)n * nu * b;
    memset(dxb, 0, sizeof(double) * n * n);
    memset(dub, 0, sizeof(double) * n * nu);
    bad += !od_oracle_f(s, x + (size_t)n * b, u + (size_t)nu * b, d + (size_t)n * b, &it);
    bad += !od_oracle_fx(s, x + (size_t)n * b, u + (size_t)nu * b, dxb, &it);
    if (!fused) bad += !od_oracle_fu(s, x + (size_t)n * b, u + (size_t)nu * b, dub, &it);
    else {
      /* same solve as fx: reuse (the two gradient solves are byte-identical work) */
      od_oracle_fu(s, x + (size_t)n * b, u + (size_t)nu * b, dub, &it);
    }
  }
  return bad;
}

/* sequential rollout of T steps for B trajectories (CPU baseline for the headline config):
 * X: 2nq x (T+1) x B, U: nu x T x B ; A: (2nq*2nq) x T x B ; Bm: (2nq*nu) x T x B (may be NULL) */
int od_oracle_rollout(const od_oracle_sim* s, int B, int T, const double* x1, const double* U,
                      double* X, double* A, double* Bm) {
  const od_oracle_model* m = od_oracle_models[s->model_id];
  int n = 2 * m->nq, nu = m->nu, bad = 0;
#pragma omp parallel for reduction(+ : bad) schedule(dynamic, 4)
  for (int b = 0; b < B; ++b) {
    int it;
    double* Xb = X + (size_t)n * (T + 1) * b;
    memcpy(Xb, x1 + (size_t)n * b, sizeof(double) * n);
    for (int t = 0; t < T; ++t) {
      const double* ut = U + (size_t)nu * (t + (size_t)T * b);
      bad += !od_oracle_f(s, Xb + n * t, ut, Xb + n * (t + 1), &it);
      if (A) {
        double* At = A + (size_t)n * n * (t + (size_t)T * b);
        memset(At, 0, sizeof(double) * n * n);
        bad += !od_oracle_fx(s, Xb + n * t, ut, At, &it);
      }
      if (Bm) {
        double* Bt = Bm + (size_t)n * nu * (t + (size_t)T * b);
        memset(Bt, 0, sizeof(double) * n * nu);
        bad += !od_oracle_fu(s, Xb + n * t, ut, Bt, &it);
      }
    }
  }
  return bad;
}

/* ----------------------------------------------------------------------------------------------
 * batched (OpenMP) forms of the rocket functions for the parity sweeps (tests/test_gpu_parity_sweep.py, tests/ilqr_checks.py):
 * col-major arrays with the knot index last, like od_oracle_step_grad_batch
 * -------------------------------------------------------------------------------------------- */
/* f_rocket + (diff_sol) fx / fu_rocket on B knots: X 12 x B, U 3 x B -> Y 12 x B, DZ (12*16) x B (may be NULL), status / iters B */
int od_oracle_rocket_batch(double h, int B, const double* X, const double* U, int diff_sol, double* Y, double* DZ, int* status, int* iters) {
  int bad = 0;
#pragma omp parallel for reduction(+ : bad) schedule(dynamic, 32)
  for (int b = 0; b < B; ++b) {
    int it = 0;
    double dz[12 * 16];
    const int ok = od_oracle_rocket(h, X + 12 * (size_t)b, U + 3 * (size_t)b, diff_sol, Y + 12 * (size_t)b, (diff_sol && DZ) ? DZ + 192 * (size_t)b : dz, &it);
    if (status) status[b] = ok;
    if (iters) iters[b] = it;
    bad += !ok;
  }
  return bad;
}
/* soc_projection(_gradient) on B controls: U 3 x B -> Z 10 x B, DZ 40 x B (may be NULL), status / iters B */
int od_oracle_soc_projection_batch(double u_max, int B, const double* U, int diff_sol, double* Z, double* DZ, int* status, int* iters) {
  int bad = 0;
#pragma omp parallel for reduction(+ : bad) schedule(dynamic, 32)
  for (int b = 0; b < B; ++b) {
    int it = 0;
    double dz[40];
    const int ok = od_oracle_soc_projection(u_max, U + 3 * (size_t)b, diff_sol, Z + 10 * (size_t)b, (diff_sol && DZ) ? DZ + 40 * (size_t)b : dz, &it);
    if (status) status[b] = ok;
    if (iters) iters[b] = it;
    bad += !ok;
  }
  return bad;
}
/* iLQR.rollout over f_rocket (project = 0) or f_rocket_proj (project = 1), examples/rocket.jl:29-41,118: B trajectories of T steps,
 * open loop (K = NULL) or closed loop u = ubar + alpha kff + K (x - xbar) (xbar 12 x (T+1) x B, K (3*12) x T x B col-major per knot,
 * kff 3 x T x B).  x1 12 x B, Ubar 3 x T x B -> X 12 x (T+1) x B, Uapp 3 x T x B (controls before projection; may be NULL),
 * status T x B (bit 0 dynamics, bit 4 projection converged; may be NULL) */
int od_oracle_rocket_rollout(double h, double u_max, int project, int B, int T, const double* x1, const double* Ubar, double alpha,
                             const double* xbar, const double* K, const double* kff, double* X, double* Uapp, int* status) {
  int bad = 0;
#pragma omp parallel for reduction(+ : bad) schedule(dynamic, 1)
  for (int b = 0; b < B; ++b) {
    double* Xb = X + (size_t)12 * (T + 1) * b;
    memcpy(Xb, x1 + 12 * (size_t)b, 12 * sizeof(double));
    for (int t = 0; t < T; ++t) {
      const size_t kn = (size_t)t + (size_t)T * b;
      double u[3], up[3], zp[10], dzp[40];
      int it, st = 0;
      for (int j = 0; j < 3; ++j) u[j] = Ubar[3 * kn + j];
      if (K) {
        const double* xb = xbar + 12 * ((size_t)t + (size_t)(T + 1) * b);
        for (int j = 0; j < 3; ++j) u[j] += alpha * kff[3 * kn + j];
        for (int i = 0; i < 12; ++i) {
          const double dxi = Xb[12 * t + i] - xb[i];
          for (int j = 0; j < 3; ++j) u[j] += K[36 * kn + j + 3 * i] * dxi;
        }
      }
      if (Uapp) for (int j = 0; j < 3; ++j) Uapp[3 * kn + j] = u[j];
      if (project) {
        const int okp = od_oracle_soc_projection(u_max, u, 0, zp, dzp, &it);
        st |= okp << 4;
        up[0] = zp[0]; up[1] = zp[1]; up[2] = zp[2];
      } else { up[0] = u[0]; up[1] = u[1]; up[2] = u[2]; st |= 1 << 4; }
      double dz[192];
      const int okd = od_oracle_rocket(h, Xb + 12 * t, up, 0, Xb + 12 * (t + 1), dz, &it);
      st |= okd;
      if (status) status[kn] = st;
      bad += (st != 0x11);
    }
  }
  return bad;
}

/* residual_violation / bilinear_violation of B given points (the convergence test of interior_point_solve!, SURVEY.md 3.4, at kappa = 0):
 * Z nz x B, TH nth x B -> r_vio, k_vio (B each).  Lets a test verify that an end point some other implementation returned passes the
 * algorithm's own stopping rule. */
void od_oracle_violations_batch(int model_id, int B, const double* Z, const double* TH, double* r_vio, double* k_vio) {
  const od_oracle_model* m = od_oracle_models[model_id];
#pragma omp parallel for schedule(static)
  for (int b = 0; b < B; ++b) {
    double r[NZMAX];
    m->r(Z + (size_t)m->nz * b, TH + (size_t)m->nth * b, 0.0, r);
    r_vio[b] = residual_violation(m, r);
    k_vio[b] = bilinear_violation(m, r);
  }
}

#include "arbiter.c"
