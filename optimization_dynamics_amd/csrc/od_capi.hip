// C ABI of libod_mi355x.so (include/od_mi355x.h): argument checking, view construction, launches.
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>

#include <vector>

#include "../../include/od_mi355x.h"
#include "od_vtable.h"

using namespace od;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

#define OD_HIP(call)                                                                      \
  do {                                                                                    \
    hipError_t e_ = (call);                                                               \
    if (e_ != hipSuccess) return fail(OD_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
  } while (0)

// A handle lives on the device that was current in od_create.  Every entry point runs on that device: the calling thread's current
// device is switched for the duration of the call and put back on the way out (the behaviour of a device guard), so that
// ImplicitDynamics(device = 1) works from a thread whose current device is 0 -- launches, allocations and attribute settings are
// per device in HIP, and nothing else in this library names one.
struct OnDevice {
  int prev = -1;
  bool switched = false;
  hipError_t err = hipSuccess;
  explicit OnDevice(int dev) {
    err = hipGetDevice(&prev);
    if (err == hipSuccess && prev != dev) {
      err = hipSetDevice(dev);
      switched = err == hipSuccess;
    }
  }
  ~OnDevice() { if (switched) (void)hipSetDevice(prev); }
  OnDevice(const OnDevice&) = delete;
  OnDevice& operator=(const OnDevice&) = delete;
};
#define OD_ON_DEVICE(h_)                                                                                   \
  OnDevice od_guard_((h_)->device);                                                                         \
  if (od_guard_.err != hipSuccess) return fail(OD_ERR_HIP, std::string("switching to the handle's device: ") + hipGetErrorString(od_guard_.err))

const ModelVT* vt_of(int model) {
  switch (model) {
#define OD_VT_CASE(name, id) case id: return vt_##name();
    OD_FOR_EACH_MODEL(OD_VT_CASE)
#undef OD_VT_CASE
    default: return nullptr;
  }
}

void defaults_of(const ModelVT* vt, od_options* o) {
  o->r_tol = vt->r_tol;
  o->kappa_eval_tol = vt->kappa_eval;
  o->kappa_grad_tol = vt->kappa_grad;
  o->max_iter = vt->max_iter;
  o->max_ls = vt->max_ls;
  o->eps_min = vt->eps_min;
  o->kappa_reg = vt->kappa_reg;
  o->gamma_reg = vt->gamma_reg;
  o->undercut = vt->undercut >= 1e299 ? INFINITY : vt->undercut;
}

template <class T> Opts<T> to_opts(const od_options& o) {
  Opts<T> r;
  r.r_tol = (T)o.r_tol;
  r.kappa_eval = (T)o.kappa_eval_tol;
  r.kappa_grad = (T)o.kappa_grad_tol;
  r.eps_min = (T)o.eps_min;
  r.kappa_reg = (T)o.kappa_reg;
  r.gamma_reg = (T)o.gamma_reg;
  r.undercut_inv = std::isinf(o.undercut) ? T(0) : (T)(1.0 / o.undercut);
  r.max_iter = o.max_iter;
  r.max_ls = o.max_ls;
  r.coop = 0;               // set by the launchers that know the lane map (od_model_tu.inc)
  return r;
}


// Least-squares fit of src/ls.jl:44-60 for the cost of src/gradient_bundle.jl:35-39,
//   sum_i | f_eta_i - f_z - M eta_i |^2 ,   M = ny x nzb,
// in closed form (normal equations M * sum(eta eta') = sum((f_eta - f_z) eta')): the cost is exactly
// quadratic, so the reference's Newton iteration lands on this minimiser in one step.
// Two kernels: (1) one lane per entry of [G | R'] = [sum eta eta' | sum eta (f_eta - f_z)'] per fit
// (N-term dot products, B * nzb * (nzb + ny) lanes), (2) one lane per fit: LU with partial pivoting
// (RoboDojo lu_solver, src/ls.jl:52) and ny back-solves.  ne = nzb * (nzb + ny) doubles of scratch per fit.
constexpr int OD_LS_MAX = 24;
__global__ __launch_bounds__(OD_BLOCK) void k_ls_accumulate(long B, int N, int ny, int nzb, const double* eta,
                                                           View<const double> feta, double* acc) {
  const int ne = nzb * (nzb + ny);
  const long t = (long)blockIdx.x * OD_BLOCK + threadIdx.x;
  if (t >= B * ne) return;
  const long b = t / ne;
  const int e = (int)(t - b * ne);
  const int r = e % nzb, c = e / nzb;          // column-major nzb x (nzb + ny)
  const long p0 = b * (N + 1);
  double s = 0.0;
  if (c < nzb) {
    for (int i = 0; i < N; ++i) s += eta[r + (long)nzb * i] * eta[c + (long)nzb * i];
  } else {
    const int a = c - nzb;
    const double fz = feta.at(a, p0);
    for (int i = 0; i < N; ++i) s += eta[r + (long)nzb * i] * (feta.at(a, p0 + 1 + i) - fz);
  }
  acc[t] = s;
}

__global__ __launch_bounds__(OD_BLOCK) void k_ls_solve(long B, int ny, int nzb, const double* acc,
                                                      View<double> dz, View<int> status) {
  const long b = (long)blockIdx.x * OD_BLOCK + threadIdx.x;
  if (b >= B) return;
  const int ne = nzb * (nzb + ny);
  double A[OD_LS_MAX * OD_LS_MAX], x[OD_LS_MAX];
  int piv[OD_LS_MAX];
  const double* g = acc + b * ne;
  for (int i = 0; i < nzb * nzb; ++i) A[i] = g[i];
  bool ok = true;
  double chk = 0.0;
  for (int k = 0; k < nzb; ++k) {
    int p = k;
    double best = fabs(A[k + nzb * k]);
    for (int i = k + 1; i < nzb; ++i) { const double v = fabs(A[i + nzb * k]); if (v > best) { best = v; p = i; } }
    piv[k] = p;
    ok = ok && (best > 0.0);
    if (p != k) for (int j = 0; j < nzb; ++j) { const double t = A[k + nzb * j]; A[k + nzb * j] = A[p + nzb * j]; A[p + nzb * j] = t; }
    const double inv = 1.0 / A[k + nzb * k];
    for (int i = k + 1; i < nzb; ++i) A[i + nzb * k] *= inv;
    for (int j = k + 1; j < nzb; ++j) { const double ukj = A[k + nzb * j]; for (int i = k + 1; i < nzb; ++i) A[i + nzb * j] -= A[i + nzb * k] * ukj; }
  }
  for (int a = 0; a < ny; ++a) {               // G symmetric: row a of M solves G x = R(a,:)'
    for (int c = 0; c < nzb; ++c) x[c] = g[nzb * nzb + c + nzb * a];
    for (int k = 0; k < nzb; ++k) { const int p = piv[k]; if (p != k) { const double t = x[k]; x[k] = x[p]; x[p] = t; } }
    for (int k = 0; k < nzb; ++k) for (int i = k + 1; i < nzb; ++i) x[i] -= A[i + nzb * k] * x[k];
    for (int k = nzb - 1; k >= 0; --k) { x[k] /= A[k + nzb * k]; for (int i = 0; i < k; ++i) x[i] -= A[i + nzb * k] * x[k]; }
    for (int c = 0; c < nzb; ++c) { dz.at(a + ny * c, b) = x[c]; chk += x[c] * 0.0; }
  }
  if (status.ok()) status.at(0, b) = (ok && chk == chk) ? 1 : 0;     // ... and the fit is finite (a sample may carry a failed solve's NaN)
}


// the same fit with compile-time sizes (the models' nzb = 2nq + nu and ny = nq): the nzb x nzb LU stays in
// registers instead of dynamically indexed scratch (planar push 12 x 12: 0.28 ms -> a few microseconds)
template <int NZB, int NY>
__global__ __launch_bounds__(OD_BLOCK) void k_ls_solve_fixed(long B, const double* acc, View<double> dz, View<int> status) {
  const long b = (long)blockIdx.x * OD_BLOCK + threadIdx.x;
  if (b >= B) return;
  constexpr int ne = NZB * (NZB + NY);
  double A[NZB * NZB], x[NZB];
  int piv[NZB];
  const double* g = acc + b * ne;
#pragma unroll
  for (int i = 0; i < NZB * NZB; ++i) A[i] = g[i];
  const bool ok = od_lu_factor<double, NZB>(A, piv);
  double chk = 0.0;
#pragma unroll
  for (int a = 0; a < NY; ++a) {               // G symmetric: row a of M solves G x = R(a,:)'
#pragma unroll
    for (int c = 0; c < NZB; ++c) x[c] = g[NZB * NZB + c + NZB * a];
    od_lu_solve<double, NZB>(A, piv, x);
#pragma unroll
    for (int c = 0; c < NZB; ++c) { dz.at(a + NY * c, b) = x[c]; chk += x[c] * 0.0; }
  }
  if (status.ok()) status.at(0, b) = (ok && chk == chk) ? 1 : 0;
}

// the whole fit of one knot in ONE workgroup (compile-time sizes): the N samples are staged in LDS in chunks, thread e owns
// entry e of [G | R'] (the same N-term sums, in the same order, as k_ls_accumulate), then thread a < NY factors G in
// registers and solves for row a of M.  One launch instead of two, no round trip of the normal equations through HBM:
// BASELINE config 3 (50 knots, N = 256): 62 + 18 us -> see profiles/r3_planar_push_coop3.json.
#if defined(__HIPCC__)
template <int NZB, int NY>
__global__ __launch_bounds__(256) void k_ls_fit_fused(long B, int N, const double* eta, View<const double> feta, View<double> dz, View<int> status) {
  constexpr int NE = NZB * (NZB + NY), CH = 256, LD = CH + 1;
  __shared__ double s_eta[NZB * LD], s_df[NY * LD], s_acc[NE];
  const long b = blockIdx.x;
  const int t = threadIdx.x;
  const long p0 = b * (N + 1);
  const int r = t % NZB, c = t / NZB;
  double fz[NY];
#pragma unroll
  for (int a = 0; a < NY; ++a) fz[a] = feta.at(a, p0);
  double acc = 0.0;
  for (int i0 = 0; i0 < N; i0 += CH) {
    const int i = i0 + t;
    __syncthreads();
    if (i < N) {
#pragma unroll
      for (int k = 0; k < NZB; ++k) s_eta[k * LD + t] = eta[k + (long)NZB * i];
#pragma unroll
      for (int a = 0; a < NY; ++a) s_df[a * LD + t] = feta.at(a, p0 + 1 + i) - fz[a];
    }
    __syncthreads();
    if (t < NE) {
      const int n = (N - i0 < CH) ? N - i0 : CH;
      const double* u = s_eta + r * LD;
      const double* v = (c < NZB) ? s_eta + c * LD : s_df + (c - NZB) * LD;
      for (int k = 0; k < n; ++k) acc += u[k] * v[k];
    }
  }
  if (t < NE) s_acc[t] = acc;
  __syncthreads();
  if (t < NY) {
    double A[NZB * NZB], x[NZB];
    int piv[NZB];
#pragma unroll
    for (int k = 0; k < NZB * NZB; ++k) A[k] = s_acc[k];
    const bool ok = od_lu_factor<double, NZB>(A, piv);
#pragma unroll
    for (int k = 0; k < NZB; ++k) x[k] = s_acc[NZB * NZB + k + NZB * t];       // G symmetric: row t of M solves G x = R(t,:)'
    od_lu_solve<double, NZB>(A, piv, x);
    double chk = 0.0;
#pragma unroll
    for (int k = 0; k < NZB; ++k) { dz.at(t + NY * k, b) = x[k]; chk += x[k] * 0.0; }
    // status: 1 = Gram matrix factorised and every row of the fit finite (a sample may carry a failed solve's NaN)
    const bool fine = ok && chk == chk;
    const unsigned long long m = __builtin_amdgcn_ballot_w64(!fine);
    if (t == 0 && status.ok()) status.at(0, b) = (m & ((1ull << NY) - 1ull)) ? 0 : 1;
  }
}
#endif

// ---- iLQR backward pass (Riccati recursion), one lane per trajectory; runtime sizes n <= 16, m <= 12 --------
// Gauss-Newton iLQR with the quadratic cost model supplied per knot (IterativeLQR's backward pass as recalled,
// SURVEY.md Appendix A; first-order dynamics only):
//   Qx = lx + A'Vx, Qu = lu + B'Vx, Qxx = lxx + A'Vxx A, Quu = luu + B'Vxx B + reg I, Qux = lux + B'Vxx A
//   K = -Quu^{-1} Qux, k = -Quu^{-1} Qu, Vx = Qx + K'Quu k + K'Qu + Qux'k, Vxx = Qxx + K'Quu K + K'Qux + Qux'K
constexpr int OD_IL_N = 16, OD_IL_M = 12;
// TA: element type of the linearisation (A, Bm) and of the gains (K, k) -- double, or float when they come from / go to the
// single-precision rocket kernels (the device-resident iteration, od_ilqr_solver.inc); the recursion itself runs in double.
template <class TA> struct IlqrArgsT {
  long B; int T, n, m; double reg;
  View<const TA> A, Bm;                              // per knot (T*B)
  View<const double> lxx, luu, lux, lx, lu;          // per knot (a view with sb = 0 shares one matrix between all knots)
  View<const double> VxxT, VxT;                      // per trajectory
  View<TA> K, k;                                     // per knot: m x n col-major, m
  View<double> dV;                                   // per trajectory: [sum k'Qu, sum 0.5 k'Quu k]
  View<int> status;                                  // per trajectory: 1 = every Quu factorised (positive pivots)
  // device-resident iteration (all null / 0 for od_ilqr_backward):
  const double* reg_dev;                             // regularisation per trajectory, read from device memory instead of `reg`
  const int* done_b;                                 // (may be null) per trajectory: non-zero = nothing to do for it
  int retry;                                         // 1: a trajectory whose Quu + reg I is not positive definite repeats ITS recursion with
                                                     // reg <- min(max(reg, 1e-8) * 10, 1e6) until it factorises; still failing at 1e6: K = k = dV = 0, status 0
  const int* skip;                                   // *skip != 0: the launch does nothing
};
using IlqrArgs = IlqrArgsT<double>;
OD_HD double od_il_next_reg(double r) { return fmin(fmax(r, 1e-8) * 10.0, 1e6); }
#if defined(__HIPCC__)
// One 256-thread workgroup per trajectory, matrices in LDS, thread (i, j) = tid % rows, tid / rows owns one entry of
// each small product (n <= 16: n*n <= 256 entries).  The first version ran one LANE per trajectory with dynamically
// indexed private arrays (scratch): 53 ms per call for the rocket (n = 12, m = 3, T = 60, 4096 trajectories) -- 77 % of an
// iLQR iteration; this one: see profiles/r1_ilqr.json.
constexpr int OD_IL_THREADS = 256;
// (NC, MC: the sizes as compile-time constants -- the substitution's right-hand side y[] then lives in registers instead of scratch
// memory and the loops over m unroll: the parameter stage's embedded model, od_ilqr_solver.inc::ilp_backward; 0, 0: any size)
template <class TA, int NC = 0, int MC = 0> __global__ __launch_bounds__(OD_IL_THREADS) void k_ilqr_backward(IlqrArgsT<TA> a) {
  if (a.skip && *a.skip) return;
  const long b = blockIdx.x;
  const int n = NC ? NC : a.n, m = NC ? MC : a.m, tid = threadIdx.x;
  __shared__ double Vxx[OD_IL_N * OD_IL_N], At[OD_IL_N * OD_IL_N], W[OD_IL_N * OD_IL_N], Qxx[OD_IL_N * OD_IL_N];
  __shared__ double Bt[OD_IL_N * OD_IL_M], WB[OD_IL_N * OD_IL_M], Qux[OD_IL_M * OD_IL_N], Kt[OD_IL_M * OD_IL_N], QK[OD_IL_M * OD_IL_N];
  __shared__ double Quu[OD_IL_M * OD_IL_M], L[OD_IL_M * OD_IL_M];
  __shared__ double Vx[OD_IL_N], Qx[OD_IL_N], Qu[OD_IL_M], kt[OD_IL_M], Quuk[OD_IL_M];
  __shared__ double dV[2];
  __shared__ int okflag;
  const int nn = n * n, nm = n * m, mm = m * m;
  // entry owned by this thread in n x n, n x m (rows n) and m x n (rows m) matrices
  const int in_ = tid % n, jn_ = tid / n;         // valid if tid < nn (n x n) or tid < nm (n x m: column jn_ < m)
  const int im_ = tid % m, jm_ = tid / m;         // valid if tid < nm (m x n: column jm_ < n) or tid < mm (m x m)
  if (a.done_b && a.done_b[b]) return;
  double reg = a.reg_dev ? a.reg_dev[b] : a.reg;
  for (;;) {                                  // (one pass unless a.retry)
  if (tid < nn) Vxx[tid] = a.VxxT.at(tid, b);
  if (tid < n) Vx[tid] = a.VxT.at(tid, b);
  if (tid == 0) { dV[0] = 0.0; dV[1] = 0.0; okflag = 1; }
  __syncthreads();
  for (int t = a.T - 1; t >= 0; --t) {
    const long kk = (long)t * a.B + b;
    if (tid < nn) At[tid] = (double)a.A.at(tid, kk);
    if (tid < nm) Bt[tid] = (double)a.Bm.at(tid, kk);
    __syncthreads();
    // W = Vxx A (n x n), WB = Vxx B (n x m)
    if (tid < nn) { double s = 0; for (int l = 0; l < n; ++l) s += Vxx[in_ + n * l] * At[l + n * jn_]; W[tid] = s; }
    if (tid < nm) { double s = 0; for (int l = 0; l < n; ++l) s += Vxx[in_ + n * l] * Bt[l + n * jn_]; WB[tid] = s; }
    __syncthreads();
    // Qxx = lxx + A'W, Qux = lux + B'W, Quu = luu + B'WB, Qx = lx + A'Vx, Qu = lu + B'Vx
    if (tid < nn) { double s = a.lxx.at(tid, kk); for (int l = 0; l < n; ++l) s += At[l + n * in_] * W[l + n * jn_]; Qxx[tid] = s; }
    if (tid < nm) { double s = a.lux.at(tid, kk); for (int l = 0; l < n; ++l) s += Bt[l + n * im_] * W[l + n * jm_]; Qux[tid] = s; }
    if (tid < mm) { double s = a.luu.at(tid, kk); for (int l = 0; l < n; ++l) s += Bt[l + n * im_] * WB[l + n * jm_]; Quu[tid] = s; }
    if (tid < n) { double s = a.lx.at(tid, kk); for (int l = 0; l < n; ++l) s += At[l + n * tid] * Vx[l]; Qx[tid] = s; }
    if (tid < m) { double s = a.lu.at(tid, kk); for (int l = 0; l < n; ++l) s += Bt[l + n * tid] * Vx[l]; Qu[tid] = s; }
    __syncthreads();
    // Cholesky of Quu + reg I (m <= 12: one thread)
    if (tid == 0) {
      for (int i = 0; i < mm; ++i) L[i] = Quu[i];
      for (int i = 0; i < m; ++i) L[i + m * i] += reg;
      for (int j = 0; j < m; ++j) {
        double d = L[j + m * j];
        for (int l = 0; l < j; ++l) d -= L[j + m * l] * L[j + m * l];
        if (!(d > 0.0)) { okflag = 0; d = 1e-12; }
        d = sqrt(d);
        L[j + m * j] = d;
        for (int i = j + 1; i < m; ++i) { double sx = L[i + m * j]; for (int l = 0; l < j; ++l) sx -= L[i + m * l] * L[j + m * l]; L[i + m * j] = sx / d; }
      }
    }
    __syncthreads();
    if (a.retry && !okflag) break;            // (uniform: okflag is shared) this pass is thrown away
    // K = -(Quu+reg)^{-1} Qux (one thread per column), k = -(Quu+reg)^{-1} Qu (thread n)
    if (tid <= n) {
      const int c = tid;
      double y[MC ? MC : OD_IL_M];
#pragma unroll
      for (int i = 0; i < (MC ? MC : m); ++i) y[i] = (c < n) ? Qux[i + m * c] : Qu[i];
#pragma unroll
      for (int i = 0; i < (MC ? MC : m); ++i) {
        double sx = y[i];
#pragma unroll
        for (int l = 0; l < i; ++l) sx -= L[i + m * l] * y[l];
        y[i] = sx / L[i + m * i];
      }
#pragma unroll
      for (int i = (MC ? MC : m) - 1; i >= 0; --i) {
        double sx = y[i];
#pragma unroll
        for (int l = i + 1; l < (MC ? MC : m); ++l) sx -= L[l + m * i] * y[l];
        y[i] = sx / L[i + m * i];
      }
#pragma unroll
      for (int i = 0; i < (MC ? MC : m); ++i) { if (c < n) Kt[i + m * c] = -y[i]; else kt[i] = -y[i]; }
    }
    __syncthreads();
    if (tid < nm) a.K.at(tid, kk) = (TA)Kt[tid];
    if (tid < m) {
      a.k.at(tid, kk) = (TA)kt[tid];
      double sx = 0; for (int l = 0; l < m; ++l) sx += Quu[tid + m * l] * kt[l];
      Quuk[tid] = sx;                                             // Quu k (Quu without reg, as in the cost-to-go expansion)
    }
    // QK = Quu K (m x n)
    if (tid < nm) { double sx = 0; for (int l = 0; l < m; ++l) sx += Quu[im_ + m * l] * Kt[l + m * jm_]; QK[tid] = sx; }
    __syncthreads();
    if (tid == 0) {
      double d1 = 0, d2 = 0;
      for (int i = 0; i < m; ++i) { d1 += kt[i] * Qu[i]; d2 += 0.5 * kt[i] * Quuk[i]; }
      dV[0] += d1; dV[1] += d2;
    }
    // value function update
    double vx_new = 0, vxx_new = 0;
    if (tid < n) {
      double sx = Qx[tid];
      for (int l = 0; l < m; ++l) sx += Kt[l + m * tid] * (Quuk[l] + Qu[l]) + Qux[l + m * tid] * kt[l];
      vx_new = sx;
    }
    if (tid < nn) {
      double sx = Qxx[tid];
      for (int l = 0; l < m; ++l) sx += Kt[l + m * in_] * (QK[l + m * jn_] + Qux[l + m * jn_]) + Qux[l + m * in_] * Kt[l + m * jn_];
      vxx_new = sx;
    }
    __syncthreads();
    if (tid < n) Vx[tid] = vx_new;
    if (tid < nn) Vxx[tid] = vxx_new;
    __syncthreads();
    if (tid < nn && in_ < jn_) { const double sx = 0.5 * (Vxx[in_ + n * jn_] + Vxx[jn_ + n * in_]); Vxx[in_ + n * jn_] = sx; Vxx[jn_ + n * in_] = sx; }
    __syncthreads();
  }
  if (!a.retry || okflag || !(reg < 1e6)) break;
  reg = od_il_next_reg(reg);
  __syncthreads();                            // (everybody has read okflag before thread 0 resets it)
  }
  if (a.retry && !okflag) {                   // not factorisable at any regularisation (non-finite data): no step for this trajectory
    for (int t = 0; t < a.T; ++t) {
      const long kk = (long)t * a.B + b;
      if (tid < nm) a.K.at(tid, kk) = TA(0);
      if (tid < m) a.k.at(tid, kk) = TA(0);
    }
    if (tid == 0) { dV[0] = 0.0; dV[1] = 0.0; }
  }
  if (tid == 0) {
    a.dV.at(0, b) = dV[0];
    a.dV.at(1, b) = dV[1];
    if (a.status.ok()) a.status.at(0, b) = okflag;
  }
}


// Round 3: TB CONSECUTIVE TRAJECTORIES PER WORKGROUP.  The kernel above reads one trajectory's matrices: in the batch-minor
// layout (element e of knot k at e*K + k) its 256 threads touch 256 different 64-byte segments for 8 useful bytes each --
// 470 loads per knot, 8x the useful HBM traffic, and that traffic was its time (rocket n = 12, m = 3, T = 60, 4096
// trajectories: 2.54 ms per call, 35 % of an iLQR iteration of BASELINE config 5).  Here thread (s, j) = (tid / TB, tid % TB)
// works on trajectory b0 + j, so the TB threads of a slot read one full segment; every matrix lives in LDS as [entry][j];
// the entries of an operation are dealt to the 256 / TB slots.  Same sums in the same order as above: same results.
// (N, M: the sizes as compile-time constants for the models in use -- entry -> (row, column) becomes shifts and multiplies instead
// of integer divisions and the inner products unroll; 0, 0: any size)
template <int TB, int N = 0, int M = 0>
__global__ __launch_bounds__(OD_IL_THREADS) void k_ilqr_backward_tb(IlqrArgs a) {
  extern __shared__ double od_il_lds[];
  constexpr int NS = OD_IL_THREADS / TB;
  const int n = N ? N : a.n, m = N ? M : a.m, tid = threadIdx.x, j = tid % TB, s = tid / TB;
  const int nn = n * n, nm = n * m, mm = m * m;
  const long b = (long)blockIdx.x * TB + j;
  const bool live = b < a.B;
  const long bb = live ? b : a.B - 1;                   // (threads past the batch repeat its last trajectory, store nothing)
  double* p = od_il_lds;
  auto take = [&](int cnt) { double* q = p; p += (size_t)cnt * TB; return q; };
  double *Vxx = take(nn), *At = take(nn), *W = take(nn), *Qxx = take(nn), *Bt = take(nm), *WB = take(nm), *Qux = take(nm), *Kt = take(nm),
         *QK = take(nm), *Quu = take(mm), *L = take(mm), *Vx = take(n), *Qx = take(n), *Qu = take(m), *kt = take(m), *Quuk = take(m);
  double* dVs = take(2);
  int* okf = (int*)take(1);
#define OD_E(M_, e) M_[(e) * TB + j]
  for (int e = s; e < nn; e += NS) OD_E(Vxx, e) = a.VxxT.at(e, bb);
  for (int e = s; e < n; e += NS) OD_E(Vx, e) = a.VxT.at(e, bb);
  if (s == 0) { OD_E(dVs, 0) = 0.0; OD_E(dVs, 1) = 0.0; okf[j] = 1; }
  __syncthreads();
  for (int t = a.T - 1; t >= 0; --t) {
    const long kk = (long)t * a.B + bb;
    for (int e = s; e < nn; e += NS) OD_E(At, e) = a.A.at(e, kk);
    for (int e = s; e < nm; e += NS) OD_E(Bt, e) = a.Bm.at(e, kk);
    __syncthreads();
    // W = Vxx A (n x n), WB = Vxx B (n x m)
    for (int e = s; e < nn; e += NS) { const int i = e % n, c = e / n; double sm = 0; for (int l = 0; l < n; ++l) sm += OD_E(Vxx, i + n * l) * OD_E(At, l + n * c); OD_E(W, e) = sm; }
    for (int e = s; e < nm; e += NS) { const int i = e % n, c = e / n; double sm = 0; for (int l = 0; l < n; ++l) sm += OD_E(Vxx, i + n * l) * OD_E(Bt, l + n * c); OD_E(WB, e) = sm; }
    __syncthreads();
    // Qxx = lxx + A'W, Qux = lux + B'W, Quu = luu + B'WB, Qx = lx + A'Vx, Qu = lu + B'Vx
    for (int e = s; e < nn; e += NS) { const int i = e % n, c = e / n; double sm = a.lxx.at(e, kk); for (int l = 0; l < n; ++l) sm += OD_E(At, l + n * i) * OD_E(W, l + n * c); OD_E(Qxx, e) = sm; }
    for (int e = s; e < nm; e += NS) { const int i = e % m, c = e / m; double sm = a.lux.at(e, kk); for (int l = 0; l < n; ++l) sm += OD_E(Bt, l + n * i) * OD_E(W, l + n * c); OD_E(Qux, e) = sm; }
    for (int e = s; e < mm; e += NS) { const int i = e % m, c = e / m; double sm = a.luu.at(e, kk); for (int l = 0; l < n; ++l) sm += OD_E(Bt, l + n * i) * OD_E(WB, l + n * c); OD_E(Quu, e) = sm; }
    for (int e = s; e < n; e += NS) { double sm = a.lx.at(e, kk); for (int l = 0; l < n; ++l) sm += OD_E(At, l + n * e) * OD_E(Vx, l); OD_E(Qx, e) = sm; }
    for (int e = s; e < m; e += NS) { double sm = a.lu.at(e, kk); for (int l = 0; l < n; ++l) sm += OD_E(Bt, l + n * e) * OD_E(Vx, l); OD_E(Qu, e) = sm; }
    __syncthreads();
    // Cholesky of Quu + reg I (m <= 12: one thread per trajectory)
    if (s == 0) {
      for (int i = 0; i < mm; ++i) OD_E(L, i) = OD_E(Quu, i);
      for (int i = 0; i < m; ++i) OD_E(L, i + m * i) += a.reg;
      for (int c = 0; c < m; ++c) {
        double d = OD_E(L, c + m * c);
        for (int l = 0; l < c; ++l) d -= OD_E(L, c + m * l) * OD_E(L, c + m * l);
        if (!(d > 0.0)) { okf[j] = 0; d = 1e-12; }
        d = sqrt(d);
        OD_E(L, c + m * c) = d;
        for (int i = c + 1; i < m; ++i) { double sx = OD_E(L, i + m * c); for (int l = 0; l < c; ++l) sx -= OD_E(L, i + m * l) * OD_E(L, c + m * l); OD_E(L, i + m * c) = sx / d; }
      }
    }
    __syncthreads();
    // K = -(Quu+reg)^{-1} Qux (column c, in place in Kt), k = -(Quu+reg)^{-1} Qu (c = n, in place in kt)
    for (int c = s; c <= n; c += NS) {
      double* y = (c < n) ? (Kt + (size_t)m * c * TB) : kt;
      for (int i = 0; i < m; ++i) y[i * TB + j] = (c < n) ? OD_E(Qux, i + m * c) : OD_E(Qu, i);
      for (int i = 0; i < m; ++i) { double sx = y[i * TB + j]; for (int l = 0; l < i; ++l) sx -= OD_E(L, i + m * l) * y[l * TB + j]; y[i * TB + j] = sx / OD_E(L, i + m * i); }
      for (int i = m - 1; i >= 0; --i) { double sx = y[i * TB + j]; for (int l = i + 1; l < m; ++l) sx -= OD_E(L, l + m * i) * y[l * TB + j]; y[i * TB + j] = sx / OD_E(L, i + m * i); }
      for (int i = 0; i < m; ++i) y[i * TB + j] = -y[i * TB + j];
    }
    __syncthreads();
    if (live) {
      for (int e = s; e < nm; e += NS) a.K.at(e, kk) = OD_E(Kt, e);
      for (int e = s; e < m; e += NS) a.k.at(e, kk) = OD_E(kt, e);
    }
    for (int e = s; e < m; e += NS) { double sx = 0; for (int l = 0; l < m; ++l) sx += OD_E(Quu, e + m * l) * OD_E(kt, l); OD_E(Quuk, e) = sx; }   // Quu k (Quu without reg)
    for (int e = s; e < nm; e += NS) { const int i = e % m, c = e / m; double sx = 0; for (int l = 0; l < m; ++l) sx += OD_E(Quu, i + m * l) * OD_E(Kt, l + m * c); OD_E(QK, e) = sx; }
    __syncthreads();
    if (s == 0) {
      double d1 = 0, d2 = 0;
      for (int i = 0; i < m; ++i) { d1 += OD_E(kt, i) * OD_E(Qu, i); d2 += 0.5 * OD_E(kt, i) * OD_E(Quuk, i); }
      OD_E(dVs, 0) += d1; OD_E(dVs, 1) += d2;
    }
    // value function update: new Vx into Qx's place (Qx is read by its own entry only), new Vxx into W
    for (int e = s; e < n; e += NS) {
      double sx = OD_E(Qx, e);
      for (int l = 0; l < m; ++l) sx += OD_E(Kt, l + m * e) * (OD_E(Quuk, l) + OD_E(Qu, l)) + OD_E(Qux, l + m * e) * OD_E(kt, l);
      OD_E(Vx, e) = sx;
    }
    for (int e = s; e < nn; e += NS) {
      const int i = e % n, c = e / n;
      double sx = OD_E(Qxx, e);
      for (int l = 0; l < m; ++l) sx += OD_E(Kt, l + m * i) * (OD_E(QK, l + m * c) + OD_E(Qux, l + m * c)) + OD_E(Qux, l + m * i) * OD_E(Kt, l + m * c);
      OD_E(W, e) = sx;
    }
    __syncthreads();
    for (int e = s; e < nn; e += NS) {
      const int i = e % n, c = e / n;
      OD_E(Vxx, e) = (i == c) ? OD_E(W, e) : 0.5 * (OD_E(W, i + n * c) + OD_E(W, c + n * i));
    }
    __syncthreads();
  }
  if (s == 0 && live) {
    a.dV.at(0, b) = OD_E(dVs, 0);
    a.dV.at(1, b) = OD_E(dVs, 1);
    if (a.status.ok()) a.status.at(0, b) = okf[j];
  }
#undef OD_E
}
// ONE TRAJECTORY PER 16-LANE DPP ROW (round 3; batch-minor data, m <= 4).  Lane i of the row keeps ROW i of every n-column
// matrix in registers (Vxx, A, A', W = Vxx A, Qxx, the new Vxx) and row i of the m-column ones (B, W B, Qux', K'); an inner product
// over the rows of another matrix reads them from their lanes with `v_fmac_f64_dpp row_newbcast` -- the instruction of the
// cooperative solve kernels (od_coop.h) -- up to sixteen to an asm block behind one `s_nop 1` (the DPP read-after-write hazard is not
// interlocked on gfx950).  The m x m block (Quu, its Cholesky factor, k) is replicated in every lane.  Transposes go through a padded
// LDS tile per row without any workgroup barrier (see the kernel); 16 trajectories per workgroup, four
// per wavefront: 4096 trajectories are 1024 wavefronts, one per SIMD.  Sums run over the same terms as the kernels above in the
// same order, fused (fmac) where those round the product first: results agree to rounding, not bit for bit.
// acc[c] += sum_{l = L0 .. L0 + LW - 1} (y[c] of lane l) * x[l]  for c = 0 .. CW - 1: one asm block, the CW accumulators interleaved
// (the compiler cannot interleave separate asm blocks; measured neutral for a lone wavefront, profiles/r3_riccati_kernels.txt)
template <int CW, int LW, int L0> __device__ __forceinline__ void od_row_block(double* acc, const double* y, const double* x) {
  if constexpr (CW == 4 && LW == 4) {
      asm("s_nop 1\n\t"
          "v_fmac_f64_dpp %0, %4, %8 row_newbcast:%12 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %5, %8 row_newbcast:%12 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %2, %6, %8 row_newbcast:%12 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %3, %7, %8 row_newbcast:%12 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %4, %9 row_newbcast:%13 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %5, %9 row_newbcast:%13 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %2, %6, %9 row_newbcast:%13 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %3, %7, %9 row_newbcast:%13 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %4, %10 row_newbcast:%14 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %5, %10 row_newbcast:%14 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %2, %6, %10 row_newbcast:%14 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %3, %7, %10 row_newbcast:%14 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %4, %11 row_newbcast:%15 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %5, %11 row_newbcast:%15 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %2, %6, %11 row_newbcast:%15 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %3, %7, %11 row_newbcast:%15 row_mask:0xf bank_mask:0xf"
          : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]) : "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(x[L0 + 0]), "v"(x[L0 + 1]), "v"(x[L0 + 2]), "v"(x[L0 + 3]), "n"(L0 + 0), "n"(L0 + 1), "n"(L0 + 2), "n"(L0 + 3));
  }   else if constexpr (CW == 4 && LW == 2) {
      asm("s_nop 1\n\t"
          "v_fmac_f64_dpp %0, %4, %8 row_newbcast:%10 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %5, %8 row_newbcast:%10 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %2, %6, %8 row_newbcast:%10 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %3, %7, %8 row_newbcast:%10 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %4, %9 row_newbcast:%11 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %5, %9 row_newbcast:%11 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %2, %6, %9 row_newbcast:%11 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %3, %7, %9 row_newbcast:%11 row_mask:0xf bank_mask:0xf"
          : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]) : "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(x[L0 + 0]), "v"(x[L0 + 1]), "n"(L0 + 0), "n"(L0 + 1));
  }   else if constexpr (CW == 4 && LW == 1) {
      asm("s_nop 1\n\t"
          "v_fmac_f64_dpp %0, %4, %8 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %5, %8 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %2, %6, %8 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %3, %7, %8 row_newbcast:%9 row_mask:0xf bank_mask:0xf"
          : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]) : "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(x[L0 + 0]), "n"(L0 + 0));
  }   else if constexpr (CW == 3 && LW == 4) {
      asm("s_nop 1\n\t"
          "v_fmac_f64_dpp %0, %3, %6 row_newbcast:%10 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %4, %6 row_newbcast:%10 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %2, %5, %6 row_newbcast:%10 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %3, %7 row_newbcast:%11 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %4, %7 row_newbcast:%11 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %2, %5, %7 row_newbcast:%11 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %3, %8 row_newbcast:%12 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %4, %8 row_newbcast:%12 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %2, %5, %8 row_newbcast:%12 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %3, %9 row_newbcast:%13 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %4, %9 row_newbcast:%13 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %2, %5, %9 row_newbcast:%13 row_mask:0xf bank_mask:0xf"
          : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]) : "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(x[L0 + 0]), "v"(x[L0 + 1]), "v"(x[L0 + 2]), "v"(x[L0 + 3]), "n"(L0 + 0), "n"(L0 + 1), "n"(L0 + 2), "n"(L0 + 3));
  }   else if constexpr (CW == 3 && LW == 2) {
      asm("s_nop 1\n\t"
          "v_fmac_f64_dpp %0, %3, %6 row_newbcast:%8 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %4, %6 row_newbcast:%8 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %2, %5, %6 row_newbcast:%8 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %3, %7 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %4, %7 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %2, %5, %7 row_newbcast:%9 row_mask:0xf bank_mask:0xf"
          : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]) : "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(x[L0 + 0]), "v"(x[L0 + 1]), "n"(L0 + 0), "n"(L0 + 1));
  }   else if constexpr (CW == 3 && LW == 1) {
      asm("s_nop 1\n\t"
          "v_fmac_f64_dpp %0, %3, %6 row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %4, %6 row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %2, %5, %6 row_newbcast:%7 row_mask:0xf bank_mask:0xf"
          : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]) : "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(x[L0 + 0]), "n"(L0 + 0));
  }   else if constexpr (CW == 2 && LW == 4) {
      asm("s_nop 1\n\t"
          "v_fmac_f64_dpp %0, %2, %4 row_newbcast:%8 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %3, %4 row_newbcast:%8 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %2, %5 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %3, %5 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %2, %6 row_newbcast:%10 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %3, %6 row_newbcast:%10 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %2, %7 row_newbcast:%11 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %3, %7 row_newbcast:%11 row_mask:0xf bank_mask:0xf"
          : "+v"(acc[0]), "+v"(acc[1]) : "v"(y[0]), "v"(y[1]), "v"(x[L0 + 0]), "v"(x[L0 + 1]), "v"(x[L0 + 2]), "v"(x[L0 + 3]), "n"(L0 + 0), "n"(L0 + 1), "n"(L0 + 2), "n"(L0 + 3));
  }   else if constexpr (CW == 2 && LW == 2) {
      asm("s_nop 1\n\t"
          "v_fmac_f64_dpp %0, %2, %4 row_newbcast:%6 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %3, %4 row_newbcast:%6 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %2, %5 row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %3, %5 row_newbcast:%7 row_mask:0xf bank_mask:0xf"
          : "+v"(acc[0]), "+v"(acc[1]) : "v"(y[0]), "v"(y[1]), "v"(x[L0 + 0]), "v"(x[L0 + 1]), "n"(L0 + 0), "n"(L0 + 1));
  }   else if constexpr (CW == 2 && LW == 1) {
      asm("s_nop 1\n\t"
          "v_fmac_f64_dpp %0, %2, %4 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %3, %4 row_newbcast:%5 row_mask:0xf bank_mask:0xf"
          : "+v"(acc[0]), "+v"(acc[1]) : "v"(y[0]), "v"(y[1]), "v"(x[L0 + 0]), "n"(L0 + 0));
  }   else if constexpr (CW == 1 && LW == 4) {
      asm("s_nop 1\n\t"
          "v_fmac_f64_dpp %0, %1, %2 row_newbcast:%6 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %1, %3 row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %1, %4 row_newbcast:%8 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %1, %5 row_newbcast:%9 row_mask:0xf bank_mask:0xf"
          : "+v"(acc[0]) : "v"(y[0]), "v"(x[L0 + 0]), "v"(x[L0 + 1]), "v"(x[L0 + 2]), "v"(x[L0 + 3]), "n"(L0 + 0), "n"(L0 + 1), "n"(L0 + 2), "n"(L0 + 3));
  }   else if constexpr (CW == 1 && LW == 2) {
      asm("s_nop 1\n\t"
          "v_fmac_f64_dpp %0, %1, %2 row_newbcast:%4 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %1, %3 row_newbcast:%5 row_mask:0xf bank_mask:0xf"
          : "+v"(acc[0]) : "v"(y[0]), "v"(x[L0 + 0]), "v"(x[L0 + 1]), "n"(L0 + 0), "n"(L0 + 1));
  }   else if constexpr (CW == 1 && LW == 1) {
      asm("s_nop 1\n\t"
          "v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
          : "+v"(acc[0]) : "v"(y[0]), "v"(x[L0 + 0]), "n"(L0 + 0));
  }
}
// acc[0 .. NC-1] += sum_{l < NL} (y[c] of lane l) * x[l]
template <int NC, int NL, int C0 = 0, int L0 = 0> __device__ __forceinline__ void od_row_mac(double* acc, const double* y, const double* x) {
  if constexpr (C0 < NC) {
    constexpr int CW = (NC - C0) >= 4 ? 4 : (NC - C0);
    if constexpr (L0 < NL) {
      constexpr int LW = (NL - L0) >= 4 ? 4 : ((NL - L0) >= 2 ? 2 : 1);
      od_row_block<CW, LW, L0>(acc + C0, y + C0, x);
      od_row_mac<NC, NL, C0, L0 + LW>(acc, y, x);
    } else {
      od_row_mac<NC, NL, C0 + CW, 0>(acc, y, x);
    }
  }
}
template <int L> __device__ __forceinline__ double od_row_bc(double x) {
  double r;
  asm("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x), "n"(L));
  return r;
}
template <int C0, int N, int M> __device__ __forceinline__ void od_row_vxx_terms(double* Wn, const double* G, const double* KT, const double* QuxT) {
  // Wn[c] += sum_j K'[i][j] (G[j] of lane c) + Qux'[i][j] (K'[j] of lane c),  c = C0 .. N - 1; four columns to an asm block
  if constexpr (C0 < N) {
    constexpr int CW = (N - C0) >= 4 ? 4 : (N - C0);
    static_assert(M >= 1 && M <= 4, "");
    if constexpr (CW == 4 && M == 1) {
      asm("s_nop 1\n\t"
          "v_fmac_f64_dpp %0, %4, %5 row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %4, %5 row_newbcast:%8 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %2, %4, %5 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %3, %4, %5 row_newbcast:%10 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %5, %6 row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %5, %6 row_newbcast:%8 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %2, %5, %6 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %3, %5, %6 row_newbcast:%10 row_mask:0xf bank_mask:0xf"
          : "+v"(Wn[C0 + 0]), "+v"(Wn[C0 + 1]), "+v"(Wn[C0 + 2]), "+v"(Wn[C0 + 3]) : "v"(G[0]), "v"(KT[0]), "v"(QuxT[0]), "n"(C0 + 0), "n"(C0 + 1), "n"(C0 + 2), "n"(C0 + 3));
    }     else if constexpr (CW == 4 && M == 2) {
      asm("s_nop 1\n\t"
          "v_fmac_f64_dpp %0, %4, %6 row_newbcast:%10 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %4, %6 row_newbcast:%11 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %2, %4, %6 row_newbcast:%12 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %3, %4, %6 row_newbcast:%13 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %6, %8 row_newbcast:%10 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %6, %8 row_newbcast:%11 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %2, %6, %8 row_newbcast:%12 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %3, %6, %8 row_newbcast:%13 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %5, %7 row_newbcast:%10 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %5, %7 row_newbcast:%11 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %2, %5, %7 row_newbcast:%12 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %3, %5, %7 row_newbcast:%13 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %7, %9 row_newbcast:%10 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %7, %9 row_newbcast:%11 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %2, %7, %9 row_newbcast:%12 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %3, %7, %9 row_newbcast:%13 row_mask:0xf bank_mask:0xf"
          : "+v"(Wn[C0 + 0]), "+v"(Wn[C0 + 1]), "+v"(Wn[C0 + 2]), "+v"(Wn[C0 + 3]) : "v"(G[0]), "v"(G[1]), "v"(KT[0]), "v"(KT[1]), "v"(QuxT[0]), "v"(QuxT[1]), "n"(C0 + 0), "n"(C0 + 1), "n"(C0 + 2), "n"(C0 + 3));
    }     else if constexpr (CW == 4 && M == 3) {
      asm("s_nop 1\n\t"
          "v_fmac_f64_dpp %0, %4, %7 row_newbcast:%13 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %4, %7 row_newbcast:%14 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %2, %4, %7 row_newbcast:%15 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %3, %4, %7 row_newbcast:%16 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %7, %10 row_newbcast:%13 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %7, %10 row_newbcast:%14 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %2, %7, %10 row_newbcast:%15 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %3, %7, %10 row_newbcast:%16 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %5, %8 row_newbcast:%13 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %5, %8 row_newbcast:%14 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %2, %5, %8 row_newbcast:%15 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %3, %5, %8 row_newbcast:%16 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %8, %11 row_newbcast:%13 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %8, %11 row_newbcast:%14 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %2, %8, %11 row_newbcast:%15 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %3, %8, %11 row_newbcast:%16 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %6, %9 row_newbcast:%13 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %6, %9 row_newbcast:%14 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %2, %6, %9 row_newbcast:%15 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %3, %6, %9 row_newbcast:%16 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %9, %12 row_newbcast:%13 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %9, %12 row_newbcast:%14 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %2, %9, %12 row_newbcast:%15 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %3, %9, %12 row_newbcast:%16 row_mask:0xf bank_mask:0xf"
          : "+v"(Wn[C0 + 0]), "+v"(Wn[C0 + 1]), "+v"(Wn[C0 + 2]), "+v"(Wn[C0 + 3]) : "v"(G[0]), "v"(G[1]), "v"(G[2]), "v"(KT[0]), "v"(KT[1]), "v"(KT[2]), "v"(QuxT[0]), "v"(QuxT[1]), "v"(QuxT[2]), "n"(C0 + 0), "n"(C0 + 1), "n"(C0 + 2), "n"(C0 + 3));
    }     else if constexpr (CW == 4 && M == 4) {
      asm("s_nop 1\n\t"
          "v_fmac_f64_dpp %0, %4, %8 row_newbcast:%16 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %4, %8 row_newbcast:%17 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %2, %4, %8 row_newbcast:%18 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %3, %4, %8 row_newbcast:%19 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %8, %12 row_newbcast:%16 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %8, %12 row_newbcast:%17 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %2, %8, %12 row_newbcast:%18 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %3, %8, %12 row_newbcast:%19 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %5, %9 row_newbcast:%16 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %5, %9 row_newbcast:%17 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %2, %5, %9 row_newbcast:%18 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %3, %5, %9 row_newbcast:%19 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %9, %13 row_newbcast:%16 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %9, %13 row_newbcast:%17 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %2, %9, %13 row_newbcast:%18 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %3, %9, %13 row_newbcast:%19 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %6, %10 row_newbcast:%16 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %6, %10 row_newbcast:%17 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %2, %6, %10 row_newbcast:%18 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %3, %6, %10 row_newbcast:%19 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %10, %14 row_newbcast:%16 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %10, %14 row_newbcast:%17 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %2, %10, %14 row_newbcast:%18 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %3, %10, %14 row_newbcast:%19 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %7, %11 row_newbcast:%16 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %7, %11 row_newbcast:%17 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %2, %7, %11 row_newbcast:%18 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %3, %7, %11 row_newbcast:%19 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %11, %15 row_newbcast:%16 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %11, %15 row_newbcast:%17 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %2, %11, %15 row_newbcast:%18 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %3, %11, %15 row_newbcast:%19 row_mask:0xf bank_mask:0xf"
          : "+v"(Wn[C0 + 0]), "+v"(Wn[C0 + 1]), "+v"(Wn[C0 + 2]), "+v"(Wn[C0 + 3]) : "v"(G[0]), "v"(G[1]), "v"(G[2]), "v"(G[3]), "v"(KT[0]), "v"(KT[1]), "v"(KT[2]), "v"(KT[3]), "v"(QuxT[0]), "v"(QuxT[1]), "v"(QuxT[2]), "v"(QuxT[3]), "n"(C0 + 0), "n"(C0 + 1), "n"(C0 + 2), "n"(C0 + 3));
    }     else if constexpr (CW == 3 && M == 1) {
      asm("s_nop 1\n\t"
          "v_fmac_f64_dpp %0, %3, %4 row_newbcast:%6 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %3, %4 row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %2, %3, %4 row_newbcast:%8 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %4, %5 row_newbcast:%6 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %4, %5 row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %2, %4, %5 row_newbcast:%8 row_mask:0xf bank_mask:0xf"
          : "+v"(Wn[C0 + 0]), "+v"(Wn[C0 + 1]), "+v"(Wn[C0 + 2]) : "v"(G[0]), "v"(KT[0]), "v"(QuxT[0]), "n"(C0 + 0), "n"(C0 + 1), "n"(C0 + 2));
    }     else if constexpr (CW == 3 && M == 2) {
      asm("s_nop 1\n\t"
          "v_fmac_f64_dpp %0, %3, %5 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %3, %5 row_newbcast:%10 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %2, %3, %5 row_newbcast:%11 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %5, %7 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %5, %7 row_newbcast:%10 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %2, %5, %7 row_newbcast:%11 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %4, %6 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %4, %6 row_newbcast:%10 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %2, %4, %6 row_newbcast:%11 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %6, %8 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %6, %8 row_newbcast:%10 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %2, %6, %8 row_newbcast:%11 row_mask:0xf bank_mask:0xf"
          : "+v"(Wn[C0 + 0]), "+v"(Wn[C0 + 1]), "+v"(Wn[C0 + 2]) : "v"(G[0]), "v"(G[1]), "v"(KT[0]), "v"(KT[1]), "v"(QuxT[0]), "v"(QuxT[1]), "n"(C0 + 0), "n"(C0 + 1), "n"(C0 + 2));
    }     else if constexpr (CW == 3 && M == 3) {
      asm("s_nop 1\n\t"
          "v_fmac_f64_dpp %0, %3, %6 row_newbcast:%12 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %3, %6 row_newbcast:%13 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %2, %3, %6 row_newbcast:%14 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %6, %9 row_newbcast:%12 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %6, %9 row_newbcast:%13 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %2, %6, %9 row_newbcast:%14 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %4, %7 row_newbcast:%12 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %4, %7 row_newbcast:%13 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %2, %4, %7 row_newbcast:%14 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %7, %10 row_newbcast:%12 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %7, %10 row_newbcast:%13 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %2, %7, %10 row_newbcast:%14 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %5, %8 row_newbcast:%12 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %5, %8 row_newbcast:%13 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %2, %5, %8 row_newbcast:%14 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %8, %11 row_newbcast:%12 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %8, %11 row_newbcast:%13 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %2, %8, %11 row_newbcast:%14 row_mask:0xf bank_mask:0xf"
          : "+v"(Wn[C0 + 0]), "+v"(Wn[C0 + 1]), "+v"(Wn[C0 + 2]) : "v"(G[0]), "v"(G[1]), "v"(G[2]), "v"(KT[0]), "v"(KT[1]), "v"(KT[2]), "v"(QuxT[0]), "v"(QuxT[1]), "v"(QuxT[2]), "n"(C0 + 0), "n"(C0 + 1), "n"(C0 + 2));
    }     else if constexpr (CW == 3 && M == 4) {
      asm("s_nop 1\n\t"
          "v_fmac_f64_dpp %0, %3, %7 row_newbcast:%15 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %3, %7 row_newbcast:%16 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %2, %3, %7 row_newbcast:%17 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %7, %11 row_newbcast:%15 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %7, %11 row_newbcast:%16 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %2, %7, %11 row_newbcast:%17 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %4, %8 row_newbcast:%15 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %4, %8 row_newbcast:%16 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %2, %4, %8 row_newbcast:%17 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %8, %12 row_newbcast:%15 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %8, %12 row_newbcast:%16 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %2, %8, %12 row_newbcast:%17 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %5, %9 row_newbcast:%15 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %5, %9 row_newbcast:%16 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %2, %5, %9 row_newbcast:%17 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %9, %13 row_newbcast:%15 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %9, %13 row_newbcast:%16 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %2, %9, %13 row_newbcast:%17 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %6, %10 row_newbcast:%15 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %6, %10 row_newbcast:%16 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %2, %6, %10 row_newbcast:%17 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %10, %14 row_newbcast:%15 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %10, %14 row_newbcast:%16 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %2, %10, %14 row_newbcast:%17 row_mask:0xf bank_mask:0xf"
          : "+v"(Wn[C0 + 0]), "+v"(Wn[C0 + 1]), "+v"(Wn[C0 + 2]) : "v"(G[0]), "v"(G[1]), "v"(G[2]), "v"(G[3]), "v"(KT[0]), "v"(KT[1]), "v"(KT[2]), "v"(KT[3]), "v"(QuxT[0]), "v"(QuxT[1]), "v"(QuxT[2]), "v"(QuxT[3]), "n"(C0 + 0), "n"(C0 + 1), "n"(C0 + 2));
    }     else if constexpr (CW == 2 && M == 1) {
      asm("s_nop 1\n\t"
          "v_fmac_f64_dpp %0, %2, %3 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %2, %3 row_newbcast:%6 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %3, %4 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %3, %4 row_newbcast:%6 row_mask:0xf bank_mask:0xf"
          : "+v"(Wn[C0 + 0]), "+v"(Wn[C0 + 1]) : "v"(G[0]), "v"(KT[0]), "v"(QuxT[0]), "n"(C0 + 0), "n"(C0 + 1));
    }     else if constexpr (CW == 2 && M == 2) {
      asm("s_nop 1\n\t"
          "v_fmac_f64_dpp %0, %2, %4 row_newbcast:%8 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %2, %4 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %4, %6 row_newbcast:%8 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %4, %6 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %3, %5 row_newbcast:%8 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %3, %5 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %5, %7 row_newbcast:%8 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %5, %7 row_newbcast:%9 row_mask:0xf bank_mask:0xf"
          : "+v"(Wn[C0 + 0]), "+v"(Wn[C0 + 1]) : "v"(G[0]), "v"(G[1]), "v"(KT[0]), "v"(KT[1]), "v"(QuxT[0]), "v"(QuxT[1]), "n"(C0 + 0), "n"(C0 + 1));
    }     else if constexpr (CW == 2 && M == 3) {
      asm("s_nop 1\n\t"
          "v_fmac_f64_dpp %0, %2, %5 row_newbcast:%11 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %2, %5 row_newbcast:%12 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %5, %8 row_newbcast:%11 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %5, %8 row_newbcast:%12 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %3, %6 row_newbcast:%11 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %3, %6 row_newbcast:%12 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %6, %9 row_newbcast:%11 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %6, %9 row_newbcast:%12 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %4, %7 row_newbcast:%11 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %4, %7 row_newbcast:%12 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %7, %10 row_newbcast:%11 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %7, %10 row_newbcast:%12 row_mask:0xf bank_mask:0xf"
          : "+v"(Wn[C0 + 0]), "+v"(Wn[C0 + 1]) : "v"(G[0]), "v"(G[1]), "v"(G[2]), "v"(KT[0]), "v"(KT[1]), "v"(KT[2]), "v"(QuxT[0]), "v"(QuxT[1]), "v"(QuxT[2]), "n"(C0 + 0), "n"(C0 + 1));
    }     else if constexpr (CW == 2 && M == 4) {
      asm("s_nop 1\n\t"
          "v_fmac_f64_dpp %0, %2, %6 row_newbcast:%14 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %2, %6 row_newbcast:%15 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %6, %10 row_newbcast:%14 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %6, %10 row_newbcast:%15 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %3, %7 row_newbcast:%14 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %3, %7 row_newbcast:%15 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %7, %11 row_newbcast:%14 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %7, %11 row_newbcast:%15 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %4, %8 row_newbcast:%14 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %4, %8 row_newbcast:%15 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %8, %12 row_newbcast:%14 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %8, %12 row_newbcast:%15 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %5, %9 row_newbcast:%14 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %5, %9 row_newbcast:%15 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %9, %13 row_newbcast:%14 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %9, %13 row_newbcast:%15 row_mask:0xf bank_mask:0xf"
          : "+v"(Wn[C0 + 0]), "+v"(Wn[C0 + 1]) : "v"(G[0]), "v"(G[1]), "v"(G[2]), "v"(G[3]), "v"(KT[0]), "v"(KT[1]), "v"(KT[2]), "v"(KT[3]), "v"(QuxT[0]), "v"(QuxT[1]), "v"(QuxT[2]), "v"(QuxT[3]), "n"(C0 + 0), "n"(C0 + 1));
    }     else if constexpr (CW == 1 && M == 1) {
      asm("s_nop 1\n\t"
          "v_fmac_f64_dpp %0, %1, %2 row_newbcast:%4 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %2, %3 row_newbcast:%4 row_mask:0xf bank_mask:0xf"
          : "+v"(Wn[C0 + 0]) : "v"(G[0]), "v"(KT[0]), "v"(QuxT[0]), "n"(C0 + 0));
    }     else if constexpr (CW == 1 && M == 2) {
      asm("s_nop 1\n\t"
          "v_fmac_f64_dpp %0, %1, %3 row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %3, %5 row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %2, %4 row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %4, %6 row_newbcast:%7 row_mask:0xf bank_mask:0xf"
          : "+v"(Wn[C0 + 0]) : "v"(G[0]), "v"(G[1]), "v"(KT[0]), "v"(KT[1]), "v"(QuxT[0]), "v"(QuxT[1]), "n"(C0 + 0));
    }     else if constexpr (CW == 1 && M == 3) {
      asm("s_nop 1\n\t"
          "v_fmac_f64_dpp %0, %1, %4 row_newbcast:%10 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %4, %7 row_newbcast:%10 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %2, %5 row_newbcast:%10 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %5, %8 row_newbcast:%10 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %3, %6 row_newbcast:%10 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %6, %9 row_newbcast:%10 row_mask:0xf bank_mask:0xf"
          : "+v"(Wn[C0 + 0]) : "v"(G[0]), "v"(G[1]), "v"(G[2]), "v"(KT[0]), "v"(KT[1]), "v"(KT[2]), "v"(QuxT[0]), "v"(QuxT[1]), "v"(QuxT[2]), "n"(C0 + 0));
    }     else if constexpr (CW == 1 && M == 4) {
      asm("s_nop 1\n\t"
          "v_fmac_f64_dpp %0, %1, %5 row_newbcast:%13 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %5, %9 row_newbcast:%13 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %2, %6 row_newbcast:%13 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %6, %10 row_newbcast:%13 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %3, %7 row_newbcast:%13 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %7, %11 row_newbcast:%13 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %4, %8 row_newbcast:%13 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %0, %8, %12 row_newbcast:%13 row_mask:0xf bank_mask:0xf"
          : "+v"(Wn[C0 + 0]) : "v"(G[0]), "v"(G[1]), "v"(G[2]), "v"(G[3]), "v"(KT[0]), "v"(KT[1]), "v"(KT[2]), "v"(KT[3]), "v"(QuxT[0]), "v"(QuxT[1]), "v"(QuxT[2]), "v"(QuxT[3]), "n"(C0 + 0));
    }
    od_row_vxx_terms<C0 + CW, N, M>(Wn, G, KT, QuxT);
  }
}
template <int A_, int M> __device__ __forceinline__ void od_row_replicate(const double* row, double (*R)[M]) {
  // R[a][j] = (row[j] of lane a): the m x m block that lanes 0 .. m-1 hold by rows, in every lane
  if constexpr (A_ < M) {
#pragma unroll
    for (int j = 0; j < M; ++j) R[A_][j] = od_row_bc<A_>(row[j]);
    od_row_replicate<A_ + 1, M>(row, R);
  }
}
template <int A_, int M> __device__ __forceinline__ void od_row_replicate1(double v, double* R) {
  if constexpr (A_ < M) { R[A_] = od_row_bc<A_>(v); od_row_replicate1<A_ + 1, M>(v, R); }
}

constexpr int OD_IL_ROW_PAD = 17;     // doubles per row of the transpose pad (16 + 1: the column reads hit 16 banks)
template <int N, int M, class TA = double>
__global__ __launch_bounds__(OD_IL_THREADS) void k_ilqr_backward_row(IlqrArgsT<TA> a) {
  if (a.skip && *a.skip) return;
  static_assert(N <= 16 && M <= 4 && M <= N, "one matrix row per lane of a 16-lane DPP row; the m x m block is replicated");
  // transposes go through LDS, each row of lanes in its own pad: rows of A -> columns of A, rows of B -> columns of B at the start
  // of a knot, the new Vxx at its end.  Writer and reader lanes belong to ONE wavefront, whose LDS operations execute in order:
  // a scheduling fence for the compiler, no workgroup barrier
  __shared__ double pad[OD_IL_THREADS / 16][16 * OD_IL_ROW_PAD];
  __shared__ double padb[OD_IL_THREADS / 16][16 * (M + 1)];
#define OD_IL_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
  const int tid = threadIdx.x, i = tid & 15, rw = tid >> 4;
  const long b = (long)blockIdx.x * (OD_IL_THREADS / 16) + rw;
  const bool live = b < a.B;
  const long bb = live ? b : a.B - 1;                 // (rows past the batch repeat its last trajectory, store nothing)
  const int ir = i < N ? i : N - 1, im = i < M ? i : M - 1;       // (lanes without a row repeat the last one)
  double Vr[N], vx, dV0, dV1;
  int okf;
  if (a.done_b && a.done_b[bb]) return;       // (the whole row leaves: its lanes share nothing with the other rows)
  double reg = a.reg_dev ? a.reg_dev[bb] : a.reg;
  // (one pass unless a.retry: a row whose Quu + reg I fails to factorise repeats its own recursion at a larger reg while the
  // other rows of the wavefront wait -- the rows share nothing but the instruction stream)
  for (;;) {
  dV0 = 0.0; dV1 = 0.0; okf = 1;
#pragma unroll
  for (int c = 0; c < N; ++c) Vr[c] = a.VxxT.at(ir + N * c, bb);
  vx = a.VxT.at(ir, bb);
  // the rows of A and B, which the first products of a knot wait for, are requested one knot ahead.  (Requesting every operand a
  // knot ahead was measured too: 64 trajectories 371 -> 325 us, but 4096 trajectories 439 -> 449 and 16 384 1.81 -> 1.93 ms -- the
  // 96 extra registers end the second wavefront per SIMD, and from 4096 trajectories on the kernel is bound by its memory traffic.)
  double Ar[N], Br[M], Ac[N], Bc[N], lxr[N], luxT[M], luur[M], lxv, luv;
  auto request_a = [&](long kq) {      // what the first products need
#pragma unroll
    for (int c = 0; c < N; ++c) Ar[c] = (double)a.A.at(ir + N * c, kq);
#pragma unroll
    for (int j = 0; j < M; ++j) Br[j] = (double)a.Bm.at(ir + N * j, kq);
  };
  auto request_b = [&](long kq) {      // the cost expansion
#pragma unroll
    for (int c = 0; c < N; ++c) lxr[c] = a.lxx.at(ir + N * c, kq);
#pragma unroll
    for (int j = 0; j < M; ++j) { luxT[j] = a.lux.at(j + M * ir, kq); luur[j] = a.luu.at(im + M * j, kq); }
    lxv = a.lx.at(ir, kq);
    luv = a.lu.at(im, kq);
  };
  request_a((long)(a.T - 1) * a.B + bb);
  for (int t = a.T - 1; t >= 0; --t) {
    const long kk = (long)t * a.B + bb;
    double W[N], WB[M], Qxx[N], QuxT[M], Quu[M], Qx, Qu;
    request_b(kk);
    // columns of A and B (rows of A', B') from the rows the lanes hold
#pragma unroll
    for (int c = 0; c < N; ++c) pad[rw][i * OD_IL_ROW_PAD + c] = Ar[c];
#pragma unroll
    for (int j = 0; j < M; ++j) padb[rw][i * (M + 1) + j] = Br[j];
    OD_IL_WAVE_SYNC();
#pragma unroll
    for (int l = 0; l < N; ++l) { Ac[l] = pad[rw][l * OD_IL_ROW_PAD + ir]; Bc[l] = padb[rw][l * (M + 1) + im]; }
    OD_IL_WAVE_SYNC();
    // W = Vxx A, WB = Vxx B  (row i)
#pragma unroll
    for (int c = 0; c < N; ++c) W[c] = 0.0;
#pragma unroll
    for (int j = 0; j < M; ++j) WB[j] = 0.0;
    od_row_mac<N, N>(W, Ar, Vr);
    od_row_mac<M, N>(WB, Br, Vr);
    if (t > 0) request_a(kk - a.B);
#pragma unroll
    for (int c = 0; c < N; ++c) Qxx[c] = lxr[c];
#pragma unroll
    for (int j = 0; j < M; ++j) { QuxT[j] = luxT[j]; Quu[j] = luur[j]; }
    Qx = lxv;
    Qu = luv;
    // Qxx = lxx + A'W, Qux' = lux' + A'(WB) (row i);  Quu = luu + B'(WB), Qu = lu + B'Vx (row i < m);  Qx = lx + A'Vx
    od_row_mac<N, N>(Qxx, W, Ac);
    od_row_mac<M, N>(QuxT, WB, Ac);
    od_row_mac<M, N>(Quu, WB, Bc);
    od_row_mac<1, N>(&Qx, &vx, Ac);
    od_row_mac<1, N>(&Qu, &vx, Bc);
    // the m x m block in every lane: Quu, Qu; Cholesky of Quu + reg I
    double QuuR[M][M], QuR[M], L[M][M];
    od_row_replicate<0, M>(Quu, QuuR);
    od_row_replicate1<0, M>(Qu, QuR);
#pragma unroll
    for (int c = 0; c < M; ++c) {
      double d = QuuR[c][c] + reg;
#pragma unroll
      for (int l = 0; l < c; ++l) d -= L[c][l] * L[c][l];
      if (!(d > 0.0)) { okf = 0; d = 1e-12; }
      const double id = od::od_rsqrt(d);      // (seed + Newton steps, < 1 ulp: od_math.h; the diagonal holds 1 / L[c][c])
      L[c][c] = id;
#pragma unroll
      for (int r = c + 1; r < M; ++r) {
        double sx = QuuR[r][c];
#pragma unroll
        for (int l = 0; l < c; ++l) sx -= L[r][l] * L[c][l];
        L[r][c] = sx * id;
      }
    }
    // K' row i = -(Quu + reg)^{-1} (column i of Qux);  k = -(Quu + reg)^{-1} Qu (replicated)
    double KT[M], kR[M];
    auto solve = [&](const double* rhs, double* y) {
#pragma unroll
      for (int r = 0; r < M; ++r) { double sx = rhs[r]; for (int l = 0; l < r; ++l) sx -= L[r][l] * y[l]; y[r] = sx * L[r][r]; }
#pragma unroll
      for (int r = M - 1; r >= 0; --r) { double sx = y[r]; for (int l = r + 1; l < M; ++l) sx -= L[l][r] * y[l]; y[r] = sx * L[r][r]; }
#pragma unroll
      for (int r = 0; r < M; ++r) y[r] = -y[r];
    };
    solve(QuxT, KT);
    solve(QuR, kR);
    if (a.retry && !okf) break;               // (the whole row agrees: the m x m block is replicated) this pass is thrown away
    if (live && i < N) {
#pragma unroll
      for (int j = 0; j < M; ++j) a.K.at(j + M * i, kk) = (TA)KT[j];
    }
    if (live && i < M) {
      double kv = kR[0];
#pragma unroll
      for (int r = 1; r < M; ++r) kv = (i == r) ? kR[r] : kv;
      a.k.at(i, kk) = (TA)kv;
    }
    // Quu k, the expected decrease, Quu K + Qux (column i)
    double Quuk[M], G[M];
    double d1 = 0.0, d2 = 0.0;
#pragma unroll
    for (int r = 0; r < M; ++r) { double sx = 0.0; for (int l = 0; l < M; ++l) sx += QuuR[r][l] * kR[l]; Quuk[r] = sx; }
#pragma unroll
    for (int r = 0; r < M; ++r) { d1 += kR[r] * QuR[r]; d2 += 0.5 * kR[r] * Quuk[r]; }
    dV0 += d1; dV1 += d2;
#pragma unroll
    for (int r = 0; r < M; ++r) { double sx = 0.0; for (int l = 0; l < M; ++l) sx += QuuR[r][l] * KT[l]; G[r] = sx + QuxT[r]; }
    // value function: Vx, Vxx (row i), symmetrised through the transpose pad
    double vxn = Qx;
#pragma unroll
    for (int l = 0; l < M; ++l) vxn += KT[l] * (Quuk[l] + QuR[l]) + QuxT[l] * kR[l];
    vx = vxn;
    od_row_vxx_terms<0, N, M>(Qxx, G, KT, QuxT);
#pragma unroll
    for (int c = 0; c < N; ++c) pad[rw][i * OD_IL_ROW_PAD + c] = Qxx[c];
    OD_IL_WAVE_SYNC();
#pragma unroll
    for (int c = 0; c < N; ++c) Vr[c] = (c == ir) ? Qxx[c] : 0.5 * (Qxx[c] + pad[rw][c * OD_IL_ROW_PAD + ir]);
    OD_IL_WAVE_SYNC();
  }
  if (!a.retry || okf || !(reg < 1e6)) break;
  reg = od_il_next_reg(reg);
  }
#undef OD_IL_WAVE_SYNC
  if (a.retry && !okf) {                      // not factorisable at any regularisation (non-finite data): no step for this trajectory
    dV0 = 0.0; dV1 = 0.0;
    for (int t = 0; t < a.T; ++t) {
      const long kk = (long)t * a.B + bb;
      if (live && i < N) {
#pragma unroll
        for (int j = 0; j < M; ++j) a.K.at(j + M * i, kk) = TA(0);
      }
      if (live && i < M) a.k.at(i, kk) = TA(0);
    }
  }
  if (live && i == 0) {
    a.dV.at(0, b) = dV0;
    a.dV.at(1, b) = dV1;
    if (a.status.ok()) a.status.at(0, b) = okf;
  }
}

// doubles of LDS per trajectory of a workgroup
static inline size_t od_il_lds_per_traj(int n, int m) { return (size_t)4 * n * n + 5 * n * m + 2 * m * m + 2 * n + 3 * m + 2 + 1; }

#include "od_ilqr_mfma.inc"

#else
// host test build (threads run one after the other, no workgroup cooperation): one lane per trajectory
template <class TA> __global__ __launch_bounds__(OD_BLOCK) void k_ilqr_backward_serial(IlqrArgsT<TA> a) {
  if (a.skip && *a.skip) return;
  const long b = (long)blockIdx.x * OD_BLOCK + threadIdx.x;
  if (b >= a.B) return;
  const int n = a.n, m = a.m;
  double Vxx[OD_IL_N * OD_IL_N], Vx[OD_IL_N], At[OD_IL_N * OD_IL_N], Bt[OD_IL_N * OD_IL_M], W[OD_IL_N * OD_IL_N];
  double Qxx[OD_IL_N * OD_IL_N], Quu[OD_IL_M * OD_IL_M], Qux[OD_IL_M * OD_IL_N], Qx[OD_IL_N], Qu[OD_IL_M];
  double L[OD_IL_M * OD_IL_M], Kt[OD_IL_M * OD_IL_N], kt[OD_IL_M];
  if (a.done_b && a.done_b[b]) return;
  double dV1, dV2, reg = a.reg_dev ? a.reg_dev[b] : a.reg;
  bool ok;
  for (;;) {
  for (int i = 0; i < n * n; ++i) Vxx[i] = a.VxxT.at(i, b);
  for (int i = 0; i < n; ++i) Vx[i] = a.VxT.at(i, b);
  dV1 = 0.0; dV2 = 0.0;
  ok = true;
  for (int t = a.T - 1; t >= 0; --t) {
    const long kk = (long)t * a.B + b;
    for (int i = 0; i < n * n; ++i) At[i] = (double)a.A.at(i, kk);
    for (int i = 0; i < n * m; ++i) Bt[i] = (double)a.Bm.at(i, kk);
    // W = Vxx * A  (n x n)
    for (int j = 0; j < n; ++j) for (int i = 0; i < n; ++i) { double s = 0; for (int l = 0; l < n; ++l) s += Vxx[i + n * l] * At[l + n * j]; W[i + n * j] = s; }
    for (int j = 0; j < n; ++j) for (int i = 0; i < n; ++i) { double s = a.lxx.at(i + n * j, kk); for (int l = 0; l < n; ++l) s += At[l + n * i] * W[l + n * j]; Qxx[i + n * j] = s; }
    for (int j = 0; j < n; ++j) for (int i = 0; i < m; ++i) { double s = a.lux.at(i + m * j, kk); for (int l = 0; l < n; ++l) s += Bt[l + n * i] * W[l + n * j]; Qux[i + m * j] = s; }
    // W(:, 0:m) = Vxx * B
    for (int j = 0; j < m; ++j) for (int i = 0; i < n; ++i) { double s = 0; for (int l = 0; l < n; ++l) s += Vxx[i + n * l] * Bt[l + n * j]; W[i + n * j] = s; }
    for (int j = 0; j < m; ++j) for (int i = 0; i < m; ++i) { double s = a.luu.at(i + m * j, kk); for (int l = 0; l < n; ++l) s += Bt[l + n * i] * W[l + n * j]; Quu[i + m * j] = s; }
    for (int i = 0; i < n; ++i) { double s = a.lx.at(i, kk); for (int l = 0; l < n; ++l) s += At[l + n * i] * Vx[l]; Qx[i] = s; }
    for (int i = 0; i < m; ++i) { double s = a.lu.at(i, kk); for (int l = 0; l < n; ++l) s += Bt[l + n * i] * Vx[l]; Qu[i] = s; }
    // Cholesky of Quu + reg I
    for (int i = 0; i < m * m; ++i) L[i] = Quu[i];
    for (int i = 0; i < m; ++i) L[i + m * i] += reg;
    for (int j = 0; j < m; ++j) {
      double d = L[j + m * j];
      for (int l = 0; l < j; ++l) d -= L[j + m * l] * L[j + m * l];
      if (!(d > 0.0)) { ok = false; d = 1e-12; }
      d = sqrt(d);
      L[j + m * j] = d;
      for (int i = j + 1; i < m; ++i) { double s = L[i + m * j]; for (int l = 0; l < j; ++l) s -= L[i + m * l] * L[j + m * l]; L[i + m * j] = s / d; }
    }
    if (a.retry && !ok) break;
    // K = -(Quu+reg)^{-1} Qux (column by column), k = -(Quu+reg)^{-1} Qu
    for (int c = 0; c <= n; ++c) {
      double y[OD_IL_M];
      for (int i = 0; i < m; ++i) y[i] = (c < n) ? Qux[i + m * c] : Qu[i];
      for (int i = 0; i < m; ++i) { double s = y[i]; for (int l = 0; l < i; ++l) s -= L[i + m * l] * y[l]; y[i] = s / L[i + m * i]; }
      for (int i = m - 1; i >= 0; --i) { double s = y[i]; for (int l = i + 1; l < m; ++l) s -= L[l + m * i] * y[l]; y[i] = s / L[i + m * i]; }
      for (int i = 0; i < m; ++i) { if (c < n) Kt[i + m * c] = -y[i]; else kt[i] = -y[i]; }
    }
    for (int i = 0; i < m * n; ++i) a.K.at(i, kk) = (TA)Kt[i];
    for (int i = 0; i < m; ++i) a.k.at(i, kk) = (TA)kt[i];
    for (int i = 0; i < m; ++i) { dV1 += kt[i] * Qu[i]; double s = 0; for (int l = 0; l < m; ++l) s += Quu[i + m * l] * kt[l]; dV2 += 0.5 * kt[i] * s; }
    // value function update (Quu without reg, as in the cost-to-go expansion)
    double Quuk[OD_IL_M];
    for (int i = 0; i < m; ++i) { double s = 0; for (int l = 0; l < m; ++l) s += Quu[i + m * l] * kt[l]; Quuk[i] = s; }
    for (int i = 0; i < n; ++i) {
      double s = Qx[i];
      for (int l = 0; l < m; ++l) s += Kt[l + m * i] * (Quuk[l] + Qu[l]) + Qux[l + m * i] * kt[l];
      Vx[i] = s;
    }
    // W(m x n) = Quu * K
    for (int j = 0; j < n; ++j) for (int i = 0; i < m; ++i) { double s = 0; for (int l = 0; l < m; ++l) s += Quu[i + m * l] * Kt[l + m * j]; W[i + m * j] = s; }
    for (int j = 0; j < n; ++j) for (int i = 0; i < n; ++i) {
      double s = Qxx[i + n * j];
      for (int l = 0; l < m; ++l) s += Kt[l + m * i] * (W[l + m * j] + Qux[l + m * j]) + Qux[l + m * i] * Kt[l + m * j];
      Vxx[i + n * j] = s;
    }
    for (int j = 0; j < n; ++j) for (int i = 0; i < j; ++i) { const double s = 0.5 * (Vxx[i + n * j] + Vxx[j + n * i]); Vxx[i + n * j] = s; Vxx[j + n * i] = s; }
  }
  if (!a.retry || ok || !(reg < 1e6)) break;
  reg = od_il_next_reg(reg);
  }
  if (a.retry && !ok) {
    dV1 = 0.0; dV2 = 0.0;
    for (int t = 0; t < a.T; ++t) {
      const long kk = (long)t * a.B + b;
      for (int i = 0; i < m * n; ++i) a.K.at(i, kk) = TA(0);
      for (int i = 0; i < m; ++i) a.k.at(i, kk) = TA(0);
    }
  }
  a.dV.at(0, b) = dV1;
  a.dV.at(1, b) = dV2;
  if (a.status.ok()) a.status.at(0, b) = ok ? 1 : 0;
}

#endif

}  // namespace

struct od_handle_s {
  const ModelVT* vt;
  int device;            // the HIP device that was current in od_create: every entry point runs there (OnDevice)
  int dtype, layout;
  od_options opts;
  double h, fric[4], u_max;
  int proj_stall_exit;   // od_set_projection_stall_exit
  int polish64;          // od_set_mixed_precision
  hipStream_t stream;
  int ppw;         // problems per wavefront; 0 = automatic (od_auto_ppw)
  int wpb;         // wavefronts per workgroup of the state pass: 0 automatic, 1 or 4
  int coop;        // cooperative state kernels (od_coop.h): 0 automatic, 1 never, 2 wherever the model has them
  double* work;    // device workspace: gradient iterates handed from pass 1 to pass 2
  size_t work_elems;
  long grad_knots; // knots of the last gradient pass (the hand-over is batch-minor with that stride); 0 = none
  double* stage;   // device staging for the host-pointer entry points (rocket, bundle)
  size_t stage_elems;
  double* hstage;  // pinned, device-mapped host staging of od_f_host / od_fx_host / od_fu_host
  double* hstage_dev;
  size_t hstage_elems;
  std::vector<od_ilqr_s*> solvers;   // live od_ilqr solvers made from this handle (od_destroy releases what they hold)
};
static void il_detach_all(od_handle_s* h, bool device_ok);      // od_ilqr_solver.inc

namespace {

template <class T> View<T> mkview(void* p, long E, long K, int layout) {
  View<T> v;
  v.p = (T*)p;
  if (layout == OD_LAYOUT_BATCH_MINOR) { v.se = K; v.sb = 1; }
  else { v.se = 1; v.sb = E; }
  return v;
}
template <class T> View<const T> mkcview(const void* p, long E, long K, int layout) {
  View<const T> v;
  v.p = (const T*)p;
  if (layout == OD_LAYOUT_BATCH_MINOR) { v.se = K; v.sb = 1; }
  else { v.se = 1; v.sb = E; }
  return v;
}

int ppw_of(const od_handle_s* h, long n) { return h->ppw > 0 ? h->ppw : od_auto_ppw(n); }

// pass-1 launch shape: 4 wavefronts per workgroup (one per SIMD) while that still spreads the batch over
// the chip, problems per wavefront from od_auto_ppw (aims at one resident wavefront per SIMD)
// The cooperative kernels put one problem on 16 lanes (4 per wavefront): they shorten the critical path of a problem
// and pay off while the batch leaves lanes idle -- up to OD_COOP_AUTO_MAX problems (measured: 8192 rollouts 5.4 ms against 5.7, 16384 10.4 against 6.6);
// larger batches fill the lanes with whole problems instead.
LaunchCfg cfg_of(const od_handle_s* h, long n) {
  LaunchCfg c;
  c.ppw = ppw_of(h, n);
  c.wpb = h->wpb > 0 ? h->wpb : 4;
  // cooperative kernels: od_set_cooperative 1 = never, 2 = wherever the model has them (16 lanes per problem first), 3 = the
  // 8-lane form first; automatic (0, and no explicit lane mapping): 16 lanes up to the model's coop_auto_max problems, 8 lanes
  // up to coop8_auto_max
  const ModelVT* vt = h->vt;
  c.coop = 0;
  if (h->coop == 2) c.coop = vt->has_coop ? 1 : (vt->has_coop8 ? 2 : 0);
  else if (h->coop == 3) c.coop = vt->has_coop8 ? 2 : (vt->has_coop ? 1 : 0);
  else if (h->coop == 0 && h->ppw == 0) {
    if (vt->has_coop == 2 && n <= vt->coop_auto_max) c.coop = 1;
    else if (vt->has_coop8 == 2 && n <= vt->coop8_auto_max) c.coop = 2;
  }
  return c;
}

int ensure_work(od_handle_s* h, size_t elems) {
  if (h->work_elems >= elems) return OD_OK;
  if (h->work) {
    if (hipStreamSynchronize(h->stream) != hipSuccess) return fail(OD_ERR_HIP, "hipStreamSynchronize failed");
    (void)hipFree(h->work);
    h->work = nullptr;
    h->work_elems = 0;
  }
  if (hipMalloc((void**)&h->work, elems * sizeof(double)) != hipSuccess) return fail(OD_ERR_HIP, "hipMalloc (gradient workspace) failed");
  h->work_elems = elems;
  return OD_OK;
}

// with cones and a finite undercut the centering floor depends on kappa_tol, so the two
// simulators' iterates differ: the eval and grad solves must then be run separately.
bool fusable(const od_handle_s* h) {
  return std::isinf(h->opts.undercut) || h->opts.kappa_eval_tol == h->opts.kappa_grad_tol ||
         h->vt->kind == 1 /* rocket dynamics: no cones */ ||
         (h->vt->id == OD_ACROBOT_NOMINAL || h->vt->id == OD_CARTPOLE_FRICTIONLESS);
}

StepArgs<double> step_args(od_handle_s* h, long B, long K, const void* x, const void* u, void* d, void* dx,
                           void* du, void* dq3, int* status, int* iters, int want_grad) {
  const ModelVT* vt = h->vt;
  const int nq = vt->nq, n = 2 * nq, nu = vt->nu, L = h->layout;
  StepArgs<double> a;
  a.B = B;
  a.h = h->h;
  for (int i = 0; i < 4; ++i) a.fric[i] = h->fric[i];
  a.opts = to_opts<double>(h->opts);
  a.x = mkcview<double>(x, n, B, L);
  a.u = mkcview<double>(u, nu, K, L);
  a.d = mkview<double>(d, n, K, L);
  a.dx = mkview<double>(dx, n * n, K, L);
  a.du = mkview<double>(du, n * nu, K, L);
  a.dq3 = mkview<double>(dq3, nq * (n + nu), K, L);
  a.q3.p = nullptr; a.q3.se = 0; a.q3.sb = 0;
  a.status = mkview<int>(status, 1, K, L);
  a.iters = mkview<int>(iters, 2, K, L);
  a.zg.p = nullptr; a.zg.se = 0; a.zg.sb = 0;
  a.want_grad = want_grad;
  a.merge_grad_status = 0;
  return a;
}

int check_mech(od_handle_s* h, const char* fn) {
  if (!h) return fail(OD_ERR_INVALID, std::string(fn) + ": null handle");
  if (h->vt->kind != 0) return fail(OD_ERR_UNSUPPORTED, std::string(fn) + ": model is not a mechanical (q1,q2,u) model");
  if (h->dtype != OD_F64) return fail(OD_ERR_UNSUPPORTED, std::string(fn) + ": mechanical models are instantiated for OD_F64 only");
  if (!(h->h > 0)) return fail(OD_ERR_INVALID, std::string(fn) + ": the handle has no positive time step");
  return OD_OK;
}

// pass 2 over K knots whose states live in `xstate` (slot k) -- shared by od_step_grad and od_rollout
int run_grad_pass(od_handle_s* h, StepArgs<double> g, long K, View<const double> xstate, LiveArgs lv = LiveArgs{nullptr, nullptr, 1}) {
  h->grad_knots = K;
  g.B = K;
  g.x = xstate;
  g.d.p = nullptr;
  g.iters.p = nullptr;
  OD_HIP(h->vt->grad_knots(g, h->stream, lv));
  return OD_OK;
}

int run_step(od_handle_s* h, const char* fn, long B, const void* x, const void* u, void* d, void* dx, void* du,
             void* dq3, int* status, int* iters, int want_grad, void* q3 = nullptr, long x_se = 0, LiveArgs lv = LiveArgs{nullptr, nullptr, 1}) {
  if (int rc = check_mech(h, fn)) return rc;
  if (B <= 0) return OD_OK;
  if (!x || (h->vt->nu > 0 && !u)) return fail(OD_ERR_INVALID, std::string(fn) + ": null input");
  OD_ON_DEVICE(h);
  StepArgs<double> a = step_args(h, B, B, x, u, d, dx, du, dq3, status, iters, want_grad);
  if (x_se) a.x.se = x_se;         // (the states are the first B slots of a longer batch-minor array: od_ilqr_solver.inc)
  a.q3 = mkview<double>(q3, h->vt->nq, B, h->layout);
  if (!want_grad) {
    OD_HIP(h->vt->step_state(a, cfg_of(h, B), h->stream, lv));
    return OD_OK;
  }
  if (int rc = ensure_work(h, (size_t)(h->vt->nz + 1) * (size_t)B)) return rc;
  a.zg = mkview<double>(h->work, h->vt->nz + 1, B, OD_LAYOUT_BATCH_MINOR);
  if (!fusable(h)) {
    // eval_sim and grad_sim iterate differently (finite undercut): run them separately like the reference
    StepArgs<double> e = a;
    e.want_grad = 0;
    OD_HIP(h->vt->step_state(e, cfg_of(h, B), h->stream, lv));
    StepArgs<double> g = a;
    g.d.p = nullptr;
    g.q3.p = nullptr;
    g.merge_grad_status = 1;
    g.opts.kappa_eval = g.opts.kappa_grad;
    OD_HIP(h->vt->step_state(g, cfg_of(h, B), h->stream, lv));
  } else {
    OD_HIP(h->vt->step_state(a, cfg_of(h, B), h->stream, lv));
  }
  return run_grad_pass(h, a, B, a.x, lv);
}

}  // namespace

// the thrust-cone projection's options as a double-precision handle runs it (single-precision handles under od_set_mixed_precision)
static Opts<double> proj_opts64() {
  od_options po;
  defaults_of(vt_rocket_projection(), &po);
  return to_opts<double>(po);
}

template <class T> static int rocket_impl(od_handle h, long B, int project, const void* x, const void* u, void* y,
                                          void* dx, void* du, void* uproj, int* status) {
  const int L = h->layout;
  RocketArgs<T> a;
  a.B = B;
  a.h = (T)h->h;
  a.u_max = (T)h->u_max;
  a.opts_dyn = to_opts<T>(h->opts);
  od_options po;
  defaults_of(vt_rocket_projection(), &po);
  if (h->dtype == OD_F32) {   // tolerances reachable in fp32
    po.r_tol = std::fmax(po.r_tol, (double)h->opts.r_tol);
  }
  a.opts_proj = to_opts<T>(po);
  a.opts_proj64 = proj_opts64();
  a.project = project;
  a.want_grad = (dx || du) ? 1 : 0;
  a.x = mkcview<T>(x, 12, B, L);
  a.u = mkcview<T>(u, 3, B, L);
  a.y = mkview<T>(y, 12, B, L);
  a.dx = mkview<T>(dx, 144, B, L);
  a.du = mkview<T>(du, 36, B, L);
  a.uproj = mkview<T>(uproj, 3, B, L);
  a.status = mkview<int>(status, 1, B, L);
  a.skip = nullptr; a.live = nullptr; a.live_mod = 1; a.proj_stall_exit = h->proj_stall_exit; a.polish64 = h->polish64; a.h64 = h->h;
  hipError_t e;
  if constexpr (sizeof(T) == 8) e = launch_rocket64(a, ppw_of(h, B), h->stream);
  else e = launch_rocket32(a, ppw_of(h, B), h->stream);
  if (e != hipSuccess) return fail(OD_ERR_HIP, std::string("od_rocket launch: ") + hipGetErrorString(e));
  return OD_OK;
}


template <class T> static RocketArgs<T> rocket_args(od_handle h, long B, int project, int want_grad) {
  RocketArgs<T> a;
  a.B = B;
  a.h = (T)h->h;
  a.u_max = (T)h->u_max;
  a.opts_dyn = to_opts<T>(h->opts);
  od_options po;
  defaults_of(vt_rocket_projection(), &po);
  if (h->dtype == OD_F32) po.r_tol = std::fmax(po.r_tol, (double)h->opts.r_tol);   // tolerances reachable in fp32
  a.opts_proj = to_opts<T>(po);
  a.opts_proj64 = proj_opts64();
  a.project = project;
  a.want_grad = want_grad;
  a.x.p = nullptr; a.u.p = nullptr; a.y.p = nullptr; a.dx.p = nullptr; a.du.p = nullptr; a.uproj.p = nullptr; a.status.p = nullptr;
  a.skip = nullptr; a.live = nullptr; a.live_mod = 1; a.proj_stall_exit = h->proj_stall_exit; a.polish64 = h->polish64; a.h64 = h->h;
  return a;
}

template <class T> static int soc_project_impl(od_handle h, long B, const void* u, void* uproj, void* duproj, void* zfull, int* status, int* iters) {
  const int L = h->layout;
  SocProjectArgs<T> sa;
  RocketArgs<T>& a = sa.a;
  a = rocket_args<T>(h, B, 1, duproj ? 1 : 0);
  a.u = mkcview<T>(u, 3, B, L);
  a.uproj = mkview<T>(uproj, 3, B, L);
  a.du = mkview<T>(duproj, 9, B, L);
  a.status = mkview<int>(status, 1, B, L);
  sa.z = mkview<T>(zfull, 10, B, L);
  sa.iters = mkview<int>(iters, 1, B, L);
  hipError_t e;
  if constexpr (sizeof(T) == 8) e = launch_soc_project64(sa, ppw_of(h, B), h->stream);
  else e = launch_soc_project32(sa, ppw_of(h, B), h->stream);
  if (e != hipSuccess) return fail(OD_ERR_HIP, std::string("od_soc_project launch: ") + hipGetErrorString(e));
  return OD_OK;
}

template <class T> static int rocket_rollout_impl(od_handle h, long B, int Tn, int nalpha, const void* alphas, int project,
                                                  const void* x1, const void* xbar, const void* ubar, const void* K,
                                                  const void* kff, void* X, void* U, int* status) {
  const int L = h->layout;
  const long P = nalpha > 0 ? B * nalpha : B, Kn = (long)Tn * B, Kc = (long)Tn * P;
  RocketRolloutArgs<T> ra;
  ra.a = rocket_args<T>(h, P, project, 0);
  ra.a.x = mkcview<T>(x1, 12, B, L);
  View<T> xv = mkview<T>(X, 12, (long)(Tn + 1) * P, L);
  ra.x0 = xv;
  ra.a.y = xv;
  ra.a.y.p += (long)P * xv.sb;
  ra.a.status = mkview<int>(status, 1, Kc, L);
  ra.Tn = Tn;
  ra.Bnom = B;
  ra.nalpha = nalpha;
  ra.alphas = (const T*)alphas;
  ra.xbar = mkcview<T>(xbar, 12, (long)(Tn + 1) * B, L);
  ra.ubar = mkcview<T>(ubar, 3, Kn, L);
  ra.K = mkcview<T>(K, 36, Kn, L);
  ra.kff = mkcview<T>(kff, 3, Kn, L);
  ra.U = mkview<T>(U, 3, Kc, L);
  ra.cq = ra.cr = ra.cqt = ra.cxref = nullptr; ra.J = nullptr; ra.okall = nullptr;
  hipError_t e;
  if constexpr (sizeof(T) == 8) e = launch_rocket_rollout64(ra, ppw_of(h, P), h->stream);
  else e = launch_rocket_rollout32(ra, ppw_of(h, P), h->stream);
  if (e != hipSuccess) return fail(OD_ERR_HIP, std::string("od_rocket_rollout launch: ") + hipGetErrorString(e));
  return OD_OK;
}

extern "C" {

int od_version(void) { return OD_ABI_VERSION; }
const char* od_last_error(void) { return g_err.c_str(); }

int od_model_dims(int model, int* nq, int* nu, int* nz, int* ntheta, int* nfric) {
  const ModelVT* vt = vt_of(model);
  if (!vt) return fail(OD_ERR_INVALID, "od_model_dims: unknown model");
  if (nq) *nq = vt->nq;
  if (nu) *nu = vt->nu;
  if (nz) *nz = vt->nz;
  if (ntheta) *ntheta = vt->nth;
  if (nfric) *nfric = vt->nfric;
  return OD_OK;
}

int od_default_friction(int model, double* mu, int n) {
  const ModelVT* vt = vt_of(model);
  if (!vt || !mu) return fail(OD_ERR_INVALID, "od_default_friction: bad arguments");
  for (int i = 0; i < n && i < 4; ++i) mu[i] = vt->fric_default[i];
  return OD_OK;
}

int od_num_models(void) { return OD_MODEL_COUNT; }

int od_model_id(const char* name) {
  if (!name) return -1;
  for (int i = 0; i < OD_MODEL_COUNT; ++i) {
    const ModelVT* vt = vt_of(i);
    if (vt && std::strcmp(vt->name, name) == 0) return i;
  }
  return -1;
}

const char* od_model_name(int model) {
  const ModelVT* vt = vt_of(model);
  return vt ? vt->name : nullptr;
}

int od_default_options(int model, od_options* out) {
  const ModelVT* vt = vt_of(model);
  if (!vt || !out) return fail(OD_ERR_INVALID, "od_default_options: bad arguments");
  defaults_of(vt, out);
  return OD_OK;
}

int od_raw_grad_dims(int model, int* nzq, int* ngc) {
  const ModelVT* vt = vt_of(model);
  if (!vt) return fail(OD_ERR_INVALID, "od_raw_grad_dims: unknown model");
  if (nzq) *nzq = vt->nzq;
  if (ngc) *ngc = vt->ngc;
  return OD_OK;
}

// InteriorPointOptions the solver can run with: a line search needs one trial (it keeps the last one whatever happens),
// tolerances are positive numbers (NaN fails every comparison here), undercut >= 1 or INFINITY
static const char* bad_options(const od_options* o) {
  if (!(o->r_tol > 0) || !(o->kappa_eval_tol > 0) || !(o->kappa_grad_tol > 0)) return "r_tol, kappa_eval_tol and kappa_grad_tol must be positive";
  if (o->max_iter < 0) return "max_iter must be >= 0";
  if (o->max_ls < 1) return "max_ls must be >= 1";
  if (!(o->eps_min >= 0) || !(o->eps_min <= 1) || !(o->kappa_reg >= 0) || !(o->gamma_reg >= 0)) return "eps_min in [0, 1], kappa_reg >= 0, gamma_reg >= 0";
  if (!(o->undercut > 0)) return "undercut must be positive (INFINITY allowed)";
  return nullptr;
}

int od_create(int model, int dtype, const od_options* opts, double dt, od_handle* out) {
  const ModelVT* vt = vt_of(model);
  if (!vt || !out) return fail(OD_ERR_INVALID, "od_create: bad arguments");
  if (!(dt >= 0)) return fail(OD_ERR_INVALID, "od_create: the time step must not be negative (0: a handle for od_ip_solve only)");
  if (opts) { if (const char* why = bad_options(opts)) return fail(OD_ERR_INVALID, std::string("od_create: ") + why); }
  if (dtype != OD_F64 && dtype != OD_F32) return fail(OD_ERR_INVALID, "od_create: bad dtype");
  if (dtype == OD_F32 && !vt->raw32)
    return fail(OD_ERR_UNSUPPORTED, "od_create: OD_F32 is instantiated for the rocket models only");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(OD_ERR_NO_DEVICE, "od_create: no HIP device visible (this library has no CPU path)");
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return fail(OD_ERR_HIP, "od_create: hipGetDevice failed");
  od_handle_s* h = new od_handle_s();
  h->vt = vt;
  h->device = dev;
  h->dtype = dtype;
  h->layout = OD_LAYOUT_BATCH_MINOR;
  if (opts) h->opts = *opts; else defaults_of(vt, &h->opts);
  h->h = dt;
  for (int i = 0; i < 4; ++i) h->fric[i] = vt->fric_default[i];
  h->u_max = 12.5;   // examples/rocket.jl:16
  h->proj_stall_exit = 0;   // (the reference runs every iteration; od_ilqr_options.proj_stall_exit switches it on for the solver's own launches)
  h->polish64 = 1;
  h->stream = nullptr;
  h->ppw = 0;
  h->coop = 0;
  h->wpb = 0;
  h->work = nullptr;
  h->work_elems = 0;
  h->grad_knots = 0;
  h->stage = nullptr;
  h->stage_elems = 0;
  h->hstage = nullptr;
  h->hstage_dev = nullptr;
  h->hstage_elems = 0;
  *out = h;
  return OD_OK;
}

int od_destroy(od_handle h) {
  if (!h) return OD_OK;
  // best effort (called from finalisers that ignore the return code): if the handle's device cannot be made current no HIP call is made on
  // another one, but the host objects go and the live solvers are detached all the same
  int rc = OD_OK;
  OnDevice g(h->device);
  const bool dev_ok = g.err == hipSuccess;
  if (!dev_ok) rc = fail(OD_ERR_HIP, std::string("od_destroy: switching to the handle's device: ") + hipGetErrorString(g.err) + " (device memory not released)");
  il_detach_all(h, dev_ok);
  if (dev_ok) {
    if (h->stage) (void)hipFree(h->stage);
    if (h->hstage) (void)hipHostFree(h->hstage);
    if (h->work) (void)hipFree(h->work);
  }
  delete h;
  return rc;
}

int od_get_device(od_handle h, int* device) {
  if (!h || !device) return fail(OD_ERR_INVALID, "od_get_device: null argument");
  *device = h->device;
  return OD_OK;
}

int od_set_options(od_handle h, const od_options* o) {
  if (!h || !o) return fail(OD_ERR_INVALID, "od_set_options: bad arguments");
  if (const char* why = bad_options(o)) return fail(OD_ERR_INVALID, std::string("od_set_options: ") + why);
  h->opts = *o;
  return OD_OK;
}
int od_get_options(od_handle h, od_options* o) {
  if (!h || !o) return fail(OD_ERR_INVALID, "od_get_options: bad arguments");
  *o = h->opts;
  return OD_OK;
}
int od_set_timestep(od_handle h, double dt) {
  if (!h || !(dt > 0)) return fail(OD_ERR_INVALID, "od_set_timestep: bad arguments");
  h->h = dt;
  return OD_OK;
}
int od_set_friction(od_handle h, const double* mu, int n) {
  if (!h || n != h->vt->nfric || (n > 0 && !mu)) return fail(OD_ERR_INVALID, "od_set_friction: model has a different number of friction coefficients");
  for (int i = 0; i < n; ++i) h->fric[i] = mu[i];
  return OD_OK;
}
int od_set_u_max(od_handle h, double u_max) {
  if (!h) return fail(OD_ERR_INVALID, "od_set_u_max: null handle");
  h->u_max = u_max;
  return OD_OK;
}
int od_set_projection_stall_exit(od_handle h, int on) {
  if (!h) return fail(OD_ERR_INVALID, "od_set_projection_stall_exit: null handle");
  h->proj_stall_exit = on ? 1 : 0;
  return OD_OK;
}
int od_set_mixed_precision(od_handle h, int on) {
  if (!h) return fail(OD_ERR_INVALID, "od_set_mixed_precision: null handle");
  h->polish64 = on ? 1 : 0;
  return OD_OK;
}
int od_set_layout(od_handle h, int layout) {
  if (!h || (layout != OD_LAYOUT_BATCH_MINOR && layout != OD_LAYOUT_BATCH_MAJOR)) return fail(OD_ERR_INVALID, "od_set_layout: bad arguments");
  h->layout = layout;
  return OD_OK;
}
int od_set_stream(od_handle h, void* s) {
  if (!h) return fail(OD_ERR_INVALID, "od_set_stream: null handle");
  OD_ON_DEVICE(h);
  if (s && h->stream != (hipStream_t)s) {
    int sdev = h->device;
    if (hipStreamGetDevice((hipStream_t)s, &sdev) == hipSuccess && sdev != h->device)
      return fail(OD_ERR_WRONG_DEVICE, "od_set_stream: the stream belongs to device " + std::to_string(sdev) + ", the handle to device " + std::to_string(h->device));
    (void)hipGetLastError();
  }
  if (h->stream != (hipStream_t)s) {
    // the handle's workspaces (gradient hand-over, staging) are shared by consecutive calls: a handle works on one
    // stream at a time, so work queued on the old stream finishes before the new one may reuse them.  If the old stream
    // cannot be synchronised (the caller destroyed it), the device is synchronised instead and the new stream is adopted
    // all the same -- the handle must not stay tied to a dead stream; the failure is still reported.
    const hipError_t e = hipStreamSynchronize(h->stream);
    h->stream = (hipStream_t)s;
    if (e != hipSuccess) {
      (void)hipGetLastError();
      (void)hipDeviceSynchronize();
      return fail(OD_ERR_HIP, std::string("od_set_stream: the previous stream could not be synchronised (") + hipGetErrorString(e) + "); the new stream is in use");
    }
  }
  return OD_OK;
}
int od_set_cooperative(od_handle h, int mode) {
  if (!h || mode < 0 || mode > 3) return fail(OD_ERR_INVALID, "od_set_cooperative: mode 0 (automatic), 1 (never), 2 (always; 16 lanes per problem where the model has both forms) or 3 (always; 8 lanes per problem first)");
  h->coop = mode;
  return OD_OK;
}

int od_get_grad_iterates(od_handle h, long K, void* out) {
  if (!h || !out || K <= 0) return fail(OD_ERR_INVALID, "od_get_grad_iterates: null handle / buffer");
  OD_ON_DEVICE(h);
  const size_t n = (size_t)(h->vt->nz + 1) * (size_t)K;
  if (!h->work || h->work_elems < n || h->grad_knots != K)
    return fail(OD_ERR_INVALID, "od_get_grad_iterates: K must be the number of knots of the last gradient pass on this handle (" + std::to_string(h->grad_knots) + ")");
  OD_HIP(hipMemcpyAsync(out, h->work, n * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  return OD_OK;
}

int od_uses_cooperative(od_handle h, long B) { return (h && h->vt && cfg_of(h, B).coop) ? 1 : 0; }

int od_set_launch_config(od_handle h, int ppw, int waves_per_block) {
  if (!h || ppw < 0 || ppw > 64 || (ppw & (ppw - 1)) || !(waves_per_block == 0 || waves_per_block == 1 || waves_per_block == 4))
    return fail(OD_ERR_INVALID, "od_set_launch_config: ppw = 0 (auto) or a power of two <= 64; waves_per_block in {0, 1, 4}");
  h->ppw = ppw;
  h->wpb = waves_per_block;
  return OD_OK;
}
int od_synchronize(od_handle h) {
  if (!h) return fail(OD_ERR_INVALID, "od_synchronize: null handle");
  OD_ON_DEVICE(h);
  OD_HIP(hipStreamSynchronize(h->stream));
  return OD_OK;
}

int od_step(od_handle h, long B, const void* x, const void* u, void* d, int* status, int* iters) {
  return run_step(h, "od_step", B, x, u, d, nullptr, nullptr, nullptr, status, iters, 0);
}

int od_step_grad(od_handle h, long B, const void* x, const void* u, void* d, void* dx, void* du, int* status, int* iters) {
  return run_step(h, "od_step_grad", B, x, u, d, dx, du, nullptr, status, iters, 1);
}

int od_step_grad_compact(od_handle h, long B, const void* x, const void* u, void* q3, void* dq3, int* status, int* iters) {
  if (int rc = check_mech(h, "od_step_grad_compact")) return rc;
  if (B <= 0) return OD_OK;
  return run_step(h, "od_step_grad_compact", B, x, u, nullptr, nullptr, nullptr, dq3, status, iters, dq3 ? 1 : 0, q3);
}

static int rollout_impl(od_handle h, const char* fn, long B, int T, const void* x1, const void* U, void* X, void* A, void* Bm, void* dq3,
                        int* status, int* iters) {
  if (int rc = check_mech(h, fn)) return rc;
  if (B <= 0 || T <= 0) return OD_OK;
  if (!x1 || !U || !X) return fail(OD_ERR_INVALID, std::string(fn) + ": null x1/U/X");
  OD_ON_DEVICE(h);
  const int want_grad = (A || Bm || dq3) ? 1 : 0;
  const bool fused = fusable(h);
  const int n = 2 * h->vt->nq, nz = h->vt->nz;
  const long K = (long)T * B;
  RolloutArgs<double> r;
  r.s = step_args(h, B, K, x1, U, nullptr, A, Bm, dq3, status, iters, (want_grad && fused) ? 1 : 0);
  // X has (T+1)*B slots; knot k's d = [q2; q3] goes to slot k + B
  View<double> xv = mkview<double>(X, n, (long)(T + 1) * B, h->layout);
  r.x0 = xv;
  r.s.d = xv;
  r.s.d.p += (long)B * xv.sb;
  r.Tn = T;
  if (want_grad) {
    if (int rc = ensure_work(h, (size_t)(nz + 1) * (size_t)K)) return rc;
    r.s.zg = mkview<double>(h->work, nz + 1, K, OD_LAYOUT_BATCH_MINOR);
  }
  OD_HIP(h->vt->rollout_state(r, cfg_of(h, B), h->stream));          // pass 1: time recursion
  if (!want_grad) return OD_OK;
  View<const double> xin;
  xin.p = xv.p; xin.se = xv.se; xin.sb = xv.sb;                        // state of knot k = slot k of X
  if (!fused) {
    // finite undercut with kappa_eval != kappa_grad: the reference's grad simulator iterates differently from its eval
    // simulator, so every knot is solved again at kappa_grad from its rolled-out state (all K knots in parallel), like
    // fx / fu called on the states of iLQR.rollout; status / iterations of that solve are merged in
    StepArgs<double> g = r.s;
    g.B = K;
    g.x = xin;
    g.d.p = nullptr;
    g.want_grad = 1;
    g.merge_grad_status = 1;
    g.opts.kappa_eval = g.opts.kappa_grad;
    OD_HIP(h->vt->step_state(g, cfg_of(h, K), h->stream, LiveArgs{nullptr, nullptr, 1}));
  }
  return run_grad_pass(h, r.s, K, xin);                                // pass 2: all T*B gradients
}

int od_rollout(od_handle h, long B, int T, const void* x1, const void* U, void* X, void* A, void* Bm, int* status, int* iters) {
  return rollout_impl(h, "od_rollout", B, T, x1, U, X, A, Bm, nullptr, status, iters);
}

int od_rollout_compact(od_handle h, long B, int T, const void* x1, const void* U, void* X, void* dq3, int* status, int* iters) {
  return rollout_impl(h, "od_rollout_compact", B, T, x1, U, X, nullptr, nullptr, dq3, status, iters);
}

int od_rollout_policy(od_handle h, long B, int T, int nalpha, const void* alphas, const void* x1, const void* xbar,
                      const void* ubar, const void* K, const void* kff, void* X, void* U, int* status, int* iters) {
  if (int rc = check_mech(h, "od_rollout_policy")) return rc;
  if (B <= 0 || T <= 0 || nalpha <= 0) return OD_OK;
  if (!alphas || !x1 || !xbar || !ubar || !K || !kff || !X || !U) return fail(OD_ERR_INVALID, "od_rollout_policy: null argument");
  OD_ON_DEVICE(h);
  const int n = 2 * h->vt->nq, nu = h->vt->nu, L = h->layout;
  const long P = B * nalpha, Kn = (long)T * B, Kc = (long)T * P;
  PolicyArgs<double> pa;
  pa.r.s = step_args(h, P, Kc, x1, nullptr, nullptr, nullptr, nullptr, nullptr, status, iters, 0);
  pa.r.s.x = mkcview<double>(x1, n, B, L);
  View<double> xv = mkview<double>(X, n, (long)(T + 1) * P, L);
  pa.r.x0 = xv;
  pa.r.s.d = xv;
  pa.r.s.d.p += (long)P * xv.sb;
  pa.r.Tn = T;
  pa.Bnom = B;
  pa.nalpha = nalpha;
  pa.alphas = (const double*)alphas;
  pa.xbar = mkcview<double>(xbar, n, (long)(T + 1) * B, L);
  pa.ubar = mkcview<double>(ubar, nu, Kn, L);
  pa.K = mkcview<double>(K, nu * n, Kn, L);
  pa.kff = mkcview<double>(kff, nu, Kn, L);
  pa.U = mkview<double>(U, nu, Kc, L);
  pa.skip = nullptr; pa.live = nullptr; pa.live_mod = 1; pa.stop_failed = 0;
  OD_HIP(h->vt->rollout_policy(pa, cfg_of(h, P), h->stream));
  return OD_OK;
}

int od_ilqr_backward(od_handle h, long B, int T, int n, int m, const void* A, const void* Bm, const void* lxx,
                     const void* luu, const void* lux, const void* lx, const void* lu, const void* VxxT,
                     const void* VxT, double reg, void* K, void* k, void* dV, int* status) {
  if (!h) return fail(OD_ERR_INVALID, "od_ilqr_backward: null handle");
  if (B <= 0 || T <= 0) return OD_OK;
  if (n <= 0 || m <= 0 || n > OD_IL_N || m > OD_IL_M) return fail(OD_ERR_INVALID, "od_ilqr_backward: n <= 16, m <= 12");
  if (!A || !Bm || !lxx || !luu || !lux || !lx || !lu || !VxxT || !VxT || !K || !k || !dV)
    return fail(OD_ERR_INVALID, "od_ilqr_backward: null argument");
  OD_ON_DEVICE(h);
  const int L = h->layout;
  const long Kn = (long)T * B;
  IlqrArgs a;
  a.B = B; a.T = T; a.n = n; a.m = m; a.reg = reg;
  a.A = mkcview<double>(A, n * n, Kn, L); a.Bm = mkcview<double>(Bm, n * m, Kn, L);
  a.lxx = mkcview<double>(lxx, n * n, Kn, L); a.luu = mkcview<double>(luu, m * m, Kn, L);
  a.lux = mkcview<double>(lux, m * n, Kn, L); a.lx = mkcview<double>(lx, n, Kn, L); a.lu = mkcview<double>(lu, m, Kn, L);
  a.VxxT = mkcview<double>(VxxT, n * n, B, L); a.VxT = mkcview<double>(VxT, n, B, L);
  a.K = mkview<double>(K, m * n, Kn, L); a.k = mkview<double>(k, m, Kn, L);
  a.dV = mkview<double>(dV, 2, B, L); a.status = mkview<int>(status, 1, B, L);
  a.reg_dev = nullptr; a.retry = 0; a.skip = nullptr; a.done_b = nullptr;
#if defined(__HIPCC__)
  {
    // batch-minor data: TB consecutive trajectories per workgroup (coalesced), as many as 64 KB of LDS hold; batch-major data
    // (a trajectory's entries are contiguous): the one-trajectory kernel
    const size_t per = od_il_lds_per_traj(n, m) * sizeof(double);
    // ... but never fewer than 1024 workgroups while the batch allows (a workgroup of TB trajectories takes TB times as long:
    // measured on the rocket, T = 60, ms per iLQR iteration: 4096 trajectories 7.10 (one per workgroup) / 6.48 (two) / 5.67 (four) / 6.00 (eight);
    // 1024 trajectories 4.30 (one) / 4.53 (two) / 5.02 (eight))
    if (L == OD_LAYOUT_BATCH_MINOR && h->coop != 1) {        // (od_set_cooperative(h, 1): the LDS kernels below)
      bool done = true;
      const dim3 g = od_grid(B, OD_IL_THREADS / 16), blk(OD_IL_THREADS);
      if (h->coop == 0 && od_ilm_applies(a)) { OD_HIP(od_ilm_launch<double>(a, h->stream)); OD_HIP(hipGetLastError()); return OD_OK; }   // rocket, matrix cores
      if (n == 12 && m == 3) hipLaunchKernelGGL((k_ilqr_backward_row<12, 3>), g, blk, 0, h->stream, a);        // rocket
      else if (n == 8 && m == 2) hipLaunchKernelGGL((k_ilqr_backward_row<8, 2>), g, blk, 0, h->stream, a);     // hopper
      else if (n == 4 && m == 1) hipLaunchKernelGGL((k_ilqr_backward_row<4, 1>), g, blk, 0, h->stream, a);     // acrobot, cartpole
      else if (n == 10 && m == 2) hipLaunchKernelGGL((k_ilqr_backward_row<10, 2>), g, blk, 0, h->stream, a);   // planar push
      else done = false;
      if (done) { OD_HIP(hipGetLastError()); return OD_OK; }
    }
    int tb = (L != OD_LAYOUT_BATCH_MINOR) ? 1 : (8 * per <= 65536 ? 8 : (4 * per <= 65536 ? 4 : (2 * per <= 65536 ? 2 : 1)));
    while (tb > 1 && B / tb < 1024) tb >>= 1;
#define OD_IL_LAUNCH(TB_, N_, M_) hipLaunchKernelGGL((k_ilqr_backward_tb<TB_, N_, M_>), od_grid(B, TB_), dim3(OD_IL_THREADS), TB_ * per, h->stream, a)
#define OD_IL_SIZES(TB_)                                                                                              \
  do {                                                                                                                \
    if (n == 12 && m == 3) OD_IL_LAUNCH(TB_, 12, 3);      /* rocket */                                                \
    else if (n == 8 && m == 2) OD_IL_LAUNCH(TB_, 8, 2);   /* hopper */                                                \
    else if (n == 4 && m == 1) OD_IL_LAUNCH(TB_, 4, 1);   /* acrobot, cartpole */                                     \
    else if (n == 10 && m == 2) OD_IL_LAUNCH(TB_, 10, 2); /* planar push */                                           \
    else OD_IL_LAUNCH(TB_, 0, 0);                                                                                     \
  } while (0)
    if (tb == 8) OD_IL_SIZES(8);
    else if (tb == 4) OD_IL_SIZES(4);
    else if (tb == 2) OD_IL_SIZES(2);
    else hipLaunchKernelGGL((k_ilqr_backward<double>), dim3((unsigned)B), dim3(OD_IL_THREADS), 0, h->stream, a);
#undef OD_IL_SIZES
#undef OD_IL_LAUNCH
  }
#else
  hipLaunchKernelGGL((k_ilqr_backward_serial<double>), od_grid(B, OD_BLOCK), dim3(OD_BLOCK), 0, h->stream, a);
#endif
  OD_HIP(hipGetLastError());
  return OD_OK;
}

// ---- quadratic trajectory cost (od_quad_cost): one lane per trajectory, one pass over X and U -------------------------------
extern "C++" {
namespace {
template <class T> struct QuadCostArgs {
  long P; int Tn, n, m;
  View<const T> X, U;
  const double *Q, *R, *QT, *xref;
  double* J;
  const int* skip;     // device flag (may be null): non-zero = the launch does nothing
  const int* kstat;    // (may be null) status per knot (T * P): okall[p] = 1 iff bit 0 is set on every knot of trajectory p --
  int* okall;          // "every solve of this rollout converged", what the Armijo selection of the iLQR iteration asks
};
// 64 consecutive trajectories per workgroup; its four wavefronts take the knots t = w, w + 4, ... and their partial sums are added
// in a fixed order (the cost decides the Armijo test: it must not depend on scheduling).  N, M: compile-time sizes (0: any)
template <class T, int N, int M> __global__ __launch_bounds__(256) void k_quad_cost(QuadCostArgs<T> a) {
  if (a.skip && *a.skip) return;
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long p = (long)blockIdx.x * 64 + l;
  const long pp = p < a.P ? p : a.P - 1;
  const int n = N ? N : a.n, m = N ? M : a.m;
  constexpr int NV = N ? N : OD_IL_N, MV = N ? (M ? M : 1) : OD_IL_M;
#if defined(__HIPCC__)
  __shared__ double part[4][64];
  const int w0 = w, w1 = w + 1;
#else                                      // host test build (threads run one after the other): the first wavefront's lane does all four
  double part[4][64];
  if (w != 0) return;
  const int w0 = 0, w1 = 4;
#endif
  int okk = 1;
  for (int wq = w0; wq < w1; ++wq) {
  double J = 0.0;
  for (int t = wq; t <= a.Tn; t += 4) {
    const long ks = (long)t * a.P + pp;
    if (a.kstat && t < a.Tn) okk &= a.kstat[ks];
    double v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = i < n ? (double)a.X.at(i, ks) - a.xref[i] : 0.0;
    const double* Mx = (t < a.Tn) ? a.Q : a.QT;
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      double r = 0.0;
#pragma unroll
      for (int j = 0; j < NV; ++j) r += (i < n && j < n) ? Mx[i + n * j] * v[j] : 0.0;
      s += v[i] * r;
    }
    J += 0.5 * s;
    if (t < a.Tn) {
      double u[MV];
#pragma unroll
      for (int i = 0; i < MV; ++i) u[i] = i < m ? (double)a.U.at(i, ks) : 0.0;
      double su = 0.0;
#pragma unroll
      for (int i = 0; i < MV; ++i) {
        double r = 0.0;
#pragma unroll
        for (int j = 0; j < MV; ++j) r += (i < m && j < m) ? a.R[i + m * j] * u[j] : 0.0;
        su += u[i] * r;
      }
      J += 0.5 * su;
    }
  }
  part[wq][l] = J;
  }
#if defined(__HIPCC__)
  __shared__ int okp[4][64];
  okp[w][l] = okk;
  __syncthreads();
  if (w == 0) okk = okp[0][l] & okp[1][l] & okp[2][l] & okp[3][l];
#endif
  if (w == 0 && p < a.P && a.okall) a.okall[p] = okk & 1;
  if (w == 0 && p < a.P) a.J[p] = ((part[0][l] + part[1][l]) + part[2][l]) + part[3][l];
}
}  // namespace
}  // extern "C++"

int od_quad_cost(od_handle h, long P, int T, int n, int m, int dtype, const void* X, const void* U, const double* Q,
                 const double* R, const double* QT, const double* xref, double* J) {
  if (!h) return fail(OD_ERR_INVALID, "od_quad_cost: null handle");
  if (P <= 0) return OD_OK;
  if (T <= 0 || n <= 0 || m <= 0 || n > OD_IL_N || m > OD_IL_M) return fail(OD_ERR_INVALID, "od_quad_cost: T >= 1, n <= 16, m <= 12");
  if (dtype != OD_F64 && dtype != OD_F32) return fail(OD_ERR_INVALID, "od_quad_cost: dtype OD_F64 or OD_F32");
  if (!X || !U || !Q || !R || !QT || !xref || !J) return fail(OD_ERR_INVALID, "od_quad_cost: null argument");
  OD_ON_DEVICE(h);
  const int L = h->layout;
#define OD_QC_LAUNCH(T_, N_, M_) hipLaunchKernelGGL((k_quad_cost<T_, N_, M_>), od_grid(P, 64), dim3(256), 0, h->stream, a)
#define OD_QC_SIZES(T_)                                                                                   \
  do {                                                                                                    \
    if (n == 12 && m == 3) OD_QC_LAUNCH(T_, 12, 3);                                                       \
    else if (n == 8 && m == 2) OD_QC_LAUNCH(T_, 8, 2);                                                    \
    else if (n == 4 && m == 1) OD_QC_LAUNCH(T_, 4, 1);                                                    \
    else if (n == 10 && m == 2) OD_QC_LAUNCH(T_, 10, 2);                                                  \
    else OD_QC_LAUNCH(T_, 0, 0);                                                                          \
  } while (0)
  if (dtype == OD_F64) {
    QuadCostArgs<double> a{P, T, n, m, mkcview<double>(X, n, (long)(T + 1) * P, L), mkcview<double>(U, m, (long)T * P, L), Q, R, QT, xref, J, nullptr, nullptr, nullptr};
    OD_QC_SIZES(double);
  } else {
    QuadCostArgs<float> a{P, T, n, m, mkcview<float>(X, n, (long)(T + 1) * P, L), mkcview<float>(U, m, (long)T * P, L), Q, R, QT, xref, J, nullptr, nullptr, nullptr};
    OD_QC_SIZES(float);
  }
#undef OD_QC_SIZES
#undef OD_QC_LAUNCH
  OD_HIP(hipGetLastError());
  return OD_OK;
}

size_t od_bundle_workspace_bytes(od_handle h, long B, int N) {
  if (!h || B <= 0 || N <= 0) return 0;
  const size_t nq = h->vt->nq, nzb = 2 * nq + h->vt->nu;
  return sizeof(double) * (nq * (size_t)(N + 1) * (size_t)B + nzb * (nzb + nq) * (size_t)B) + sizeof(int) * (size_t)(N + 1) * (size_t)B;
}

static int run_ls(od_handle h, long B, int N, int ny, int nzb, const double* eta, View<const double> fv, double* acc,
                  void* M, int* status) {
  const long ne = (long)nzb * (nzb + ny);
#if defined(__HIPCC__)
  {
    View<double> Mf = mkview<double>(M, ny * nzb, B, h->layout);
    View<int> sf = mkview<int>(status, 1, B, h->layout);
    bool fused = true;
    if (nzb == 12 && ny == 5) hipLaunchKernelGGL((k_ls_fit_fused<12, 5>), dim3((unsigned)B), dim3(256), 0, h->stream, B, N, eta, fv, Mf, sf);        // planar push
    else if (nzb == 10 && ny == 4) hipLaunchKernelGGL((k_ls_fit_fused<10, 4>), dim3((unsigned)B), dim3(256), 0, h->stream, B, N, eta, fv, Mf, sf);   // hopper
    else if (nzb == 5 && ny == 2) hipLaunchKernelGGL((k_ls_fit_fused<5, 2>), dim3((unsigned)B), dim3(256), 0, h->stream, B, N, eta, fv, Mf, sf);     // acrobot, cartpole
    else fused = false;
    if (fused) { OD_HIP(hipGetLastError()); return OD_OK; }
  }
#endif
  hipLaunchKernelGGL(k_ls_accumulate, od_grid(B * ne, OD_BLOCK), dim3(OD_BLOCK), 0, h->stream, B, N, ny, nzb, eta, fv, acc);
  OD_HIP(hipGetLastError());
  View<double> Mv = mkview<double>(M, ny * nzb, B, h->layout);
  View<int> sv = mkview<int>(status, 1, B, h->layout);
  const dim3 grid = od_grid(B, OD_BLOCK), block(OD_BLOCK);
  if (nzb == 12 && ny == 5) hipLaunchKernelGGL((k_ls_solve_fixed<12, 5>), grid, block, 0, h->stream, B, (const double*)acc, Mv, sv);        // planar push
  else if (nzb == 10 && ny == 4) hipLaunchKernelGGL((k_ls_solve_fixed<10, 4>), grid, block, 0, h->stream, B, (const double*)acc, Mv, sv);   // hopper
  else if (nzb == 5 && ny == 2) hipLaunchKernelGGL((k_ls_solve_fixed<5, 2>), grid, block, 0, h->stream, B, (const double*)acc, Mv, sv);     // acrobot, cartpole
  else hipLaunchKernelGGL(k_ls_solve, grid, block, 0, h->stream, B, ny, nzb, (const double*)acc, Mv, sv);
  OD_HIP(hipGetLastError());
  return OD_OK;
}

int od_bundle_grad(od_handle h, long B, int N, const void* x, const void* u, const void* eta, void* dz,
                   void* ws, size_t ws_bytes, int* status) {
  if (int rc = check_mech(h, "od_bundle_grad")) return rc;
  if (B <= 0) return OD_OK;
  if (N <= 0 || !x || !eta || !dz || !ws) return fail(OD_ERR_INVALID, "od_bundle_grad: bad arguments");
  if (ws_bytes < od_bundle_workspace_bytes(h, B, N)) return fail(OD_ERR_INVALID, "od_bundle_grad: workspace too small");
  OD_ON_DEVICE(h);
  const int nq = h->vt->nq, nzb = 2 * nq + h->vt->nu;
  const long P = (long)(N + 1) * B;
  BundleArgs<double> a;
  a.s = step_args(h, B, B, x, u, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0);
  a.N = N;
  a.eta = (const double*)eta;
  a.feta = mkview<double>(ws, nq, P, OD_LAYOUT_BATCH_MINOR);
  a.status = mkview<int>(nullptr, 1, P, OD_LAYOUT_BATCH_MINOR);
  View<const double> fv;
  fv.p = a.feta.p; fv.se = a.feta.se; fv.sb = a.feta.sb;
  // workspace layout: feta (nq*P doubles) | normal-equation entries (nzb*(nzb+nq)*B doubles) | sample status (P ints)
  double* acc = (double*)ws + (size_t)nq * P;
  a.status.p = (int*)(acc + (size_t)nzb * (nzb + nq) * B);
  OD_HIP(h->vt->bundle(a, P, cfg_of(h, P), h->stream));
  return run_ls(h, B, N, nq, nzb, (const double*)eta, fv, acc, dz, status);
}

int od_ls_fit(od_handle h, long B, int N, int ny, int nzb, const void* eta, const void* feta, void* M, int* status) {
  if (!h) return fail(OD_ERR_INVALID, "od_ls_fit: null handle");
  if (B <= 0) return OD_OK;
  if (N <= 0 || ny <= 0 || nzb <= 0 || ny > OD_LS_MAX || nzb > OD_LS_MAX || !eta || !feta || !M)
    return fail(OD_ERR_INVALID, "od_ls_fit: bad arguments (ny, nzb <= 24)");
  OD_ON_DEVICE(h);
  View<const double> fv = mkcview<double>(feta, ny, (long)(N + 1) * B, OD_LAYOUT_BATCH_MINOR);
  if (int rc = ensure_work(h, (size_t)nzb * (nzb + ny) * (size_t)B)) return rc;
  h->grad_knots = 0;                                   // the workspace no longer holds gradient iterates
  return run_ls(h, B, N, ny, nzb, (const double*)eta, fv, h->work, M, status);
}

int od_model_indices(int model, int which, int* idx, int cap) {
  const ModelVT* vt = vt_of(model);
  if (!vt) return fail(OD_ERR_INVALID, "od_model_indices: unknown model");
  int n = 0;
  const int* src = nullptr;
  int zq[16];
  if (which == OD_IDX_CONFIGURATION) {
    // the solution block: the next configuration q3 (mechanical models) / next state (rocket) / projected control
    n = vt->nzq;
    for (int i = 0; i < n && i < 16; ++i) zq[i] = i;      // every model stores it first (csrc/gen/*.h: ZQ = 0..nzq-1)
    src = zq;
  } else if (which == OD_IDX_GAMMA) { n = vt->ngam; src = vt->gam.data(); }
  else if (which == OD_IDX_B) { n = vt->nbfr; src = vt->bfr.data(); }
  else return fail(OD_ERR_INVALID, "od_model_indices: unknown index set");
  if (idx) for (int i = 0; i < n && i < cap; ++i) idx[i] = src[i];
  return n;
}

int od_step_full(od_handle h, long B, const void* x, const void* u, void* z, void* dz, int* status, int* iters) {
  if (int rc = check_mech(h, "od_step_full")) return rc;
  if (B <= 0) return OD_OK;
  if (!x || !z || (h->vt->nu > 0 && !u)) return fail(OD_ERR_INVALID, "od_step_full: null x/u/z");
  if (dz && !fusable(h)) return fail(OD_ERR_UNSUPPORTED, "od_step_full: finite undercut with kappa_eval != kappa_grad");
  OD_ON_DEVICE(h);
  const ModelVT* vt = h->vt;
  FullArgs<double> a;
  a.s = step_args(h, B, B, x, u, nullptr, nullptr, nullptr, nullptr, status, iters, dz ? 1 : 0);
  a.z = mkview<double>(z, vt->nz, B, h->layout);
  a.dz = mkview<double>(dz, vt->nz * (2 * vt->nq + vt->nu), B, h->layout);
  OD_HIP(vt->step_full(a, ppw_of(h, B), h->stream));
  return OD_OK;
}

int od_ip_solve(od_handle h, long B, const void* z0, const void* theta, void* z, void* dz, int* status, int* iters) {
  if (!h) return fail(OD_ERR_INVALID, "od_ip_solve: null handle");
  if (B <= 0) return OD_OK;
  if (!z0 || !theta || !z) return fail(OD_ERR_INVALID, "od_ip_solve: null z0/theta/z");
  OD_ON_DEVICE(h);
  const ModelVT* vt = h->vt;
  const int L = h->layout;
  if (h->dtype == OD_F64) {
    RawArgs<double> a;
    a.B = B;
    a.opts = to_opts<double>(h->opts);
    if (!dz) a.opts.kappa_grad = a.opts.kappa_eval;
    a.z0 = mkcview<double>(z0, vt->nz, B, L);
    a.th = mkcview<double>(theta, vt->nth, B, L);
    a.z = mkview<double>(z, vt->nz, B, L);
    a.dz = mkview<double>(dz, vt->nzq * vt->ngc, B, L);
    a.status = mkview<int>(status, 1, B, L);
    a.iters = mkview<int>(iters, 2, B, L);
    a.want_grad = dz ? 1 : 0;
    OD_HIP(vt->raw64(a, ppw_of(h, B), h->stream));
  } else {
    RawArgs<float> a;
    a.B = B;
    a.opts = to_opts<float>(h->opts);
    if (!dz) a.opts.kappa_grad = a.opts.kappa_eval;
    a.z0 = mkcview<float>(z0, vt->nz, B, L);
    a.th = mkcview<float>(theta, vt->nth, B, L);
    a.z = mkview<float>(z, vt->nz, B, L);
    a.dz = mkview<float>(dz, vt->nzq * vt->ngc, B, L);
    a.status = mkview<int>(status, 1, B, L);
    a.iters = mkview<int>(iters, 2, B, L);
    a.want_grad = dz ? 1 : 0;
    OD_HIP(vt->raw32(a, ppw_of(h, B), h->stream));
  }
  return OD_OK;
}

int od_rocket_rollout(od_handle h, long B, int T, int nalpha, const void* alphas, int project, const void* x1,
                      const void* xbar, const void* ubar, const void* K, const void* kff, void* X, void* U, int* status) {
  if (!h) return fail(OD_ERR_INVALID, "od_rocket_rollout: null handle");
  if (h->vt->id != OD_ROCKET_DYNAMICS) return fail(OD_ERR_UNSUPPORTED, "od_rocket_rollout: needs an OD_ROCKET_DYNAMICS handle");
  if (B <= 0 || T <= 0) return OD_OK;
  if (!x1 || !ubar || !X) return fail(OD_ERR_INVALID, "od_rocket_rollout: null x1/ubar/X");
  if (nalpha > 0 && (!alphas || !xbar || !K || !kff)) return fail(OD_ERR_INVALID, "od_rocket_rollout: policy arguments missing");
  OD_ON_DEVICE(h);
  if (h->dtype == OD_F64) return rocket_rollout_impl<double>(h, B, T, nalpha, alphas, project, x1, xbar, ubar, K, kff, X, U, status);
  return rocket_rollout_impl<float>(h, B, T, nalpha, alphas, project, x1, xbar, ubar, K, kff, X, U, status);
}

int od_rocket(od_handle h, long B, int project, const void* x, const void* u, void* y, void* dx, void* du,
              void* uproj, int* status) {
  if (!h) return fail(OD_ERR_INVALID, "od_rocket: null handle");
  if (h->vt->id != OD_ROCKET_DYNAMICS) return fail(OD_ERR_UNSUPPORTED, "od_rocket: needs an OD_ROCKET_DYNAMICS handle");
  if (B <= 0) return OD_OK;
  if (!x || !u) return fail(OD_ERR_INVALID, "od_rocket: null input");
  OD_ON_DEVICE(h);
  if (h->dtype == OD_F64) return rocket_impl<double>(h, B, project, x, u, y, dx, du, uproj, status);
  return rocket_impl<float>(h, B, project, x, u, y, dx, du, uproj, status);
}

int od_soc_project(od_handle h, long B, const void* u, void* uproj, void* duproj, int* status) {
  if (!h) return fail(OD_ERR_INVALID, "od_soc_project: null handle");
  if (h->vt->id != OD_ROCKET_DYNAMICS) return fail(OD_ERR_UNSUPPORTED, "od_soc_project: needs an OD_ROCKET_DYNAMICS handle");
  if (B <= 0) return OD_OK;
  if (!u) return fail(OD_ERR_INVALID, "od_soc_project: null input");
  OD_ON_DEVICE(h);
  if (h->dtype == OD_F64) return soc_project_impl<double>(h, B, u, uproj, duproj, nullptr, status, nullptr);
  return soc_project_impl<float>(h, B, u, uproj, duproj, nullptr, status, nullptr);
}

int od_soc_project_full(od_handle h, long B, const void* u, void* z, void* duproj, int* status, int* iters) {
  if (!h) return fail(OD_ERR_INVALID, "od_soc_project_full: null handle");
  if (h->vt->id != OD_ROCKET_DYNAMICS) return fail(OD_ERR_UNSUPPORTED, "od_soc_project_full: needs an OD_ROCKET_DYNAMICS handle");
  if (B <= 0) return OD_OK;
  if (!u || !z) return fail(OD_ERR_INVALID, "od_soc_project_full: null u / z");
  OD_ON_DEVICE(h);
  if (h->dtype == OD_F64) return soc_project_impl<double>(h, B, u, nullptr, duproj, z, status, iters);
  return soc_project_impl<float>(h, B, u, nullptr, duproj, z, status, iters);
}

// ---- host scalar path ------------------------------------------------------------------------
// One knot on host vectors.  The staging area is PINNED, DEVICE-MAPPED HOST MEMORY (hipHostMalloc, coherent): the kernels read
// x, u and write d / fx / fu straight across the link, so a call is two host memcpys of a few hundred bytes, the launches and one
// synchronisation -- no hipMemcpy at all (round 2 staged through device memory with up to five small copies: 51 / 60 / 55 us per
// f / fx / fu call; now see DESIGN.md section 6).
static int host_call(od_handle h, const double* x, const double* u, double* d, double* dx, double* du) {
  if (int rc = check_mech(h, "od_f_host")) return rc;
  const int n = 2 * h->vt->nq, nu = h->vt->nu;
  if (!x || (nu > 0 && !u)) return fail(OD_ERR_INVALID, "od_f_host / od_fx_host / od_fu_host: null x / u");
  if (!d && !dx && !du) return fail(OD_ERR_INVALID, "od_f_host / od_fx_host / od_fu_host: no output buffer");
  OD_ON_DEVICE(h);
  const size_t need = (size_t)n + nu + n + (size_t)n * n + (size_t)n * nu;
  if (h->hstage_elems < need) {
    if (h->hstage) (void)hipHostFree(h->hstage);
    h->hstage = nullptr; h->hstage_elems = 0;
    OD_HIP(hipHostMalloc((void**)&h->hstage, need * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
    OD_HIP(hipHostGetDevicePointer((void**)&h->hstage_dev, h->hstage, 0));
    h->hstage_elems = need;
  }
  double* hx = h->hstage;                     // host view
  double* gx = h->hstage_dev;                 // the same bytes as the device sees them
  const size_t ou = n, od_ = ou + nu, oA = od_ + n, oB = oA + (size_t)n * n;
  std::memcpy(hx, x, n * sizeof(double));
  if (nu) std::memcpy(hx + ou, u, nu * sizeof(double));
  const int want_grad = (dx || du) ? 1 : 0;
  int rc = run_step(h, "od_f_host", 1, gx, gx + ou, d ? gx + od_ : nullptr, dx ? gx + oA : nullptr, du ? gx + oB : nullptr, nullptr,
                    nullptr, nullptr, want_grad);
  if (rc) return rc;
  OD_HIP(hipStreamSynchronize(h->stream));
  if (d) std::memcpy(d, hx + od_, n * sizeof(double));
  if (dx) std::memcpy(dx, hx + oA, (size_t)n * n * sizeof(double));
  if (du) std::memcpy(du, hx + oB, (size_t)n * nu * sizeof(double));
  return OD_OK;
}

int od_f_host(od_handle h, const double* x, const double* u, double* d) { return host_call(h, x, u, d, nullptr, nullptr); }
int od_fx_host(od_handle h, const double* x, const double* u, double* dx) { return host_call(h, x, u, nullptr, dx, nullptr); }
int od_fu_host(od_handle h, const double* x, const double* u, double* du) { return host_call(h, x, u, nullptr, nullptr, du); }
// f, fx and fu of one knot from ONE solve (the reference makes three, src/dynamics.jl:88,103,123); any output may be NULL
int od_ffxfu_host(od_handle h, const double* x, const double* u, double* d, double* dx, double* du) {
  return host_call(h, x, u, d, dx, du);
}

// device staging for host-pointer calls: `elems` doubles, grown on demand
static int ensure_stage(od_handle_s* h, size_t elems) {
  if (h->stage_elems >= elems) return OD_OK;
  if (h->stage) {
    if (hipStreamSynchronize(h->stream) != hipSuccess) return fail(OD_ERR_HIP, "hipStreamSynchronize failed");
    (void)hipFree(h->stage);
    h->stage = nullptr;
    h->stage_elems = 0;
  }
  OD_HIP(hipMalloc((void**)&h->stage, elems * sizeof(double)));
  h->stage_elems = elems;
  return OD_OK;
}

// f_rocket / fx_rocket / fu_rocket (project = 0) and the *_proj variants (project = 1) for one (x, u) on host vectors
// (src/models/rocket/dynamics.jl:101-164, 215-268); y 12, dx 12 x 12, du 12 x 3, uproj 3 (col-major); outputs may be NULL
// (an OD_F32 handle computes in float: the host doubles are converted on the way in and out -- the staging area holds
// elements of the handle's type)
extern "C++" {
template <class T> static int rocket_host_impl(od_handle h, int project, const double* x, const double* u, double* y, double* dx, double* du, double* uproj, int* status) {
  if (int rc = ensure_stage(h, 12 + 3 + 12 + 144 + 36 + 3 + 1)) return rc;
  T* px = (T*)h->stage; T* pu = px + 12; T* py = pu + 3; T* pdx = py + 12; T* pdu = pdx + 144; T* pup = pdu + 36;
  int* pst = (int*)((double*)h->stage + 12 + 3 + 12 + 144 + 36 + 3);
  T hx[15];
  for (int i = 0; i < 12; ++i) hx[i] = (T)x[i];
  for (int i = 0; i < 3; ++i) hx[12 + i] = (T)u[i];
  OD_HIP(hipMemcpyAsync(px, hx, 15 * sizeof(T), hipMemcpyHostToDevice, h->stream));
  OD_HIP(hipStreamSynchronize(h->stream));      // hx lives on this stack frame
  if (int rc = od_rocket(h, 1, project, px, pu, py, dx ? pdx : nullptr, du ? pdu : nullptr, pup, pst)) return rc;
  T out[12 + 144 + 36 + 3];
  OD_HIP(hipMemcpyAsync(out, py, sizeof(out), hipMemcpyDeviceToHost, h->stream));
  if (status) OD_HIP(hipMemcpyAsync(status, pst, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  OD_HIP(hipStreamSynchronize(h->stream));
  if (y) for (int i = 0; i < 12; ++i) y[i] = (double)out[i];
  if (dx) for (int i = 0; i < 144; ++i) dx[i] = (double)out[12 + i];
  if (du) for (int i = 0; i < 36; ++i) du[i] = (double)out[12 + 144 + i];
  if (uproj && project) for (int i = 0; i < 3; ++i) uproj[i] = (double)out[12 + 144 + 36 + i];
  return OD_OK;
}

}  // extern "C++"

int od_rocket_host(od_handle h, int project, const double* x, const double* u, double* y, double* dx, double* du, double* uproj, int* status) {
  if (!h || !x || !u) return fail(OD_ERR_INVALID, "od_rocket_host: null argument");
  OD_ON_DEVICE(h);
  if (h->dtype == OD_F32) return rocket_host_impl<float>(h, project, x, u, y, dx, du, uproj, status);
  return rocket_host_impl<double>(h, project, x, u, y, dx, du, uproj, status);
}

// soc_projection (duproj = NULL) / soc_projection_gradient for one u on host vectors (dynamics.jl:168-214)
extern "C++" {
template <class T> static int soc_project_host_impl(od_handle h, const double* u, double* uproj, double* duproj, int* status) {
  if (int rc = ensure_stage(h, 3 + 3 + 9 + 1)) return rc;
  T* pu = (T*)h->stage; T* pup = pu + 3; T* pd = pup + 3;
  int* pst = (int*)((double*)h->stage + 3 + 3 + 9);
  T hu[3] = {(T)u[0], (T)u[1], (T)u[2]};
  OD_HIP(hipMemcpyAsync(pu, hu, sizeof(hu), hipMemcpyHostToDevice, h->stream));
  OD_HIP(hipStreamSynchronize(h->stream));
  if (int rc = od_soc_project(h, 1, pu, pup, duproj ? pd : nullptr, pst)) return rc;
  T out[12];
  OD_HIP(hipMemcpyAsync(out, pup, sizeof(out), hipMemcpyDeviceToHost, h->stream));
  if (status) OD_HIP(hipMemcpyAsync(status, pst, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  OD_HIP(hipStreamSynchronize(h->stream));
  if (uproj) for (int i = 0; i < 3; ++i) uproj[i] = (double)out[i];
  if (duproj) for (int i = 0; i < 9; ++i) duproj[i] = (double)out[3 + i];
  return OD_OK;
}

}  // extern "C++"

int od_soc_project_host(od_handle h, const double* u, double* uproj, double* duproj, int* status) {
  if (!h || !u) return fail(OD_ERR_INVALID, "od_soc_project_host: null argument");
  OD_ON_DEVICE(h);
  if (h->dtype == OD_F32) return soc_project_host_impl<float>(h, u, uproj, duproj, status);
  return soc_project_host_impl<double>(h, u, uproj, duproj, status);
}

// gradient!(sim, gb, q1, q2, u1) for one knot on host vectors (src/gradient_bundle.jl:87-104): x = [q1; q2], eta
// (2nq+nu) x N col-major, dz nq x (2nq+nu) col-major.  status: 1 = every sample converged and the fit is determined
int od_bundle_grad_host(od_handle h, int N, const double* x, const double* u, const double* eta, double* dz, int* status) {
  if (int rc = check_mech(h, "od_bundle_grad_host")) return rc;
  if (!x || !eta || !dz || N <= 0 || (h->vt->nu > 0 && !u)) return fail(OD_ERR_INVALID, "od_bundle_grad_host: null argument");
  OD_ON_DEVICE(h);
  const int nq = h->vt->nq, n = 2 * nq, nu = h->vt->nu, nzb = n + nu;
  const size_t ws = (od_bundle_workspace_bytes(h, 1, N) + 7) / 8;
  if (int rc = ensure_stage(h, (size_t)n + nu + (size_t)nzb * N + (size_t)nq * nzb + 1 + ws)) return rc;
  double* px = h->stage; double* pu = px + n; double* pe = pu + nu; double* pdz = pe + (size_t)nzb * N;
  int* pst = (int*)(pdz + (size_t)nq * nzb);
  double* pws = pdz + (size_t)nq * nzb + 1;
  OD_HIP(hipMemcpyAsync(px, x, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
  if (nu) OD_HIP(hipMemcpyAsync(pu, u, nu * sizeof(double), hipMemcpyHostToDevice, h->stream));
  OD_HIP(hipMemcpyAsync(pe, eta, (size_t)nzb * N * sizeof(double), hipMemcpyHostToDevice, h->stream));
  if (int rc = od_bundle_grad(h, 1, N, px, pu, pe, pdz, pws, ws * 8, pst)) return rc;
  OD_HIP(hipMemcpyAsync(dz, pdz, (size_t)nq * nzb * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (status) OD_HIP(hipMemcpyAsync(status, pst, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  OD_HIP(hipStreamSynchronize(h->stream));
  return OD_OK;
}

}  // extern "C"

#include <vector>
#include "od_ilqr_solver.inc"

#include "od_comm.inc"
