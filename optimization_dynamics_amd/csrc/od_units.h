// Per-problem "units of work" = what one GPU lane does.  Each unit mirrors one reference entry
// point; the HIP kernels in od_kernels.hip just map lanes to units.  (tests/ also compiles this
// header on the host to check the logic against the oracle; that harness is not shipped.)
#pragma once
#include "od_solver.h"

namespace od {

// strided view: element e of problem b lives at p[e*se + b*sb]
//   batch-minor SoA (coalesced across lanes): se = batch capacity, sb = 1
//   batch-major  (Julia n x B matrix)       : se = 1, sb = elements per problem
#if defined(__HIP_DEVICE_COMPILE__)
// keeps the compiler from rewriting a running pointer as base + e*stride with one hoisted scalar
// offset per element: those offsets (dozens per view) otherwise live in SGPRs across the whole
// interior-point loop and are spilled to VGPR lanes (v_writelane/v_readlane)
#define OD_OPAQUE_PTR(p) asm volatile("" : "+v"(p))
#define OD_GLOBAL_PTR(T, p) ((T __attribute__((address_space(1)))*)(p))
// small tables of constants the host wrote before the launch, read at a wavefront-uniform address inside a long loop: through the
// constant address space (scalar loads, their own counter, the scalar cache) from a pointer the compiler cannot look through --
// read where they are used instead of living in SGPRs (and, spilled, in VGPR lanes) across the loop
#define OD_CONST_PTR(T, p) ((const T __attribute__((address_space(4)))*)(p))
#define OD_OPAQUE_SPTR(p) asm volatile("" : "+s"(p))
#else
#define OD_OPAQUE_PTR(p) (void)0
#define OD_GLOBAL_PTR(T, p) (p)
#define OD_CONST_PTR(T, p) (p)
#define OD_OPAQUE_SPTR(p) (void)0
#endif

template <class T> struct View {
  T* p;
  long se, sb;
  OD_HD T& at(long e, long b) const { return p[e * se + b * sb]; }
  OD_HD bool ok() const { return p != nullptr; }
  // sequential access to elements 0, 1, 2, ... of problem b
  struct Cursor {
    T* q;
    long se;
    // (the opaque pointer has lost its address space: say "global" again, or the accesses become FLAT)
    OD_HD void put(T v) { *OD_GLOBAL_PTR(T, q) = v; q += se; OD_OPAQUE_PTR(q); }
    OD_HD T get() { const T v = *OD_GLOBAL_PTR(T, q); q += se; OD_OPAQUE_PTR(q); return v; }
    OD_HD void skip(int n) { q += n * se; OD_OPAQUE_PTR(q); }
  };
  OD_HD Cursor cursor(long b) const { Cursor c{p + b * sb, se}; OD_OPAQUE_PTR(c.q); return c; }
};

// ---- where the KKT factors of one problem live -------------------------------------------------------------
// RegFact: the generated Fact struct (registers; what does not fit spills to scratch).  LdsFact: LDS, 16 lane slots
// per 64-thread workgroup, for models whose factors do not fit the register file anyway (planar push: 325
// doubles per lane = the whole of its 2.5 KB scratch frame).  LDS is ~10x closer than scratch, but 29 KB per
// wavefront allow only 5 wavefronts per CU, so the launcher uses it when the batch fits one round of the 1024 SIMDs
// (od_model_tu.inc).  Accesses go through an opaque offset so that the compiler cannot keep the values in
// registers (and spill those) after all.
template <class M, class T> struct RegFactStore {
  using Fact = typename M::template Fact<T>;
  OD_HD static Fact make() { return Fact{}; }
};
#if defined(__HIP_DEVICE_COMPILE__)
template <class M, class T> struct LdsFactStore {
  static constexpr int SLOTS = 16;
  struct Fact {
    struct Arr {
      T* base;
      __device__ __forceinline__ T& operator[](int i) const {
        unsigned o = (unsigned)i * SLOTS;
        asm volatile("" : "+v"(o));
        return base[o];
      }
    } v;
    int piv[M::MTAIL > 0 ? M::MTAIL : 1];
    bool sw[M::NSWAP > 0 ? M::NSWAP : 1];
  };
  __device__ __forceinline__ static Fact make() {
    // the kernels that use this store run the interior-point iterations only (pass 1 defers its gradient, the bundle
    // takes none): the state program's slots, not the larger pivoted program's -- planar push 229 instead of 325 doubles
    // per lane = 29 KB instead of 42 KB per wavefront, five wavefronts per CU instead of three: the 804 wavefronts of
    // BASELINE config 3 are then resident at once (768 were, the last 36 waited for a second round)
    __shared__ T lds_fact[(M::NFACT_S > 0 ? M::NFACT_S : 1) * SLOTS];
    Fact f;
    f.v.base = lds_fact + (threadIdx.x & (SLOTS - 1));
    return f;
  }
};
#else   // the host test build has no LDS: same interface, registers
template <class M, class T> struct LdsFactStore : RegFactStore<M, T> {};
#endif

// ---- f / fx / fu (src/dynamics.jl:81-128) ------------------------------------------------------
// Two passes.  Pass 1 ("state"): the interior-point solve of every knot (sequential in t for
// rollouts), which records per knot the iterate z_g at which the reference's grad simulator would
// stop and the clamp differentiate_solution! would use ((nz+1) doubles).  Pass 2 ("grad"): the
// implicit-function solve of every knot, one lane per knot, fully parallel.  The gradient of knot t
// does not feed knot t+1, so it is off the rollout's critical path and its arrays are out of the
// solve loop's register budget.
template <class T> struct StepArgs {
  long B;
  T h;
  T fric[4];
  Opts<T> opts;
  View<const T> x;   // 2nq per problem: [q1; q2]
  View<const T> u;   // nu
  View<T> d;         // 2nq: [q2; q3]                                  (f)
  View<T> q3;        // nq: q3 alone (compact output, alternative to d)
  View<T> dx;        // 2nq x 2nq col-major, ALL entries written        (fx)
  View<T> du;        // 2nq x nu  col-major, ALL entries written        (fu)
  View<T> dq3;       // compact nq x (2nq+nu) col-major = d q3/d(q1,q2,u) (alternative to dx/du)
  View<int> status;  // bit0 eval converged, bit1 grad converged, bit2 factorisation ok
  View<int> iters;   // 2 per problem: iterations to kappa_eval / kappa_grad
  View<T> zg;        // workspace, nz+1 per problem: gradient iterate and clamp (pass 1 -> pass 2)
  int want_grad;
  int merge_grad_status;   // 1: this is the separate grad solve of a non-fusable step: merge into status / iters[1]
};

// The linearisation inside the device-resident iLQR iteration (od_ilqr_solver.inc) runs the kernels of od_step_grad on the T*B knots of
// the nominal trajectories: a launch-level skip flag (every trajectory has converged) and a per-trajectory predicate (only the
// trajectories that took a step have new states) ride beside the step arguments as a kernel parameter of their own -- StepArgs is
// also what the rollout kernels take and stays as it is.  All null for every other caller.
struct LiveArgs {
  const int* skip;    // non-zero = the launch does nothing
  const int* live;    // knot k belongs to trajectory k % live_mod and is computed only if live[k % live_mod] != 0
  long live_mod;
  OD_HD bool dead(long k) const { return live && !live[k % live_mod]; }
};

// pass-2 output: dq3/d(q1,q2,u) scattered into the reference's dx / du layout (and/or compact dq3)
template <class M, class T> struct StepSink {
  static constexpr bool DEFER_GRAD = false;
  static constexpr bool FULL_STATE = false;
  const StepArgs<T>& a;
  long b;
  OD_HD void defer(const T*, T) {}
  OD_HD void grad(int i, int c, T v) {
    constexpr int nq = M::NQ, n = 2 * M::NQ;
    if (a.dq3.ok()) a.dq3.at(i + nq * c, b) = v;
    if (c < n) { if (a.dx.ok()) a.dx.at((nq + i) + n * c, b) = v; }
    else { if (a.du.ok()) a.du.at((nq + i) + n * (c - n), b) = v; }
  }
};

// pass-1 sink: record where the gradient has to be taken
template <class M, class T> struct DeferSink {
  static constexpr bool DEFER_GRAD = true;
  static constexpr bool FULL_STATE = false;
  const View<T>& zg;
  long k;
  OD_HD void grad(int, int, T) {}
  OD_HD void defer(const T* z, T reg) {
    auto c = zg.cursor(k);
#pragma unroll
    for (int i = 0; i < M::NZ; ++i) c.put(z[i]);
    c.put(reg);
  }
};

// pass 1 for one knot k given its state and control in registers; returns q3 in q3out
template <class M, class T, class Store = RegFactStore<M, T>>
OD_HD int knot_state(const StepArgs<T>& a, long k, const T* xin, const T* uin, T* q3out) {
  constexpr int nq = M::NQ;
  T th[M::NTH], z[M::NZ];
  mech_setup<M>(xin, xin + nq, uin, a.fric, a.h, th, z);
  DeferSink<M, T> sink{a.zg, k};
  int it[2];
  auto f = Store::make();
  const int st = ip_step_grad<M, T, DeferSink<M, T>>(a.opts, th, z, true, a.want_grad != 0, sink, it, f);
#pragma unroll
  for (int i = 0; i < nq; ++i) q3out[i] = z[M::ZQ[i]];
  if (a.d.ok()) {
    auto c = a.d.cursor(k);
#pragma unroll
    for (int i = 0; i < nq; ++i) c.put(xin[nq + i]);
#pragma unroll
    for (int i = 0; i < nq; ++i) c.put(q3out[i]);
  }
  if (a.q3.ok()) {
    auto c = a.q3.cursor(k);
#pragma unroll
    for (int i = 0; i < nq; ++i) c.put(q3out[i]);
  }
  if (a.merge_grad_status) {
    // the eval solve already wrote its bits: add GRAD_OK of this solve, keep FACTOR_OK only if both factorised
    if (a.status.ok()) { const int e = a.status.at(0, k); a.status.at(0, k) = (e & ~OD_ST_FACTOR_OK) | (st & OD_ST_GRAD_OK) | (e & st & OD_ST_FACTOR_OK); }
    if (a.iters.ok()) { auto c = a.iters.cursor(k); c.skip(1); c.put(it[1]); }
  } else {
    if (a.status.ok()) a.status.at(0, k) = st;
    if (a.iters.ok()) { auto c = a.iters.cursor(k); c.put(it[0]); c.put(it[1]); }
  }
  return st;
}

// independent knots (od_step, od_step_grad pass 1)
template <class M, class T, class Store = RegFactStore<M, T>> OD_HD void unit_step_state(const StepArgs<T>& a, long b) {
  constexpr int nq = M::NQ, n = 2 * M::NQ;
  T x[n], u[M::NU > 0 ? M::NU : 1], q3[nq];
#pragma unroll
  for (int i = 0; i < n; ++i) x[i] = a.x.at(i, b);
#pragma unroll
  for (int i = 0; i < M::NU; ++i) u[i] = a.u.at(i, b);
  knot_state<M, T, Store>(a, b, x, u, q3);
}

// rollouts (od_rollout pass 1): T sequential knots per trajectory, state carried in registers.
// Knot index of (t, b) is k = t*B + b; X has T+1 slots per trajectory, slot 0 = x1.
template <class T> struct RolloutArgs {
  StepArgs<T> s;     // x = x1 (2nq per trajectory); u / d / status / iters / zg are per-knot views
  View<T> x0;        // slot 0 of X (receives a copy of x1)
  int Tn;
};

template <class M, class T, class Store = RegFactStore<M, T>> OD_HD void unit_rollout_state(const RolloutArgs<T>& ra, long b) {
  constexpr int nq = M::NQ, n = 2 * M::NQ;
  const StepArgs<T>& a = ra.s;
  T x[n], u[M::NU > 0 ? M::NU : 1], q3[nq];
#pragma unroll
  for (int i = 0; i < n; ++i) x[i] = a.x.at(i, b);
  if (ra.x0.ok()) {
#pragma unroll
    for (int i = 0; i < n; ++i) ra.x0.at(i, b) = x[i];
  }
  // the control of knot t+1 is requested before knot t is solved: its latency (and the drain of knot
  // t-1's stores that an in-order wait would include) stays off the recursion's critical path
  T un[M::NU > 0 ? M::NU : 1];
  {
    auto c = a.u.cursor(b);
#pragma unroll
    for (int i = 0; i < M::NU; ++i) un[i] = c.get();
  }
  for (int t = 0; t < ra.Tn; ++t) {
    const long k = (long)t * a.B + b;
#pragma unroll
    for (int i = 0; i < M::NU; ++i) u[i] = un[i];
    if (t + 1 < ra.Tn) {
      auto c = a.u.cursor(k + a.B);
#pragma unroll
      for (int i = 0; i < M::NU; ++i) un[i] = c.get();
    }
    knot_state<M, T, Store>(a, k, x, u, q3);
#pragma unroll
    for (int i = 0; i < nq; ++i) { x[i] = x[nq + i]; x[nq + i] = q3[i]; }
  }
}

// ---- closed-loop rollout = forward pass of iLQR (examples/*.jl: iLQR.solve! line search) ----
// Candidate trajectory p = a*Bnom + b follows nominal trajectory b with step size alphas[a]:
//   u_t = ubar_t + alpha * k_t + K_t (x_t - xbar_t).
// All step sizes of the Armijo line search are rolled out speculatively in one launch.
template <class T> struct PolicyArgs {
  RolloutArgs<T> r;      // r.s.B = P = Bnom*nalpha candidates; r.s.x = x1 of the Bnom nominal trajectories
  long Bnom;
  int nalpha;
  const T* alphas;
  View<const T> xbar;    // 2nq per slot, (T+1)*Bnom slots
  View<const T> ubar;    // nu per nominal knot
  View<const T> K;       // nu x 2nq col-major per nominal knot
  View<const T> kff;     // nu per nominal knot
  View<T> U;             // nu per candidate knot: controls actually applied
  const int* skip;       // device flag (may be null): non-zero = the launch does nothing (device-resident iLQR iteration, od_ilqr_solver.inc)
  const int* live;       // (may be null) per nominal trajectory: candidate p is rolled out only if live[p % live_mod] != 0
  long live_mod;
  int stop_failed;       // non-zero (the forward pass of od_ilqr_*): a candidate ends at its first knot whose solve did not converge; the
                         // status of its remaining knots is written as 0, their states and controls are left as they were.  The Armijo
                         // selection takes only rollouts whose every knot converged (k_il_select), so the rest of such a rollout is never
                         // looked at -- and a diverged candidate would otherwise keep its wavefront waiting for a 100-iteration solve at
                         // every remaining knot.  0 (od_rollout_policy): every knot of every candidate, as asked
};
// (shared by the three forms of the kernel: the candidate's remaining knots are marked not converged)
template <class T> OD_HD void policy_mark_rest_failed(const PolicyArgs<T>& pa, int t, long p) {
  const StepArgs<T>& a = pa.r.s;
  if (a.status.ok()) for (int t2 = t + 1; t2 < pa.r.Tn; ++t2) a.status.at(0, (long)t2 * a.B + p) = 0;
}

template <class M, class T> OD_HD void unit_rollout_policy(const PolicyArgs<T>& pa, long p) {
  constexpr int nq = M::NQ, n = 2 * M::NQ, nu = M::NU > 0 ? M::NU : 1;
  const StepArgs<T>& a = pa.r.s;
  const long b = p % pa.Bnom;
  const T alpha = pa.alphas[p / pa.Bnom];
  T x[n], u[nu], q3[nq];
#pragma unroll
  for (int i = 0; i < n; ++i) x[i] = a.x.at(i, b);
  if (pa.r.x0.ok()) {
#pragma unroll
    for (int i = 0; i < n; ++i) pa.r.x0.at(i, p) = x[i];
  }
  for (int t = 0; t < pa.r.Tn; ++t) {
    const long kn = (long)t * pa.Bnom + b, kc = (long)t * a.B + p;
    T dx[n];
    {
      auto c = pa.xbar.cursor(kn);
#pragma unroll
      for (int i = 0; i < n; ++i) dx[i] = x[i] - c.get();
    }
    {
      auto cu = pa.ubar.cursor(kn);
      auto ck = pa.kff.cursor(kn);
      auto cK = pa.K.cursor(kn);
#pragma unroll
      for (int j = 0; j < M::NU; ++j) u[j] = cu.get() + alpha * ck.get();
#pragma unroll
      for (int i = 0; i < n; ++i) {            // K is nu x n col-major: column i multiplies dx[i]
#pragma unroll
        for (int j = 0; j < M::NU; ++j) u[j] += cK.get() * dx[i];
      }
      auto co = pa.U.cursor(kc);
#pragma unroll
      for (int j = 0; j < M::NU; ++j) co.put(u[j]);
    }
    const int st = knot_state<M, T>(a, kc, x, u, q3);
    if (pa.stop_failed && !(st & OD_ST_EVAL_OK)) { policy_mark_rest_failed(pa, t, p); break; }
#pragma unroll
    for (int i = 0; i < nq; ++i) { x[i] = x[nq + i]; x[nq + i] = q3[i]; }
  }
}

// pass 2: knot k reads its state (a.x, slot k) and control, rebuilds theta, and differentiates at the
// recorded iterate; writes the constant blocks of fx / fu as well ([0 I] on top, src/dynamics.jl:105-108)
template <class M, class T> OD_HD void unit_grad_knot(const StepArgs<T>& a, long k) {
  constexpr int nq = M::NQ, n = 2 * M::NQ;
  T x[n], u[M::NU > 0 ? M::NU : 1], th[M::NTH], z[M::NZ];
#pragma unroll
  for (int i = 0; i < n; ++i) x[i] = a.x.at(i, k);
#pragma unroll
  for (int i = 0; i < M::NU; ++i) u[i] = a.u.at(i, k);
  mech_setup<M>(x, x + nq, u, a.fric, a.h, th, z);
#pragma unroll
  for (int i = 0; i < M::NZ; ++i) z[i] = a.zg.at(i, k);
  const T reg = a.zg.at(M::NZ, k);
  StepSink<M, T> sink{a, k};
  const bool ok = gradient_at<M>(th, z, reg, sink);
  if (a.dx.ok()) {
#pragma unroll
    for (int c = 0; c < n; ++c) {
#pragma unroll
      for (int i = 0; i < nq; ++i) a.dx.at(i + n * c, k) = (c == nq + i) ? T(1) : T(0);
    }
  }
  if (a.du.ok()) {
#pragma unroll
    for (int c = 0; c < M::NU; ++c) {
#pragma unroll
      for (int i = 0; i < nq; ++i) a.du.at(i + n * c, k) = T(0);
    }
  }
  if (!ok && a.status.ok()) a.status.at(0, k) = a.status.at(0, k) & ~OD_ST_FACTOR_OK;
}

template <class T> struct NoGradSink {
  static constexpr bool DEFER_GRAD = true;     // (never asked for a gradient: no factor<true> / solve<true> code in the kernel)
  static constexpr bool FULL_STATE = false;
  OD_HD void grad(int, int, T) {}
  OD_HD void defer(const T*, T) {}
};

// ---- gradient bundle samples (src/gradient_bundle.jl:87-100) ---------------------------------
// problem p = b*(N+1) + i : i = 0 nominal, i >= 1 perturbed by eta[:, i-1]; EVAL simulator, no grad.
template <class T> struct BundleArgs {
  StepArgs<T> s;          // x,u per knot b; outputs unused except opts/h/fric
  int N;
  const T* eta;           // (2nq+nu) x N col-major, shared by all knots
  View<T> feta;           // nq per problem p
  View<int> status;       // per problem p
  // device-resident iLQR with the bundle as linearisation (od_ilqr_solver.inc): skip non-zero = the launch does nothing; the samples of
  // knot k are computed only if live[k % live_mod] != 0 (the trajectory took a step: the others' samples stand as they are).  Null otherwise.
  const int* skip = nullptr;
  const int* live = nullptr;
  long live_mod = 1;
  OD_HD bool dead(long p) const { return live && !live[(p / (N + 1)) % live_mod]; }
};

template <class M, class T, class Store = RegFactStore<M, T>> OD_HD void unit_bundle_sample(const BundleArgs<T>& ba, long p) {
  constexpr int nq = M::NQ, n = 2 * M::NQ, nzb = 2 * M::NQ + M::NU;
  const StepArgs<T>& a = ba.s;
  const long b = p / (ba.N + 1);
  const int i = (int)(p - b * (ba.N + 1));
  T x[n], u[M::NU > 0 ? M::NU : 1];
#pragma unroll
  for (int k = 0; k < n; ++k) x[k] = a.x.at(k, b) + (i > 0 ? ba.eta[k + nzb * (i - 1)] : T(0));
#pragma unroll
  for (int k = 0; k < M::NU; ++k) u[k] = a.u.at(k, b) + (i > 0 ? ba.eta[n + k + nzb * (i - 1)] : T(0));
  T th[M::NTH], z[M::NZ];
  mech_setup<M>(x, x + nq, u, a.fric, a.h, th, z);
  NoGradSink<T> sink;
  int it[2];
  auto f = Store::make();
  const int st = ip_step_grad<M, T, NoGradSink<T>>(a.opts, th, z, true, false, sink, it, f);
#pragma unroll
  for (int k = 0; k < nq; ++k) ba.feta.at(k, p) = z[M::ZQ[k]];
  if (ba.status.ok()) ba.status.at(0, p) = st;
}

// ---- whole solution of a step: z at kappa_eval and dz/d(q1, q2, u1) at kappa_grad, every row --------------
// What RoboDojo's `process!` reads contact forces and their sensitivities from (sim.traj.gamma / b, sim.grad.dgamma1d*,
// db1d*: sized in src/dynamics.jl:36-46).  One fused launch (not the two-pass split): a feature
// path, not the hot one.
template <class T> struct FullArgs {
  StepArgs<T> s;      // x, u, opts, h, fric
  View<T> z;          // NZ
  View<T> dz;         // NZ x NGC col-major
};
template <class M, class T> struct FullSink {
  static constexpr bool DEFER_GRAD = false;
  static constexpr bool FULL_STATE = true;
  static constexpr bool ALL_ROWS = true;
  const FullArgs<T>& a;
  long b;
  OD_HD void defer(const T*, T) {}
  OD_HD void grad(int i, int c, T v) { if (a.dz.ok()) a.dz.at(i + M::NZ * c, b) = v; }
};
template <class M, class T> OD_HD void unit_step_full(const FullArgs<T>& a, long b) {
  constexpr int nq = M::NQ, n = 2 * M::NQ;
  T x[n], u[M::NU > 0 ? M::NU : 1], th[M::NTH], z[M::NZ];
#pragma unroll
  for (int i = 0; i < n; ++i) x[i] = a.s.x.at(i, b);
#pragma unroll
  for (int i = 0; i < M::NU; ++i) u[i] = a.s.u.at(i, b);
  mech_setup<M>(x, x + nq, u, a.s.fric, a.s.h, th, z);
  FullSink<M, T> sink{a, b};
  int it[2];
  const int st = ip_step_grad<M>(a.s.opts, th, z, true, a.s.want_grad != 0, sink, it);
#pragma unroll
  for (int i = 0; i < M::NZ; ++i) a.z.at(i, b) = z[i];
  if (a.s.status.ok()) a.s.status.at(0, b) = st;
  if (a.s.iters.ok()) { a.s.iters.at(0, b) = it[0]; a.s.iters.at(1, b) = it[1]; }
}

// ---- raw interior-point solve on user-supplied (z0, theta): rocket dynamics / projection ------
// (src/models/rocket/dynamics.jl:101-210).  dz: NZQ x NGC col-major (rows ZQ, first NGC theta cols)
template <class T> struct RawArgs {
  long B;
  Opts<T> opts;
  View<const T> z0;   // NZ
  View<const T> th;   // NTH
  View<T> z;          // NZ (solution)
  View<T> dz;         // NZQ*NGC
  View<int> status, iters;
  int want_grad;
};

template <class M, class T> struct RawSink {
  static constexpr bool DEFER_GRAD = false;
  static constexpr bool FULL_STATE = true;
  const RawArgs<T>& a;
  long b;
  OD_HD void defer(const T*, T) {}
  OD_HD void grad(int i, int c, T v) { if (a.dz.ok()) a.dz.at(i + M::NZQ * c, b) = v; }
};

template <class M, class T> OD_HD void unit_raw(const RawArgs<T>& a, long b) {
  T th[M::NTH], z[M::NZ];
#pragma unroll
  for (int i = 0; i < M::NTH; ++i) th[i] = a.th.at(i, b);
#pragma unroll
  for (int i = 0; i < M::NZ; ++i) z[i] = a.z0.at(i, b);
  RawSink<M, T> sink{a, b};
  int it[2];
  const int st = ip_step_grad<M>(a.opts, th, z, true, a.want_grad != 0, sink, it);
#pragma unroll
  for (int i = 0; i < M::NZ; ++i) a.z.at(i, b) = z[i];
  if (a.status.ok()) a.status.at(0, b) = st;
  if (a.iters.ok()) { a.iters.at(0, b) = it[0]; a.iters.at(1, b) = it[1]; }
}

// ---- rocket f / fx / fu with optional thrust-cone projection (dynamics.jl:101-268) -----------
template <class T> struct RocketArgs {
  long B;
  T h, u_max;
  Opts<T> opts_dyn, opts_proj;
  int project;        // 0: f_rocket*, 1: f_rocket_proj*
  int want_grad;
  View<const T> x;    // 12
  View<const T> u;    // 3
  View<T> y;          // 12                      (f_rocket / f_rocket_proj)
  View<T> dx;         // 12 x 12 col-major       (fx_*)
  View<T> du;         // 12 x 3  col-major       (fu_*; projected: dz_dyn[:,u] * dproj[1:3,1:3])
  View<T> uproj;      // 3 (projected control; optional)
  View<int> status;   // bit0/1 dyn eval/grad ok, bit4/5 projection eval/grad ok
  // device-resident iLQR iteration (od_ilqr_solver.inc); all null otherwise
  const int* skip;    // non-zero = the launch does nothing
  const int* live;    // per trajectory: knot b belongs to trajectory b % live_mod and is computed only if live[b % live_mod] != 0
  long live_mod;
  int proj_stall_exit; // 1: a projection solve that has stalled at the boundary of the cone is abandoned (od_solver.h::model_stall)
  int polish64;        // single-precision handles: 1 = the dynamics solution refined with the residual in double (rocket_refine64) and the implicit gradient from a double factorisation there (rocket_grad64); the thrust-cone projection solved in double (soc_project_solve)
  double h64;          // the time step in double (h is rounded to T)
  Opts<double> opts_proj64;   // the projection's options as a double-precision handle has them (soc_project_solve)
};

// d(projected u)/du: 3 x 3 col-major kept in registers (column index is dynamic -> selects)
template <class T> struct ProjSink {
  static constexpr bool DEFER_GRAD = false;
  static constexpr bool FULL_STATE = true;
  T* d;
  OD_HD void defer(const T*, T) {}
  OD_HD void grad(int i, int c, T v) {
#pragma unroll
    for (int k = 0; k < 3; ++k) d[i + 3 * k] = (c == k) ? v : d[i + 3 * k];
  }
};
// dz_dyn: x-columns go straight to dx, the three u-columns are collected for the chain product
template <class T> struct RocketDynSink {
  static constexpr bool DEFER_GRAD = false;
  static constexpr bool FULL_STATE = true;
  const RocketArgs<T>& a;
  long b;
  T* dyn_u;
  OD_HD void defer(const T*, T) {}
  OD_HD void grad(int i, int c, T v) {
    if (c < 12) { if (a.dx.ok()) a.dx.at(i + 12 * c, b) = v; }
    else {
#pragma unroll
      for (int k = 0; k < 3; ++k) dyn_u[i + 12 * k] = (c - 12 == k) ? v : dyn_u[i + 12 * k];
    }
  }
};

// Mixed precision for the single-precision handles (BASELINE config 5; SURVEY.md section 7: "keep the final Newton refinement in
// fp64").  The single-precision Newton iteration of the dynamics step stops at r_tol = 1e-4 -- 1e-8 is below the resolution of
// float -- and its implicit gradient carries the conditioning of a float LU.  The solution is refined with the residual of the SAME
// equations in double at the float solution (inputs x, u as the floats they are, the time step in double; rocket_refine64 below), and
// -rz^{-1} rtheta is taken from a double factorisation at the refined point (rocket_grad64); both are rounded to float on the way
// out, so the results are the double-precision answers to float resolution (6e-8) -- inside the north_star's 1e-6 / 1e-4 -- for one
// factorisation of a 12 x 12 system with 44 factor entries.  The thrust-cone projection stays in single precision: its result is only
// kappa_tol = 1e-4 accurate by construction.
template <class S> struct CastSink64 {
  S& s;
  OD_HD void grad(int i, int c, double v) { s.grad(i, c, (float)v); }
};
// the implicit gradient -rz^{-1} rtheta in double AT the (refined, float-rounded) solution y: residual (for the shared trigonometric
// values), Jacobian, factorisation and the NGC solves all at that point.  (Round 4 took the gradient from the factorisation of the
// point BEFORE the polishing step: the float iteration stops at r_tol = 1e-4, so that point can sit 1e-4 away from the solution and the
// gradient inherited an error of that size times the curvature -- one knot in 25 000 at 1.01e-4 relative, found by the round-5 sweep,
// tests/parity_checks.py::check_rocket_sweep.)
template <class MD, class Sink> OD_HD bool rocket_grad64(const float* x, const float* u, double h, const float* y, Sink& sink) {
  double th[MD::NTH], z[MD::NZ], r[MD::NZ], pre[MD::NPRE > 0 ? MD::NPRE : 1], tr[MD::NTR > 0 ? MD::NTR : 1];
#pragma unroll
  for (int i = 0; i < 12; ++i) { th[i] = (double)x[i]; z[i] = (double)y[i]; }
  th[12] = (double)u[0]; th[13] = (double)u[1]; th[14] = (double)u[2]; th[15] = h;
  MD::eval_pre(th, pre);
  MD::eval_r(z, th, pre, tr, r);
  typename MD::template Fact<double> f;
  const bool ok = eval_factor<MD>(z, th, pre, tr, 0.0, f);
  double g[MD::NNZTH];
  MD::eval_rth(z, th, pre, tr, g);
  for (int c = 0; c < MD::NGC; ++c) {
    double b[MD::NZ];
#pragma unroll
    for (int i = 0; i < MD::NZ; ++i) b[i] = 0.0;
#pragma unroll
    for (int k = 0; k < MD::NNZTH; ++k) b[MD::RTH_ROW[k]] = (MD::RTH_COL[k] == c) ? g[k] : b[MD::RTH_ROW[k]];
    MD::solve(f, b, b);
#pragma unroll
    for (int i = 0; i < MD::NZQ; ++i) sink.grad(i, c, -b[MD::ZQ[i]]);
  }
  return ok;
}

// The same refinement where only the state is wanted (the rollout kernels, 60 sequential knots per trajectory): the residual in
// double at the single-precision solution, the correction through the SINGLE-precision factors the Newton iteration left behind
// (mixed-precision iterative refinement: the error contracts by |J_float^{-1} J - I| ~ 1e-3 per step, from ~1e-6 to ~1e-9) -- one
// residual evaluation in double and one float back-solve instead of a double factorisation.
template <class MD, class F32> OD_HD void rocket_refine64(const float* x, const float* u, double h, const float* th32, float* y, F32& f, bool have_fact) {
  double th[MD::NTH], z[MD::NZ], r[MD::NZ], pre[MD::NPRE > 0 ? MD::NPRE : 1], tr[MD::NTR > 0 ? MD::NTR : 1];
#pragma unroll
  for (int i = 0; i < 12; ++i) { th[i] = (double)x[i]; z[i] = (double)y[i]; }
  th[12] = (double)u[0]; th[13] = (double)u[1]; th[14] = (double)u[2]; th[15] = h;
  MD::eval_pre(th, pre);
  MD::eval_r(z, th, pre, tr, r);
  float rf[MD::NZ], D[MD::NZ];
#pragma unroll
  for (int i = 0; i < MD::NZ; ++i) rf[i] = (float)r[i];
  if (!have_fact) {           // (the float iteration converged at its starting point: nothing factorised yet)
    float pre32[MD::NPRE > 0 ? MD::NPRE : 1], tr32[MD::NTR > 0 ? MD::NTR : 1], r32[MD::NZ];
    MD::eval_pre(th32, pre32);
    MD::eval_r(y, th32, pre32, tr32, r32);
    eval_factor<MD>(y, th32, pre32, tr32, 0.0f, f);
  }
  MD::solve(f, rf, D);
#pragma unroll
  for (int i = 0; i < MD::NZ; ++i) y[i] = (float)(z[i] - (double)D[i]);
}

// The thrust-cone projection of a single-precision handle under od_set_mixed_precision (the default): the whole solve in double, on the
// float inputs, rounded to float on the way out -- the projected control and its gradient are the double-precision handle's to float
// resolution (6e-8).  An interior-point solve has no "solution" to polish: it returns the first iterate below kappa_tol, a point of the
// path, and the path is ill-conditioned in its own rounding -- every iteration multiplies the duality measure by (1 - alpha) with
// alpha = 0.97-0.999, so a relative error e of alpha becomes e / (1 - alpha) of the measure, and next to the apex of the cone the
// iterate moves like its square root.  Measured on the host build (tools/proj_paths_host.py, profiles/r6_projection_paths_host.json): a
// float solve agrees with the double one to 1e-6 on 65 % of apex-heavy controls (2e-3 on all); a variant that ran the float iterations
// down to k_vio < 1 (or 100: the first three) and the remaining ones in double reached 90 % and was dropped -- three float iterations at
// the start are enough to lose it.  Round 5's float projection remains under od_set_mixed_precision(h, 0).
template <class S> struct CastSinkFull64 {
  static constexpr bool DEFER_GRAD = false;
  static constexpr bool FULL_STATE = true;
  S& s;
  OD_HD void defer(const double*, double) {}
  OD_HD void grad(int i, int c, double v) { s.grad(i, c, (float)v); }
};
// zp: in = the initial guess soc_projection prescribes (MP::ZI_VAL), out = the whole solution vector; GRAD = false: kernels that never take the projection's gradient
template <class MP, bool GRAD, class T, class Sink>
OD_HD int soc_project_solve(const RocketArgs<T>& a, const T* thp, T* zp, bool want_grad, Sink& sink, int* itp) {
  if constexpr (sizeof(T) == 4) {
    if (a.polish64) {
      double z[MP::NZ], th[MP::NTH];
#pragma unroll
      for (int i = 0; i < MP::NZ; ++i) z[i] = MP::ZI_VAL[i];      // (the initial point of soc_projection in double: 0.1 is not a float)
#pragma unroll
      for (int i = 0; i < MP::NTH; ++i) th[i] = (double)thp[i];
      int st;
      if constexpr (GRAD) {
        CastSinkFull64<Sink> cs{sink};
        st = ip_step_grad<MP>(a.opts_proj64, th, z, true, want_grad, cs, itp, a.proj_stall_exit != 0);
      } else {
        NoGradSink<double> ns;             // (state only: the projected control z[0..2] is all the rollout kernels take, and all the snapshot keeps)
        st = ip_step_grad<MP>(a.opts_proj64, th, z, true, false, ns, itp, a.proj_stall_exit != 0);
      }
#pragma unroll
      for (int i = 0; i < MP::NZ; ++i) zp[i] = (float)z[i];
      return st;
    }
  }
  return ip_step_grad<MP>(a.opts_proj, thp, zp, true, want_grad, sink, itp, a.proj_stall_exit != 0);
}

// one rocket knot: (x, u) in registers -> y in registers; per-knot outputs (dx, du, uproj, status) at index b
// GRADS = false (the rollout kernels): state only -- no gradient code in the kernel at all (it would never run there, but its
// arrays would still shape the register allocation of the time recursion)
template <class MD, class MP, class T> OD_HD void rocket_knot_state(const RocketArgs<T>& a, long b, const T* x, T* u, T* y, int* ok_and = nullptr) {
  int st = 0;
  NoGradSink<T> ns;
  if (a.project) {
    T zp[MP::NZ], thp[MP::NTH];
#pragma unroll
    for (int i = 0; i < MP::NZ; ++i) zp[i] = T(MP::ZI_VAL[i]);
    thp[0] = u[0]; thp[1] = u[1]; thp[2] = u[2]; thp[3] = a.u_max;
    int itp[2];
    const int sp_ = soc_project_solve<MP, false>(a, thp, zp, false, ns, itp);
    st |= (sp_ & 1) << 4;
    u[0] = zp[0]; u[1] = zp[1]; u[2] = zp[2];
    if (a.uproj.ok()) { a.uproj.at(0, b) = u[0]; a.uproj.at(1, b) = u[1]; a.uproj.at(2, b) = u[2]; }
  }
  T th[MD::NTH];
#pragma unroll
  for (int i = 0; i < 12; ++i) { th[i] = x[i]; y[i] = x[i]; }
  th[12] = u[0]; th[13] = u[1]; th[14] = u[2]; th[15] = a.h;
  int it[2];
  typename MD::template Fact<T> fd;
  const int sd = ip_step_grad<MD, T, NoGradSink<T>>(a.opts_dyn, th, y, true, false, ns, it, fd);
  if constexpr (sizeof(T) == 4) {
    if (a.polish64) rocket_refine64<MD>(x, u, a.h64, th, y, fd, it[0] > 0);
  }
  st |= (sd & (OD_ST_EVAL_OK | OD_ST_FACTOR_OK));
  if (a.status.ok()) a.status.at(0, b) = st;
  if (ok_and) *ok_and &= st;                  // (bit 0: the dynamics solve converged)
}

template <class MD, class MP, class T>
OD_HD void rocket_knot(const RocketArgs<T>& a, long b, const T* x, T* u, T* y) {
  int st = 0;
  T dproj[9];   // d uproj / d u  (3x3 col-major)
  if (a.project) {
    // soc_projection (dynamics.jl:168-186): z .= 0.1; z[3]+=1; z[10]+=1; z[7]=0; theta=[u; u_max]
    T zp[MP::NZ], thp[MP::NTH];
#pragma unroll
    for (int i = 0; i < MP::NZ; ++i) zp[i] = T(MP::ZI_VAL[i]);
    thp[0] = u[0]; thp[1] = u[1]; thp[2] = u[2]; thp[3] = a.u_max;
    ProjSink<T> ps{dproj};
#pragma unroll
    for (int i = 0; i < 9; ++i) dproj[i] = T(0);
    int itp[2];
    const int sp_ = soc_project_solve<MP, true>(a, thp, zp, a.want_grad != 0, ps, itp);
    st |= (sp_ & 3) << 4;
    u[0] = zp[0]; u[1] = zp[1]; u[2] = zp[2];
    if (a.uproj.ok()) { a.uproj.at(0, b) = u[0]; a.uproj.at(1, b) = u[1]; a.uproj.at(2, b) = u[2]; }
  }
  // f_rocket (dynamics.jl:101-114): z0 = x, theta = [x; u; h]
  T th[MD::NTH];
#pragma unroll
  for (int i = 0; i < 12; ++i) { th[i] = x[i]; y[i] = x[i]; }
  th[12] = u[0]; th[13] = u[1]; th[14] = u[2]; th[15] = a.h;
  // du with projection needs the 12x3 block times dproj: collect the u-columns in registers
  T dyn_u[36];
  RocketDynSink<T> ds{a, b, dyn_u};
#pragma unroll
  for (int i = 0; i < 36; ++i) dyn_u[i] = T(0);
  int it[2];
  int sd;
  if constexpr (sizeof(T) == 4) {
    if (a.polish64) {
      // the float Newton iteration, its solution refined with the residual in double through the float factors (rocket_refine64: ~1e-9),
      // then the gradient from a double factorisation AT that solution
      NoGradSink<T> ns;
      typename MD::template Fact<T> fd;
      sd = ip_step_grad<MD, T, NoGradSink<T>>(a.opts_dyn, th, y, true, false, ns, it, fd);
      rocket_refine64<MD>(x, u, a.h64, th, y, fd, it[0] > 0);
      if (a.want_grad) {
        CastSink64<RocketDynSink<T>> cs{ds};
        if (!rocket_grad64<MD>(x, u, a.h64, y, cs)) sd &= ~OD_ST_FACTOR_OK;
        if (sd & OD_ST_EVAL_OK) sd |= OD_ST_GRAD_OK;
      }
    } else {
      sd = ip_step_grad<MD>(a.opts_dyn, th, y, true, a.want_grad != 0, ds, it);
    }
  } else {
    sd = ip_step_grad<MD>(a.opts_dyn, th, y, true, a.want_grad != 0, ds, it);
  }
  st |= (sd & 7);
  if (a.want_grad && a.du.ok()) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
      for (int i = 0; i < 12; ++i) {
        T v;
        if (a.project) {   // mul!(du, du_dyn_cache, du_proj_cache)  (dynamics.jl:264-267)
          v = dyn_u[i] * dproj[3 * c] + dyn_u[i + 12] * dproj[1 + 3 * c] + dyn_u[i + 24] * dproj[2 + 3 * c];
        } else {
          v = dyn_u[i + 12 * c];
        }
        a.du.at(i + 12 * c, b) = v;
      }
    }
  }
  if (a.status.ok()) a.status.at(0, b) = st;
}

// soc_projection / soc_projection_gradient alone (dynamics.jl:168-214): u -> uproj (3), d uproj / d u (3 x 3
// col-major, written to `du`); x, y, dx unused.  status bits 16 / 32 as in od_rocket.
// (od_soc_project_full: the whole solution z of the projection's interior-point solve -- the iterate its gradient was taken at, the two
// tolerances of this solve being equal -- and the iterations it took: what a checker needs to recompute -rz^{-1} rtheta there)
template <class T> struct SocProjectArgs {
  RocketArgs<T> a;
  View<T> z;          // 10 (optional)
  View<int> iters;    // 1 (optional)
};
template <class MP, class T> OD_HD void unit_soc_project(const SocProjectArgs<T>& sa, long b) {
  const RocketArgs<T>& a = sa.a;
  T zp[MP::NZ], thp[MP::NTH], dproj[9];
#pragma unroll
  for (int i = 0; i < MP::NZ; ++i) zp[i] = T(MP::ZI_VAL[i]);
#pragma unroll
  for (int i = 0; i < 3; ++i) thp[i] = a.u.at(i, b);
  thp[3] = a.u_max;
#pragma unroll
  for (int i = 0; i < 9; ++i) dproj[i] = T(0);
  ProjSink<T> ps{dproj};
  int itp[2];
  const int sp_ = soc_project_solve<MP, true>(a, thp, zp, a.want_grad != 0, ps, itp);
  if (a.uproj.ok()) {
#pragma unroll
    for (int i = 0; i < 3; ++i) a.uproj.at(i, b) = zp[i];
  }
  if (a.want_grad && a.du.ok()) {
#pragma unroll
    for (int i = 0; i < 9; ++i) a.du.at(i, b) = dproj[i];
  }
  if (sa.z.ok()) {
#pragma unroll
    for (int i = 0; i < MP::NZ; ++i) sa.z.at(i, b) = zp[i];
  }
  if (sa.iters.ok()) sa.iters.at(0, b) = itp[0];
  if (a.status.ok()) a.status.at(0, b) = (sp_ & 3) << 4;
}

template <class MD, class MP, class T> OD_HD void unit_rocket(const RocketArgs<T>& a, long b) {
  T x[12], u[3], y[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) x[i] = a.x.at(i, b);
#pragma unroll
  for (int i = 0; i < 3; ++i) u[i] = a.u.at(i, b);
  rocket_knot<MD, MP, T>(a, b, x, u, y);
  if (a.y.ok()) {
#pragma unroll
    for (int i = 0; i < 12; ++i) a.y.at(i, b) = y[i];
  }
}

// rocket rollouts (iLQR.rollout / forward pass over f_rocket or f_rocket_proj, examples/rocket.jl:29-41,118):
// candidate p = a*Bnom + b; with nalpha = 0 the rollout is open loop (controls from ubar, knot index t*Bnom + b),
// otherwise u_t = ubar_t + alpha k_t + K_t (x_t - xbar_t).  State only (a.want_grad = 0); X gets T+1 slots.
template <class T> struct RocketRolloutArgs {
  RocketArgs<T> a;       // a.B = P candidates; a.x = x1 of the Bnom nominal trajectories; a.y = X view shifted by one slot
  View<T> x0;            // slot 0 of X
  int Tn;
  long Bnom;
  int nalpha;
  const T* alphas;
  View<const T> xbar, ubar, K, kff;
  View<T> U;             // controls applied (before projection), per candidate knot
  // device-resident iLQR iteration with a DIAGONAL quadratic objective (od_ilqr_solver.inc): the cost of every candidate is summed
  // up along its rollout -- J_p = sum_t 1/2 (x_t - xref)'Q(x_t - xref) + 1/2 u_t'R u_t + 1/2 (x_T - xref)'QT(x_T - xref), in double,
  // knot after knot (a fixed order: the cost decides the Armijo comparison) -- instead of a second pass over the candidates'
  // states; okall[p] = every dynamics solve of the rollout converged.  All null otherwise.
  const double *cq, *cr, *cqt, *cxref;   // diagonals of Q (12), R (3) -- contiguous: cr = cq + 12 --, QT (12); xref (12)
  double* J;
  int* okall;
};

template <class MD, class MP, class T> OD_HD void unit_rocket_rollout(const RocketRolloutArgs<T>& ra, long p) {
  const RocketArgs<T>& a = ra.a;
  const long b = ra.nalpha > 0 ? p % ra.Bnom : p;
  const T alpha = ra.nalpha > 0 ? ra.alphas[p / ra.Bnom] : T(0);
  T x[12], u[3], y[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) x[i] = a.x.at(i, b);
  if (ra.x0.ok()) {
#pragma unroll
    for (int i = 0; i < 12; ++i) ra.x0.at(i, p) = x[i];
  }
  double Jacc = 0.0;
  int okk = 1;
  // Everything a knot reads and writes goes through running pointers (View::Cursor: one 64-bit add per element): indexing the views
  // with (element, knot) leaves ~70 hoisted row offsets in SGPRs across the recursion, and with the 27 constants of the objective
  // the kernel spilled 357 SGPRs to VGPR lanes -- ~1 500 of the ~5 300 instructions of a knot (tools/isa_attrib.py rocket).
  for (int t = 0; t < ra.Tn; ++t) {
    const long kn = (long)t * ra.Bnom + b, kc = (long)t * a.B + p;
    {
      auto c = ra.ubar.cursor(kn);
#pragma unroll
      for (int j = 0; j < 3; ++j) u[j] = c.get();
    }
    if (ra.nalpha > 0) {
      auto cf = ra.kff.cursor(kn);
#pragma unroll
      for (int j = 0; j < 3; ++j) u[j] += alpha * cf.get();
      auto cx_ = ra.xbar.cursor(kn);
      auto ck = ra.K.cursor(kn);
#pragma unroll
      for (int i = 0; i < 12; ++i) {
        const T dxi = x[i] - cx_.get();
#pragma unroll
        for (int j = 0; j < 3; ++j) u[j] += ck.get() * dxi;          // (element j + 3 i)
      }
    }
    if (ra.U.ok()) {
      auto c = ra.U.cursor(kc);
#pragma unroll
      for (int j = 0; j < 3; ++j) c.put(u[j]);
    }
    if (ra.J) {
      auto cq = OD_CONST_PTR(double, ra.cq);       // diag Q (12), diag R (3) behind it
      auto cx = OD_CONST_PTR(double, ra.cxref);
      OD_OPAQUE_SPTR(cq); OD_OPAQUE_SPTR(cx);
      double s = 0.0;
#pragma unroll
      for (int i = 0; i < 12; ++i) { const double v = (double)x[i] - cx[i]; s += cq[i] * v * v; }
#pragma unroll
      for (int j = 0; j < 3; ++j) { const double v = (double)u[j]; s += cq[12 + j] * v * v; }
      Jacc += 0.5 * s;
    }
    rocket_knot_state<MD, MP, T>(a, kc, x, u, y, ra.okall ? &okk : nullptr);
    if (a.y.ok()) {
      auto c = a.y.cursor(kc);
#pragma unroll
      for (int i = 0; i < 12; ++i) c.put(y[i]);
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) x[i] = y[i];
  }
  if (ra.J) {
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 12; ++i) { const double v = (double)x[i] - ra.cxref[i]; s += ra.cqt[i] * v * v; }
    ra.J[p] = Jacc + 0.5 * s;
  }
  if (ra.okall) ra.okall[p] = okk;
}

}  // namespace od
