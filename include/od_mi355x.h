/*
 * od_mi355x.h -- C ABI of the MI355X-native batched implicit-dynamics engine (libod_mi355x.so).
 *
 * Drop-in boundary for the hot path of thowell/optimization_dynamics: the per-timestep interior-
 * point solve + implicit-function gradient behind the iLQR dynamics callbacks.  Each entry point
 * names the reference interface it replaces (paths relative to the reference repository).
 * The reference is pure Julia; the binding a maintainer would add is a `ccall` shim
 * (julia/OptimizationDynamicsMI355X.jl).  tests/ bind the same symbols via ctypes.
 *
 * Conventions
 *  - All batched entry points take DEVICE pointers and are asynchronous on the handle's HIP stream
 *    (od_set_stream); the *_host entry points take host pointers, copy, launch and synchronise.
 *  - Return value: 0 on success, negative od_error otherwise; od_last_error() gives the message.
 *    Solver non-convergence is NOT an error (the reference only returns a Bool and still copies the
 *    result out, src/models/rocket/dynamics.jl:178-186): it is reported per problem in `status`.
 *  - Per-problem arrays are addressed as element e of problem k:
 *      OD_LAYOUT_BATCH_MINOR (default):  base[e * K + k]   (K = number of problems in the array;
 *                                         lanes of a wavefront touch consecutive addresses)
 *      OD_LAYOUT_BATCH_MAJOR:            base[k * E + e]   (E = elements per problem; this is a
 *                                         Julia/Fortran  E x K  column-major matrix)
 *    Matrices per problem (dx, du, dz) are column-major inside their E elements, like the reference's
 *    Julia matrices.  For rollouts the problem index of knot t of trajectory b is k = t*B + b.
 *  - Element type is double for OD_F64 handles and float for OD_F32 handles.
 *  - Devices: a handle belongs to the HIP device that was current when od_create made it (od_get_device).  Every entry point
 *    that takes the handle (or a solver made from it) switches the calling thread to that device for the duration of the call and
 *    restores the caller's current device on the way out, so a process that drives several GPUs needs no hipSetDevice around its
 *    calls.  Pointers passed in must be accessible from the handle's device; a stream of another device is refused
 *    (OD_ERR_WRONG_DEVICE).
 *  - status bits: 1 = state converged to (r_tol, kappa_eval_tol), 2 = gradient iterate converged to
 *    (r_tol, kappa_grad_tol), 4 = all KKT factorisations were non-singular.
 */
#ifndef OD_MI355X_H
#define OD_MI355X_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct od_handle_s* od_handle;

enum od_error {
  OD_OK = 0, OD_ERR_INVALID = -1, OD_ERR_UNSUPPORTED = -2, OD_ERR_HIP = -3, OD_ERR_NO_DEVICE = -4, OD_ERR_WRONG_DEVICE = -5
};

/* model ids: src/models/<model>; hopper is RoboDojo.hopper (examples/hopper.jl:14) */
enum od_model {
  OD_ACROBOT_IMPACT = 0, OD_ACROBOT_NOMINAL = 1, OD_CARTPOLE_FRICTION = 2, OD_CARTPOLE_FRICTIONLESS = 3,
  OD_PLANAR_PUSH = 4, OD_ROCKET_DYNAMICS = 5, OD_ROCKET_PROJECTION = 6, OD_HOPPER = 7, OD_NUM_MODELS = 8
};
enum od_dtype { OD_F64 = 0, OD_F32 = 1 };
enum od_layout { OD_LAYOUT_BATCH_MINOR = 0, OD_LAYOUT_BATCH_MAJOR = 1 };
enum od_status_bits { OD_STATUS_EVAL_OK = 1, OD_STATUS_GRAD_OK = 2, OD_STATUS_FACTOR_OK = 4 };

/* InteriorPointOptions as set by get_simulator (src/dynamics.jl:25-33) and RocketInfo
 * (src/models/rocket/dynamics.jl:21-27,77-86).  kappa_eval_tol / kappa_grad_tol are the two
 * ImplicitDynamics keywords (src/dynamics.jl:51-53).  undercut = INFINITY is allowed. */
typedef struct {
  double r_tol, kappa_eval_tol, kappa_grad_tol;
  int max_iter, max_ls;
  double eps_min, kappa_reg, gamma_reg, undercut;
} od_options;

/* ABI version of this header; od_version() returns the library's.  Bumped whenever a struct passed by pointer changes size or a default
 * changes behaviour (100 -> 101: od_ilqr_options gained rho_max and proj_stall_exit, handles start with the projection's stall exit off,
 * the od_comm_* entry points; a single-precision handle solves the thrust-cone projection in double under od_set_mixed_precision).
 * Bindings compare the two at load (optimization_dynamics_amd/_lib.py, julia/OptimizationDynamicsMI355X.jl). */
#define OD_ABI_VERSION 101
int od_version(void);
const char* od_last_error(void);

/* model table: dimensions of src/models/<model> (nq, nu, nz = num_var, ntheta = num_data, nfric) */
int od_model_dims(int model, int* nq, int* nu, int* nz, int* ntheta, int* nfric);
const char* od_model_name(int model);
/* models are addressed by id; ids 0..7 are the enum above, models added with the generator
 * (python -m optimization_dynamics_amd.codegen --add spec.py, the counterpart of deps/build.jl:27-48) follow */
int od_default_friction(int model, double* mu, int n);   /* friction_coefficients(model) of the generated model, n <= 4 */
int od_num_models(void);
int od_model_id(const char* name);   /* -1 if unknown */
/* option preset of the example that uses the model (examples/<model>.jl) */
int od_default_options(int model, od_options* out);

/* ImplicitDynamics(model, h, r, rz, rtheta; r_tol, kappa_eval_tol, kappa_grad_tol, ...)
 * (src/dynamics.jl:51-79): one handle = eval_sim + grad_sim of one model.  opts may be NULL. */
int od_create(int model, int dtype, const od_options* opts, double h, od_handle* out);
int od_destroy(od_handle h);
/* the HIP device the handle lives on (the device that was current in od_create) */
int od_get_device(od_handle h, int* device);
int od_set_options(od_handle h, const od_options* opts);
int od_get_options(od_handle h, od_options* out);
int od_set_timestep(od_handle h, double dt);
/* friction_coefficients(model) (e.g. cartpole_friction.friction .= [0.35; 0.35], examples/cartpole.jl:21) */
int od_set_friction(od_handle h, const double* mu, int n);
/* RocketInfo.u_max (src/models/rocket/dynamics.jl:8) */
int od_set_u_max(od_handle h, double u_max);
/* The thrust-cone projection (soc_projection, src/models/rocket/dynamics.jl:168-186, eps_min = 0) can stall on the boundary of the
 * cone away from the solution -- accepted step lengths ~1e-13 for the rest of its max_iter iterations, result reported as not
 * converged (status bits 16 / 32 clear), in the CPU oracle alike; ~0.02 % of random controls.
 * on = 0 (DEFAULT): every iteration, as the reference runs them -- od_rocket, od_soc_project and od_rocket_rollout return what the
 * reference's loop returns, status bits included.
 * on = 1: such a solve is abandoned once its accepted step length has been below 1e-9 (OD_F32: 1e-3, csrc/od_rocket_proj_direct.h)
 * for 4 consecutive iterations -- status bits as if max_iter had been reached -- so that a lockstep wavefront does not wait ~90
 * iterations for it.  A solve that does not stall is not touched (bit-identical results).  Of the stalled ones the full loop
 * rescues about a third by accumulated rounding drift (50 iterations at alpha ~ 1e-13, then convergence in three steps): with the
 * exit they are reported as not converged instead; in OD_F32 the 1e-3 threshold also abandons the odd solve that would have
 * converged late.  The device-resident iLQR solver switches the exit on for its own launches (od_ilqr_options.proj_stall_exit,
 * default 1: a forward pass is 60 projections deep in lockstep), whatever the handle's setting. */
int od_set_projection_stall_exit(od_handle h, int on);
/* OD_F32 rocket handles (BASELINE config 5 asks for single precision): on = 1 (default) refines every dynamics step with the residual
 * of the same equations in double at the single-precision solution and takes the implicit gradient -rz^{-1} rtheta from a double
 * factorisation there; and (since ABI 101) solves the thrust-cone projection in double on the float inputs -- an interior-point solve
 * returns a point of its path, not a root, so a float solve agrees with the double one only to 2e-3 next to the apex of the cone and
 * no final polish can repair it.  Results are rounded to float on the way out: states, projected controls and gradients are the
 * double-precision handle's to float resolution (within the 1e-6 / 1e-4 bars).  on = 0: single precision throughout (states 5e-4,
 * projected controls 2e-3, gradients 2e-2), the faster forward pass.  No effect on OD_F64. */
int od_set_mixed_precision(od_handle h, int on);
int od_set_layout(od_handle h, int layout);
/* A handle runs on one stream at a time (its gradient hand-over and staging workspaces are reused by consecutive
 * calls): changing the stream first waits for the work queued on the previous one.  Use one handle per stream for
 * concurrency. */
int od_set_stream(od_handle h, void* hip_stream);
/* launch tuning of the solve pass.  ppw: problems per 64-lane wavefront (power of two <= 64, 0 = automatic)
 * -- a wavefront is as slow as its slowest lane, so small batches are spread over more wavefronts.
 * waves_per_block: 1 or 4 wavefronts per workgroup (4 = one per SIMD of a CU), 0 = automatic. */
int od_set_launch_config(od_handle h, int ppw, int waves_per_block);
/* cooperative solve pass (one problem per 16 or per 8 lanes, the contacts / cones of a problem spread over them): shortens
 * the critical path of a problem.  mode 0 = automatic (per model and batch size), 1 = never (lane-per-problem kernels),
 * 2 = always where the model has them (the 16-lane form where it has both), 3 = always, the 8-lane form first.
 * Results agree with the lane-per-problem kernels to rounding; which kernel the automatic mode picks depends on the batch
 * size, so pin a mode where results must not depend on it at rounding level.  od_ilqr_backward follows the same switch: mode 1
 * keeps its workgroup (LDS) kernels, modes 2 and 3 take the one-trajectory-per-16-lanes kernel where it exists (m <= 4), mode 0
 * runs the models' sizes (n, m) = (12, 3), (8, 2), (10, 2), (4, 1) on the matrix cores, one wavefront per trajectory (batch-minor
 * layout; csrc/od_ilqr_mfma.inc). */
int od_set_cooperative(od_handle h, int mode);
/* diagnostics: the iterate at which the last od_step_grad* / od_rollout* call on this handle differentiated each of its
 * K knots -- z at the first iterate satisfying (r_tol, kappa_grad) and, in row nz, the clamp of the orthant variables
 * RoboDojo's differentiate_solution! uses -- as (nz+1) x K, batch-minor, into a device buffer (a copy of the hand-over
 * workspace of the two-pass scheme).  Lets a checker recompute dz = -rz^{-1} rtheta at exactly that point. */
int od_get_grad_iterates(od_handle h, long K, void* out);
/* 1 if a solve pass over B problems would run the cooperative kernels under the handle's current settings */
int od_uses_cooperative(od_handle h, long B);
int od_synchronize(od_handle h);

/* f (src/dynamics.jl:81-94) for B knots: d = [q2; q3].  x: 2nq, u: nu, d: 2nq per problem.
 * status, iters (2 ints per problem: iterations to kappa_eval / kappa_grad) may be NULL. */
int od_step(od_handle h, long B, const void* x, const void* u, void* d, int* status, int* iters);

/* f + fx + fu (src/dynamics.jl:81-128) with one interior-point solve per knot (two launches: the solves,
 * then all implicit-function gradients; the handle owns the hand-over workspace).  dx: 2nq x 2nq, du: 2nq x nu per
 * problem, every entry written (the reference writes 3 blocks into a caller-zeroed matrix).
 * Any of d, dx, du may be NULL.  status: OD_STATUS_EVAL_OK = state converged at kappa_eval, OD_STATUS_GRAD_OK = gradient
 * iterate converged at kappa_grad, OD_STATUS_FACTOR_OK = no pivot vanished.  With a finite undercut (and cones, and
 * kappa_eval != kappa_grad) the two simulators of the reference iterate differently: the library then runs two
 * solves per knot like the reference and merges their status bits and iteration counts. */
int od_step_grad(od_handle h, long B, const void* x, const void* u, void* d, void* dx, void* du,
                 int* status, int* iters);

/* same solve, compact outputs: q3 (nq) and dq3 = d q3 / d(q1, q2, u1): nq x (2nq+nu) per problem
 * (= grad.dq3dq1[1], dq3dq2[1], dq3du1[1] of the reference's grad_sim, src/dynamics.jl:110-111,125) */
int od_step_grad_compact(od_handle h, long B, const void* x, const void* u, void* q3, void* dq3,
                         int* status, int* iters);

/* iLQR.rollout through f plus the derivative sweep through fx, fu (examples/acrobot.jl:92 and the
 * solver's per-knot fx/fu calls) for B trajectories of T steps: launch 1 runs the time recursion on the
 * device, launch 2 all T*B implicit gradients in parallel.
 * x1: 2nq per trajectory; U: nu per knot (T*B knots); X: 2nq per slot ((T+1)*B slots, slot 0 = x1);
 * A: 2nq x 2nq per knot; Bm: 2nq x nu per knot.  A, Bm, status (T*B), iters (2*T*B) may be NULL.
 * With a finite undercut (and cones, and kappa_eval != kappa_grad) a third launch solves every knot again at
 * kappa_grad from its rolled-out state, as the reference's fx / fu do (status / iterations merged as in od_step_grad). */
int od_rollout(od_handle h, long B, int T, const void* x1, const void* U, void* X, void* A, void* Bm,
               int* status, int* iters);
/* the same two launches with the linearisation in compact form: dq3 = d q3 / d(q1, q2, u1), nq x (2nq+nu) column-major
 * per knot -- the only non-constant block of fx / fu (src/dynamics.jl:105-111,125): half the bytes of A and B, and what
 * a multi-GPU gather of the linearisation should ship. */
int od_rollout_compact(od_handle h, long B, int T, const void* x1, const void* U, void* X, void* dq3,
                       int* status, int* iters);

/* ---- multi-GPU: one host process per GPU, rollouts sharded, no collective on the data path.  The one exchange the
 * path can have -- an all-gather of the per-knot linearisation for an outer loop whose backward pass runs on every rank (or on the host),
 * the consumer being IterativeLQR's Riccati pass (examples/acrobot.jl:97-113) -- sits behind the boundary too, over RCCL on the handle's
 * stream, so that a Julia process per GPU needs nothing but `ccall`.  The reference has no distributed code: these
 * entry points replace nothing, they are what its single process turns into on eight devices.
 *
 * od_comm_unique_id: rank 0 makes the 128-byte id (ncclGetUniqueId) and hands it to the other ranks out of band (a file, a socket,
 * Julia's Distributed).  od_comm_create: every rank, same id, its rank and the world size -- a collective call (ncclCommInitRank) on
 * the handle's device; the communicator belongs to that device.  od_comm_info: size and rank as the communicator reports them
 * (ncclCommCount / ncclCommUserRank) and its device.  OD_ERR_UNSUPPORTED where librccl cannot be loaded. */
#define OD_COMM_ID_BYTES 128
typedef struct od_comm_s* od_comm;
int od_comm_unique_id(void* id /* OD_COMM_ID_BYTES, host */);
int od_comm_create(od_handle h, const void* id, int rank, int world, od_comm* out);
int od_comm_info(od_comm c, int* world, int* rank, int* device);
int od_comm_destroy(od_comm c);
/* all-gather of od_rollout_compact's outputs of B trajectories x T steps per rank (every rank the same B and T: pad the last shard):
 * X (2nq per slot, (T+1)*B slots) -> X_all, dq3 (nq x (2nq+nu) per knot, T*B knots) -> dq3_all; either pair may be NULL.  Block r of an
 * output is rank r's array as it stands (its layout, its B): X_all = world blocks of 2nq*(T+1)*B doubles, dq3_all = world blocks of
 * nq*(2nq+nu)*T*B.  Device pointers; asynchronous on the handle's stream, after the rollout that was queued there. */
int od_allgather_compact(od_handle h, od_comm c, long B, int T, const void* X, const void* dq3, void* X_all, void* dq3_all);
/* the same for any device buffer of `bytes` bytes per rank (recv: world * bytes) -- e.g. the gains of a backward pass that ran on one rank */
int od_comm_allgather(od_handle h, od_comm c, const void* send, void* recv, size_t bytes);

/* ---- next row of the scope table: the iLQR iteration around the path -----------------
 * Forward pass / Armijo line search of IterativeLQR (iLQR.solve!, examples/acrobot.jl:113): closed-loop
 * rollouts u_t = ubar_t + alpha k_t + K_t (x_t - xbar_t) of B nominal trajectories for nalpha step sizes
 * at once (candidate p = a*B + b).  alphas: nalpha doubles (device); x1: 2nq per nominal trajectory;
 * xbar: 2nq per slot ((T+1)*B); ubar, kff: nu per nominal knot; K: nu x 2nq col-major per nominal knot;
 * X: 2nq per slot ((T+1)*B*nalpha), U: nu per candidate knot (T*B*nalpha).  State only (no gradients). */
int od_rollout_policy(od_handle h, long B, int T, int nalpha, const void* alphas, const void* x1, const void* xbar,
                      const void* ubar, const void* K, const void* kff, void* X, void* U, int* status, int* iters);

/* Riccati backward pass (Gauss-Newton iLQR) for B trajectories with a per-knot quadratic cost model:
 * A (n x n), Bm (n x m) from od_rollout; lxx (n x n), luu (m x m), lux (m x n), lx (n), lu (m) per knot;
 * VxxT (n x n), VxT (n) per trajectory; reg added to the diagonal of Quu.  Outputs K (m x n), k (m) per knot,
 * dV = [sum k'Qu, sum 1/2 k'Quu k] per trajectory, status 1 = all Quu positive definite.  n <= 16, m <= 12.
 * The kernels (od_set_cooperative) agree to rounding, not bit for bit; each gives a trajectory the same bits in any batch. */
int od_ilqr_backward(od_handle h, long B, int T, int n, int m, const void* A, const void* Bm, const void* lxx,
                     const void* luu, const void* lux, const void* lx, const void* lu, const void* VxxT,
                     const void* VxT, double reg, void* K, void* k, void* dV, int* status);

/* Cost of P trajectories under a quadratic objective (the stage / terminal costs of the reference's examples, e.g.
 * examples/hopper.jl:207-220; the Armijo test of the forward pass evaluates it on every candidate):
 *   J_p = sum_{t<T} 1/2 (x_t - xref)'Q(x_t - xref) + 1/2 u_t'R u_t  +  1/2 (x_T - xref)'QT(x_T - xref).
 * X: n per slot ((T+1)*P), U: m per knot (T*P), in the handle's layout and of element type `dtype` (OD_F64 / OD_F32: the
 * states a single-precision rocket handle rolled out are read as they are); Q, QT (n x n), R (m x m) col-major, xref (n) and J (P)
 * are doubles on the device.  n <= 16, m <= 12.  One pass over X and U. */
int od_quad_cost(od_handle h, long P, int T, int n, int m, int dtype, const void* X, const void* U, const double* Q,
                 const double* R, const double* QT, const double* xref, double* J);

/* ---- the whole iLQR iteration on the device -----------------------------------------------------------
 * iLQR.solver / iLQR.solve! of IterativeLQR.jl as the reference drives it (examples/acrobot.jl:97-113, examples/rocket.jl:118-139)
 * for B independent problems that share their launches: quadratic stage / terminal objective (od_quad_cost); terminal goals
 * x_T[idx] = goal (od_ilqr_set_objective) and affine stage / terminal equality and inequality rows (od_ilqr_set_constraints) by
 * augmented Lagrangian.  Every problem has its own regularisation schedule, penalty, convergence flag and constraint flag on the
 * device and follows the path it would follow in a batch of one.  One iteration =
 *   Riccati backward pass (a trajectory whose Quu + reg I is not positive definite repeats its own recursion at 10x the
 *   regularisation, up to 1e6) | closed-loop rollouts of ALL step sizes of all trajectories (the rocket sums their costs on the
 *   way) | cost of every candidate, its constraint terms | Armijo selection per trajectory | copy of the accepted candidate and
 *   expansion of the merit on its slots | linearisation (fx, fu) on the states of the trajectories that moved | per-trajectory
 *   bookkeeping (regularisation x10 / :5, converged if dJ < obj_tol or reg >= 1e6, cost history)
 * -- six to eight launches on the handle's stream, NO host synchronisation, no allocation: od_ilqr_iterate can be recorded in a
 * HIP graph.  A converged trajectory drops out of every kernel (rollouts, cost, linearisation -- for the mechanical models the
 * two passes of od_step_grad are predicated per trajectory like the rocket's kernels); once all have, a flag on the device turns
 * later iterations into empty launches, so a caller may enqueue max_iter iterations blindly.  The handle may be a mechanical model (n = 2nq, m = nu:
 * od_rollout_policy / od_step_grad underneath) or OD_ROCKET_DYNAMICS in OD_F64 / OD_F32 (n = 12, m = 3: od_rocket_rollout /
 * od_rocket with or without the thrust-cone projection).  Batch-minor layout only.  The solver borrows the handle: no other
 * call on the handle while solver work is in flight, and the handle must outlive the solver. */
typedef struct od_ilqr_s* od_ilqr;
typedef struct {
  double reg;          /* initial / minimal regularisation of Quu (1e-6)                                   */
  double c1;           /* Armijo constant (1e-4)                                                           */
  double obj_tol;      /* inner loop ends when an iteration improved no cost by more than this (1e-6)      */
  double con_tol;      /* outer loop ends when max |x_T[idx] - goal| over the batch is below this (1e-3)   */
  double rho_init, rho_scale;   /* penalty schedule (1, 10); cf. iLQR.Options, examples/acrobot.jl:98-107  */
  int max_iter, max_al_iter;    /* (50, 1)                                                                 */
  int project;         /* rocket handles: 1 = f_rocket_proj (thrust-cone projection on the path), 0 = f_rocket */
  int history;         /* cost histories kept on the device: iterations (0 = max_iter * max_al_iter)      */
  int proj_stall_exit; /* rocket handles: the solver's own launches abandon stalled thrust-cone projections (1; see
                          od_set_projection_stall_exit -- the handle's setting is not used by the solver)  */
  double rho_max;      /* the penalty never grows beyond this (1e8; IterativeLQR's maximum penalty as recalled) */
} od_ilqr_options;
typedef struct {
  int iterations;      /* iLQR iterations run since od_ilqr_init (all augmented-Lagrangian rounds)         */
  int al_iterations;   /* multiplier updates done                                                          */
  int done;            /* inner loop converged (or regularisation exhausted)                               */
  int al_done;         /* terminal constraints met to con_tol                                              */
  int bad_linearisations;  /* knots of the current linearisation whose dynamics solve did not converge    */
  double reg, rho, max_dJ, max_violation;
} od_ilqr_info;
int od_ilqr_default_options(od_ilqr_options* out);
/* alphas: nalpha step sizes tried in this order (host pointer; e.g. 1, 1/2, ..., 2^-10).  Allocates every buffer. */
int od_ilqr_create(od_handle h, long B, int T, int nalpha, const double* alphas, const od_ilqr_options* opts, od_ilqr* out);
int od_ilqr_destroy(od_ilqr s);
/* J = sum_t 1/2 (x_t - xref)'Q(x_t - xref) + 1/2 u_t'R u_t + 1/2 (x_T - xref)'QT(x_T - xref); Q, QT n x n and R m x m
 * column-major, symmetric; ngoal terminal equality constraints x_T[goal_idx[i]] = goal[i] (ngoal = 0: none).  Host pointers. */
int od_ilqr_set_objective(od_ilqr s, const double* Q, const double* R, const double* QT, const double* xref, int ngoal,
                          const int* goal_idx, const double* goal);
/* Affine constraints handled by the augmented Lagrangian (iLQR.Constraint with idx_ineq, examples/rocket.jl:82-110,
 * examples/planar_push.jl:105-106): stage rows  Cs x_t + Ds u_t - ds  for t < T (Cs ns x n, Ds ns x m col-major) and terminal rows
 * Ct x_T - dt (Ct nt x n); the first ns_ineq / nt_ineq rows are inequalities (<= 0), the others equalities; ns, nt <= 16.
 * Merit lam'c + rho/2 c'Ac with the active set A = equalities + {c_i >= 0 or lam_i > 0}; lam <- max(0, lam + rho c) on
 * inequalities; violation = max(|c_eq|, max(c_ineq, 0)) against con_tol.  In addition to the goal rows of od_ilqr_set_objective.
 * Host pointers; call before od_ilqr_init.  (0, 0, NULL, ...) removes them. */
int od_ilqr_set_constraints(od_ilqr s, int ns, int ns_ineq, const double* Cs, const double* Ds, const double* ds,
                            int nt, int nt_ineq, const double* Ct, const double* dt);
/* ---- parameter stage: a first stage of its own dimensions, nonlinear constraints on it (examples/hopper.jl) ----------------------
 * examples/hopper.jl:16-50,52-175,234-266 optimises, with the controls, the two initial configurations theta = [q1; q2] of the
 * hopper's gait: its first stage takes theta as ADDITIONAL CONTROLS (x_1 in R^8 fixed, u_1 = [u; theta] in R^10, f1: [q2; q3; theta] in
 * R^16), the later stages carry theta in their state (R^16, ft), the first stage has a cost and nonlinear constraints on theta
 * (stage1_con: the foot does not move between the two configurations) and the terminal constraint couples x_T with theta
 * (terminal_con: half a metre further, otherwise the pose the gait started from).  od_ilqr_set_parameter_stage poses that problem
 * for a mechanical handle (2 n <= 16, m + n <= 12): SLOT 0 of every trajectory is theta -- x1 of od_ilqr_init / od_ilqr_solve is its
 * initial value, X[:, 0] of od_ilqr_get the optimised one --; the objective of od_ilqr_set_objective applies to slots 1 .. T (Q, QT)
 * and to every control (R); slot 0 costs 1/2 theta' diag(w_theta) theta + cost_const instead; `constraint` names a generated
 * function c(theta; p) = 0 (python -m optimization_dynamics_amd.codegen --add-constraint spec.py, the counterpart of
 * iLQR.Constraint(f, nx, nu) differentiating a user's function; od_constraint_id("hopper_foot") is the one of the example); the
 * terminal rows Ct_x x_T + Ct_theta theta - dt (nt x n col-major each, the first nt_ineq inequalities <= 0) join the augmented
 * Lagrangian like the rows of od_ilqr_set_constraints, whose stage rows keep applying to (slot state, control) at every knot.  The
 * iteration is the one described above -- Riccati pass on the model of the reference's formulation at its dimensions (2 n states,
 * m + n controls), candidates rolled out from their own theta + alpha dtheta by the kernels of the path -- with no host
 * synchronisation.  Host pointers; call before od_ilqr_init; NULL removes the stage.
 * NOT combinable with goal components (od_ilqr_set_objective, ngoal > 0) or terminal rows of od_ilqr_set_constraints (nt > 0): the
 * stage's terminal model consists of QT and its own coupled rows, so either call answers OD_ERR_UNSUPPORTED while the other kind is
 * in force (state such conditions as coupled rows with Ct_theta = 0); at most 64 constraint parameters. */
typedef struct {
  int constraint;            /* id of a generated constraint function on theta (od_constraint_id), -1: none; all rows equalities */
  int n_p; const double* p;  /* its parameters (od_constraint_dims) */
  const double* w_theta;     /* n weights of the cost 1/2 theta' diag(w) theta (obj1's weights on u[3:10], examples/hopper.jl:209) */
  double cost_const;         /* added to every cost (the first stage's cost of the fixed x_1) */
  int nt, nt_ineq;           /* terminal rows coupling x_T and theta */
  const double *Ct_x, *Ct_theta, *dt;
} od_ilqr_parameter_stage;
int od_ilqr_set_parameter_stage(od_ilqr s, const od_ilqr_parameter_stage* ps);
/* GradientBundle as the solver's linearisation -- examples/planar_push.jl:15,22,29-30 with GB = true: fx_gb / fu_gb
 * (src/gradient_bundle.jl:109-147) instead of the implicit gradients.  Every linearisation of the nominal trajectory then runs
 * gradient! on its T*B knots (N + 1 steps of the eval simulator per knot with the samples eta, the least-squares fit of src/ls.jl:
 * the kernels of od_bundle_grad) and fills fx = [0 I; dz_q1 dz_q2], fu = [0; dz_u].  eta: (2 nq + nu) x N col-major on the HOST
 * (gb.ls.eta: one non-zero per column, eps * randn at a random coordinate, src/gradient_bundle.jl:49-54); mechanical models, also
 * with a parameter stage.  N = 0 or eta = NULL: implicit gradients again.  Call before od_ilqr_init. */
int od_ilqr_set_gradient_bundle(od_ilqr s, int N, const double* eta);
/* registry of the generated constraint functions (csrc/gen/con_list.h): rows nc on nx variables with np parameters */
int od_num_constraints(void);
int od_constraint_id(const char* name);       /* -1 if unknown */
const char* od_constraint_name(int id);
int od_constraint_dims(int id, int* nc, int* nx, int* np);
/* initialize_controls! + rollout + first linearisation and cost (examples/acrobot.jl:108-113): x1 n per trajectory, U0 m per knot
 * (T*B knots), doubles on the device.  Resets multipliers, penalty, regularisation and counters.  Asynchronous. */
int od_ilqr_init(od_ilqr s, const double* x1, const double* U0);
/* niter iterations, asynchronous, capturable */
int od_ilqr_iterate(od_ilqr s, int niter);
/* augmented-Lagrangian round: violation check, lam += rho c, rho *= rho_scale, cost of the nominal trajectory under the new
 * multipliers, regularisation reset, convergence flag cleared.  Asynchronous, capturable. */
int od_ilqr_al_update(od_ilqr s);
/* solve!(solver): init, then max_al_iter rounds of max_iter iterations with the multiplier update between them.  Iterations are
 * enqueued in chunks; the host waits only for the chunk before the previous one (the queue never drains) to stop early. */
int od_ilqr_solve(od_ilqr s, const double* x1, const double* U0);
/* results (device pointers, doubles, any may be NULL): X n per slot ((T+1)*B), U m per knot, J per trajectory (with the
 * multiplier terms), K (m x n col-major per knot) and k of the last backward pass.  With a parameter stage slot 0 of X is the
 * optimised theta, K is the feedback on the mechanical state (zero at knot 0, where the state is the stage's control) and k the
 * feed-forward of the m mechanical controls.  Asynchronous. */
int od_ilqr_get(od_ilqr s, double* X, double* U, double* J, double* K, double* k);
/* hist: up to `cap` rows of B costs (row i = costs after iteration i), device pointer; returns the number of rows kept so far
 * (synchronises) or a negative error */
int od_ilqr_get_history(od_ilqr s, double* hist, int cap);
/* the decisions behind that history, row i = iteration i, B entries each (device pointers, any may be NULL; same row count and
 * return value as od_ilqr_get_history): step_index = index into `alphas` of the step size the Armijo test accepted (-1: none, the
 * regularisation went up instead; -2: the trajectory had converged before this iteration), reg = the trajectory's regularisation
 * after the iteration, rho = its penalty.  What IterativeLQR prints per iteration with verbose = true (alpha, cost, penalty). */
int od_ilqr_get_trace(od_ilqr s, int* step_index, double* reg, double* rho, int cap);
/* per trajectory (device pointers, any may be NULL; asynchronous): flags (bit 0: inner loop converged, bit 1: constraints met to
 * con_tol), violation as of the last od_ilqr_al_update / od_ilqr_solve, penalty rho.  The B problems are independent solves: each
 * has its own regularisation schedule, penalty and flags, and follows the path it would follow in a batch of one. */
int od_ilqr_get_status(od_ilqr s, int* flags, double* violation, double* penalty);
int od_ilqr_get_info(od_ilqr s, od_ilqr_info* out);   /* synchronises the handle's stream; done / al_done: every trajectory's flag; reg, rho: the largest */

/* gradient! (src/gradient_bundle.jl:87-104) for B knots: N+1 eval-simulator steps per knot with the
 * caller's perturbations eta ((2nq+nu) x N col-major, shared by all knots; the reference draws them
 * in the GradientBundle constructor :49-54) followed by the least-squares fit of src/ls.jl:44-60.
 * dz: nq x (2nq+nu) per knot.  workspace: device scratch of od_bundle_workspace_bytes().
 * status (per knot): 1 = the Gram matrix of the perturbations is non-singular and the fit is finite (a sample whose
 * solve failed with a non-finite state makes it 0; non-converged but finite samples enter the fit as in the reference). */
size_t od_bundle_workspace_bytes(od_handle h, long B, int N);
int od_bundle_grad(od_handle h, long B, int N, const void* x, const void* u, const void* eta,
                   void* dz, void* workspace, size_t workspace_bytes, int* status);

/* LeastSquares update! (src/ls.jl:44-60) alone, for B independent fits: minimise
 * sum_i |f_eta_i - f_z - M eta_i|^2 over M (ny x nzb).  feta: ny per sample, (N+1)*B samples in
 * BATCH_MINOR order, sample index b*(N+1)+i with i = 0 the unperturbed f_z; eta: nzb x N col-major;
 * M: ny x nzb per fit (handle layout).  ny, nzb <= 24.  status as for od_bundle_grad. */
int od_ls_fit(od_handle h, long B, int N, int ny, int nzb, const void* eta, const void* feta, void* M,
              int* status);

/* interior_point_solve!(ip) on caller-provided z0, theta (src/models/rocket/dynamics.jl:109,178):
 * z: nz per problem; dz: nzq x ngc per problem (rows = solution block, cols = leading theta
 * columns, see od_raw_grad_dims); dz may be NULL (diff_sol = false).  This is the generic loop as the literal restatement has it, for
 * any model of the table -- for OD_ROCKET_PROJECTION too: the generated elimination with its pivoted tail, the literal line search;
 * od_soc_project / od_rocket are the entry points that complete that solve's eps_min = 0 steps as exact arithmetic has them. */
int od_raw_grad_dims(int model, int* nzq, int* ngc);

/* The whole solution of a step: z (nz per problem) at kappa_eval and dz = d z / d(q1, q2, u1)
 * (nz x (2nq + nu) col-major per problem; NULL = no differentiation) at kappa_grad -- what RoboDojo's `process!`
 * fills sim.traj.gamma / sim.traj.b and sim.grad.dgamma1d* / db1d* from (sized in src/dynamics.jl:36-46).
 * od_model_indices gives the rows: which = OD_IDX_CONFIGURATION (next configuration q3), OD_IDX_GAMMA (impact
 * impulses, nc of them), OD_IDX_B (friction impulses, nb); returns the count, writes at most cap indices. */
enum od_index_set { OD_IDX_CONFIGURATION = 0, OD_IDX_GAMMA = 1, OD_IDX_B = 2 };
int od_model_indices(int model, int which, int* idx, int cap);
int od_step_full(od_handle h, long B, const void* x, const void* u, void* z, void* dz, int* status, int* iters);
int od_ip_solve(od_handle h, long B, const void* z0, const void* theta, void* z, void* dz,
                int* status, int* iters);

/* f_rocket / fx_rocket / fu_rocket (project = 0, src/models/rocket/dynamics.jl:101-164) and
 * f_rocket_proj / fx_rocket_proj / fu_rocket_proj (project = 1, :215-268) on an
 * OD_ROCKET_DYNAMICS handle.  x: 12, u: 3, y: 12, dx: 12x12, du: 12x3, uproj: 3 per problem.
 * status bits 1,2,4 = dynamics solve; 16, 32 = projection solve state / gradient converged. */
int od_rocket(od_handle h, long B, int project, const void* x, const void* u, void* y, void* dx,
              void* du, void* uproj, int* status);

/* soc_projection / soc_projection_gradient (src/models/rocket/dynamics.jl:168-214) on an OD_ROCKET_DYNAMICS
 * handle: Euclidean projection of u (3 per problem) onto the thrust cone {|u_1:2| <= u_3 <= u_max} by the
 * interior-point solve the reference uses (z0 and options of :169-175).  uproj: 3; duproj: 3 x 3 col-major
 * d uproj / d u (NULL = diff_sol false); status bits 16 / 32 = state / gradient converged.
 * The reference runs this solve with eps_min = 0: every step goes all the way to the boundary of an orthant, and a literal
 * floating-point transcription of the loop then reads rounding noise in three places (where the blocking variable lands, the sign of
 * the next affine direction of a variable at zero, the line search's r_c <= r_vio between two residuals that are exactly zero).  The
 * solve here completes each as EXACT arithmetic has it -- blocking variable exactly zero, that variable's direction from its own
 * complementarity row, first trial accepted on the linear equality rows -- and follows the exact-arithmetic path (binary128) to 1e-7
 * on 99.95 % of controls; a literal double-precision loop stays on it on ~95 %. */
int od_soc_project(od_handle h, long B, const void* u, void* uproj, void* duproj, int* status);
/* the same solve with its whole solution: z (10 per problem) = [u_proj (3), p, s, w, y, v (3)] of the projection's KKT system
 * (src/models/rocket/codegen.jl:45-64) -- the iterate soc_projection_gradient differentiates at (ip.z, dynamics.jl:180-185; the
 * two tolerances of this solve are equal) -- and iters (1 per problem, may be NULL) the interior-point iterations it took.  Lets a
 * checker recompute -rz^{-1} rtheta at exactly that point. */
int od_soc_project_full(od_handle h, long B, const void* u, void* z, void* duproj, int* status, int* iters);

/* iLQR.rollout / forward pass over f_rocket (project = 0) or f_rocket_proj (project = 1), time recursion on the
 * device (examples/rocket.jl:29-41,118).  nalpha = 0: open loop, controls ubar (3 per knot, T*B knots), B
 * trajectories.  nalpha > 0: closed loop u = ubar + alpha k + K (x - xbar) for nalpha step sizes (candidate
 * p = a*B + b; xbar 12 per slot ((T+1)*B), K 3 x 12 col-major and kff 3 per nominal knot, alphas on the device).
 * X: 12 per slot ((T+1)*P slots, P = B or B*nalpha), U (optional): controls applied before projection, 3 per
 * candidate knot; status per candidate knot. */
int od_rocket_rollout(od_handle h, long B, int T, int nalpha, const void* alphas, int project, const void* x1,
                      const void* xbar, const void* ubar, const void* K, const void* kff, void* X, void* U, int* status);

/* host-pointer scalar path (B = 1), the plumbing config: reference signatures f(d,model,x,u,w),
 * fx(dx,...), fu(du,...) (src/dynamics.jl:81,96,116).  Column-major, every entry written. */
int od_f_host(od_handle h, const double* x, const double* u, double* d);
int od_fx_host(od_handle h, const double* x, const double* u, double* dx);
int od_fu_host(od_handle h, const double* x, const double* u, double* du);
/* f, fx and fu of one knot from ONE solve (the reference's three callbacks each solve again, src/dynamics.jl:88,103,123);
 * any of d, dx, du may be NULL.  A Julia closure triple can call this once per (x, u) and serve fx / fu from the result. */
int od_ffxfu_host(od_handle h, const double* x, const double* u, double* d, double* dx, double* du);
/* the scalar rocket entry points on host vectors, OD_ROCKET_DYNAMICS handle (src/models/rocket/dynamics.jl):
 * project = 0: f_rocket / fx_rocket / fu_rocket (:101-164); project = 1: f/fx/fu_rocket_proj (:215-268, du includes the
 * chain product with the projection gradient).  y 12, dx 12 x 12, du 12 x 3 col-major, uproj 3; outputs may be NULL. */
int od_rocket_host(od_handle h, int project, const double* x, const double* u, double* y, double* dx, double* du,
                   double* uproj, int* status);
/* soc_projection (duproj = NULL) / soc_projection_gradient (:168-214) for one u (3) -> uproj (3), duproj (3 x 3) */
int od_soc_project_host(od_handle h, const double* u, double* uproj, double* duproj, int* status);
/* gradient!(sim, gb, q1, q2, u1) for one knot on host vectors (src/gradient_bundle.jl:87-104): x = [q1; q2] (2nq),
 * eta (2nq+nu) x N col-major (gb.ls.eta), dz nq x (2nq+nu) col-major (gb.dz) */
int od_bundle_grad_host(od_handle h, int N, const double* x, const double* u, const double* eta, double* dz, int* status);

#ifdef __cplusplus
}
#endif
#endif /* OD_MI355X_H */
