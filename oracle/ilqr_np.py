"""numpy restatement of the Riccati backward pass used by optimization_dynamics_amd.ilqr -- TEST
INFRASTRUCTURE ONLY (checker for od_ilqr_backward).  Gauss-Newton iLQR as in IterativeLQR.jl's backward
pass as recalled (un-vendored, unpinned)."""
import numpy as np


def backward(A, Bm, lxx, luu, lux, lx, lu, Vxx, Vx, reg):
    """one trajectory.  A: (T,n,n) B: (T,n,m) lxx: (T,n,n) luu: (T,m,m) lux: (T,m,n) lx: (T,n) lu: (T,m)"""
    T, n, m = Bm.shape
    K = np.zeros((T, m, n)); k = np.zeros((T, m)); dV = np.zeros(2)
    Vxx = Vxx.copy(); Vx = Vx.copy()
    for t in range(T - 1, -1, -1):
        Qx = lx[t] + A[t].T @ Vx
        Qu = lu[t] + Bm[t].T @ Vx
        Qxx = lxx[t] + A[t].T @ Vxx @ A[t]
        Quu = luu[t] + Bm[t].T @ Vxx @ Bm[t]
        Qux = lux[t] + Bm[t].T @ Vxx @ A[t]
        Qr = Quu + reg * np.eye(m)
        K[t] = -np.linalg.solve(Qr, Qux)
        k[t] = -np.linalg.solve(Qr, Qu)
        dV += [k[t] @ Qu, 0.5 * k[t] @ Quu @ k[t]]
        Vx = Qx + K[t].T @ Quu @ k[t] + K[t].T @ Qu + Qux.T @ k[t]
        Vxx = Qxx + K[t].T @ Quu @ K[t] + K[t].T @ Qux + Qux.T @ K[t]
        Vxx = 0.5 * (Vxx + Vxx.T)
    return K, k, dV


def backward_batch(A, Bm, lxx, luu, lux, lx, lu, Vxx, Vx, reg):
    """the same recursion for P trajectories at once (vectorised over the leading axis; numpy's batched solve).
    A: (P,T,n,n) B: (P,T,n,m) lxx: (P,T,n,n) luu: (P,T,m,m) lux: (P,T,m,n) lx: (P,T,n) lu: (P,T,m) Vxx: (P,n,n) Vx: (P,n);
    reg: scalar or (P,).  -> K (P,T,m,n), k (P,T,m), dV (P,2)"""
    P, T, n, m = Bm.shape
    K = np.zeros((P, T, m, n)); k = np.zeros((P, T, m)); dV = np.zeros((P, 2))
    Vxx = Vxx.copy(); Vx = Vx.copy()
    reg = np.broadcast_to(np.asarray(reg, dtype=np.float64), (P,))
    I = np.eye(m)[None] * reg[:, None, None]
    tr = lambda M: np.swapaxes(M, -1, -2)
    for t in range(T - 1, -1, -1):
        At, Bt = A[:, t], Bm[:, t]
        Qx = lx[:, t] + np.einsum("pji,pj->pi", At, Vx)
        Qu = lu[:, t] + np.einsum("pji,pj->pi", Bt, Vx)
        VA, VB = Vxx @ At, Vxx @ Bt
        Qxx = lxx[:, t] + tr(At) @ VA
        Quu = luu[:, t] + tr(Bt) @ VB
        Qux = lux[:, t] + tr(Bt) @ VA
        Qr = Quu + I
        K[:, t] = -np.linalg.solve(Qr, Qux)
        k[:, t] = -np.linalg.solve(Qr, Qu[:, :, None])[:, :, 0]
        dV[:, 0] += np.einsum("pi,pi->p", k[:, t], Qu)
        dV[:, 1] += 0.5 * np.einsum("pi,pij,pj->p", k[:, t], Quu, k[:, t])
        Kt = K[:, t]
        Vx = Qx + np.einsum("pji,pjk,pk->pi", Kt, Quu, k[:, t]) + np.einsum("pji,pj->pi", Kt, Qu) + np.einsum("pji,pj->pi", Qux, k[:, t])
        Vxx = Qxx + tr(Kt) @ Quu @ Kt + tr(Kt) @ Qux + tr(Qux) @ Kt
        Vxx = 0.5 * (Vxx + tr(Vxx))
    return K, k, dV


# ------------------------------------------------------------------------------------------------------------------------------------
# The whole augmented-Lagrangian iLQR loop in numpy, one problem at a time, driven by the ORACLE's dynamics -- the independent checker of
# od_ilqr_* (include/od_mi355x.h; DESIGN.md section 7 states the rules; IterativeLQR.jl's solve! as the reference drives it,
# examples/acrobot.jl:97-113, examples/rocket.jl:118-139, internals recalled -- SURVEY.md Appendix A).  It shares no code with the
# product: objective, constraint rows, active sets, Riccati recursion, Armijo rule, regularisation schedule and multiplier update are
# restated here from the rules, and every decision is logged per iteration so that a test can compare them one by one.
#
#   merit        J = sum_t 1/2 (x_t - xref)'Q(x_t - xref) + 1/2 u_t'R u_t + 1/2 (x_T - xref)'QT(x_T - xref)
#                    + goal rows  lam'c + rho/2 c'c,  c = x_T[idx] - goal
#                    + stage rows (t < T) and terminal rows  lam_i c_i + [i active] rho/2 c_i^2,  c = C x + D u - d; the leading n_ineq
#                      rows are inequalities (<= 0), active iff c_i >= 0 or lam_i > 0; equalities are always active
#   iteration    expansion of the merit | Riccati pass with Quu + reg I; not positive definite: the pass is repeated at
#                max(reg, 1e-8) x 10, x 100, ... up to 1e6, else no step | candidates u = ubar + alpha k + K (x - xbar) for the step sizes in
#                order: the first whose rollout converged everywhere, whose cost is finite and <= J + c1 (alpha dV1 + alpha^2 dV2),
#                dV1 = sum k'Qu, dV2 = sum 1/2 k'Quu k | step taken: reg <- max(reg / 5, reg0), converged if J_old - J_new < obj_tol;
#                no step: reg <- min(10 reg, 1e6), converged (given up) at 1e6
#   round        violation = max(|c_eq|, max(c_ineq, 0), |goal rows|) < con_tol: done; else lam <- lam + rho c (inequalities: max(0, .)),
#                rho <- min(rho rho_scale, rho_max), reg <- reg0, merit re-evaluated under the new multipliers
class Problem:
    """a quadratic objective with optional goal rows / affine stage and terminal rows (the arguments of od_ilqr_set_objective and
    od_ilqr_set_constraints), and the dynamics as two callables:
       step(x, u) -> (converged, x_next)                       one knot (the oracle's f)
       linearise(X (T, n), U (T, m)) -> A (T, n, n), B (T, n, m)   the oracle's fx, fu on the knots of a trajectory"""

    def __init__(self, step, linearise, Q, R, QT, xref, goal_idx=None, goal=None, stage=None, terminal=None):
        self.step, self.linearise = step, linearise
        self.Q, self.R, self.QT, self.xref = [np.asarray(a, dtype=np.float64) for a in (Q, R, QT, xref)]
        self.n, self.m = self.Q.shape[0], self.R.shape[0]
        self.goal_idx = None if goal_idx is None else np.asarray(goal_idx, dtype=int)
        self.goal = None if goal is None else np.asarray(goal, dtype=np.float64)
        self.stage = None if stage is None else (np.asarray(stage[0], float).reshape(-1, self.n), np.asarray(stage[1], float).reshape(-1, self.m),
                                                 np.asarray(stage[2], float).reshape(-1), int(stage[3]))
        self.terminal = None if terminal is None else (np.asarray(terminal[0], float).reshape(-1, self.n), np.asarray(terminal[1], float).reshape(-1), int(terminal[2]))

    @property
    def constrained(self):
        return self.goal_idx is not None or self.stage is not None or self.terminal is not None


def _active(c, lam, n_ineq):
    a = np.ones(c.shape, dtype=bool)
    a[..., :n_ineq] = (c[..., :n_ineq] >= 0.0) | (lam[..., :n_ineq] > 0.0)
    return a


class _Mult:
    def __init__(self, p, T):
        self.goal = None if p.goal_idx is None else np.zeros(p.goal_idx.size)
        self.stage = None if p.stage is None else np.zeros((T, p.stage[2].size))
        self.term = None if p.terminal is None else np.zeros(p.terminal[1].size)


def _rows(p, X, U):
    """constraint values: goal rows (ng,), stage rows (T, ns), terminal rows (nt,) (None where absent)"""
    cg = None if p.goal_idx is None else X[-1, p.goal_idx] - p.goal
    cs = None if p.stage is None else X[:-1] @ p.stage[0].T + U @ p.stage[1].T - p.stage[2]
    ct = None if p.terminal is None else p.terminal[0] @ X[-1] - p.terminal[1]
    return cg, cs, ct


def merit(p, X, U, lam, rho):
    dx = X - p.xref
    J = 0.5 * np.einsum("ti,ij,tj->", dx[:-1], p.Q, dx[:-1]) + 0.5 * np.einsum("ti,ij,tj->", U, p.R, U) + 0.5 * dx[-1] @ p.QT @ dx[-1]
    cg, cs, ct = _rows(p, X, U)
    if cg is not None:
        J += lam.goal @ cg + 0.5 * rho * cg @ cg
    if cs is not None:
        a = _active(cs, lam.stage, p.stage[3])
        J += (lam.stage * cs).sum() + 0.5 * rho * (cs[a] ** 2).sum()
    if ct is not None:
        a = _active(ct, lam.term, p.terminal[2])
        J += lam.term @ ct + 0.5 * rho * (ct[a] ** 2).sum()
    return float(J)


def violation(p, X, U):
    cg, cs, ct = _rows(p, X, U)
    v = 0.0
    if cg is not None:
        v = max(v, np.abs(cg).max())
    if cs is not None:
        k = p.stage[3]
        v = max(v, np.maximum(cs[:, :k], 0.0).max(initial=0.0), np.abs(cs[:, k:]).max(initial=0.0))
    if ct is not None:
        k = p.terminal[2]
        v = max(v, np.maximum(ct[:k], 0.0).max(initial=0.0), np.abs(ct[k:]).max(initial=0.0))
    return float(v)


def expansion(p, X, U, lam, rho):
    """quadratic model of the merit along the trajectory: lxx (T,n,n), luu (T,m,m), lux (T,m,n), lx (T,n), lu (T,m), Vxx (n,n), Vx (n)"""
    T, n, m = U.shape[0], p.n, p.m
    dx = X - p.xref
    lx = dx[:-1] @ p.Q.T
    lu = U @ p.R.T
    lxx = np.repeat(p.Q[None], T, 0); luu = np.repeat(p.R[None], T, 0); lux = np.zeros((T, m, n))
    Vx = p.QT @ dx[-1]; Vxx = p.QT.copy()
    cg, cs, ct = _rows(p, X, U)
    if cs is not None:
        C, D, d, k = p.stage
        ra = np.where(_active(cs, lam.stage, k), rho, 0.0)            # (T, ns)
        w = lam.stage + ra * cs
        lx = lx + w @ C; lu = lu + w @ D
        lxx = lxx + np.einsum("tr,ri,rj->tij", ra, C, C)
        luu = luu + np.einsum("tr,ri,rj->tij", ra, D, D)
        lux = lux + np.einsum("tr,ri,rj->tij", ra, D, C)
    if cg is not None:
        Vx[p.goal_idx] += lam.goal + rho * cg
        Vxx[p.goal_idx, p.goal_idx] += rho
    if ct is not None:
        C, d, k = p.terminal
        ra = np.where(_active(ct, lam.term, k), rho, 0.0)
        Vx = Vx + C.T @ (lam.term + ra * ct)
        Vxx = Vxx + np.einsum("r,ri,rj->ij", ra, C, C)
    return lxx, luu, lux, lx, lu, Vxx, Vx


def _backward_pd(A, Bm, quad, reg):
    """backward() with the positive-definiteness test of Quu + reg I at every knot (Cholesky); -> (K, k, dV) or None"""
    lxx, luu, lux, lx, lu, Vxx, Vx = quad
    T, n, m = Bm.shape
    K = np.zeros((T, m, n)); k = np.zeros((T, m)); dV = np.zeros(2)
    Vxx = Vxx.copy(); Vx = Vx.copy()
    for t in range(T - 1, -1, -1):
        Qx = lx[t] + A[t].T @ Vx
        Qu = lu[t] + Bm[t].T @ Vx
        Qxx = lxx[t] + A[t].T @ Vxx @ A[t]
        Quu = luu[t] + Bm[t].T @ Vxx @ Bm[t]
        Qux = lux[t] + Bm[t].T @ Vxx @ A[t]
        Qr = Quu + reg * np.eye(m)
        try:
            if not np.isfinite(Qr).all():
                return None
            np.linalg.cholesky(0.5 * (Qr + Qr.T))
        except np.linalg.LinAlgError:
            return None
        K[t] = -np.linalg.solve(Qr, Qux)
        k[t] = -np.linalg.solve(Qr, Qu)
        dV += [k[t] @ Qu, 0.5 * k[t] @ Quu @ k[t]]
        # (the value function of the regularised gains: V = Q + K'Quu K + K'Qux + Qux'K, the same map the device kernels apply)
        Vx = Qx + K[t].T @ Quu @ k[t] + K[t].T @ Qu + Qux.T @ k[t]
        Vxx = Qxx + K[t].T @ Quu @ K[t] + K[t].T @ Qux + Qux.T @ K[t]
        Vxx = 0.5 * (Vxx + Vxx.T)
    if not (np.isfinite(K).all() and np.isfinite(k).all()):
        return None
    return K, k, dV


def rollout(p, x1, Ubar, policy=None):
    """open loop (policy None) or closed loop, policy = (alpha, Xbar, K, k): -> X (T+1, n), U (T, m), every step converged"""
    T = Ubar.shape[0]
    X = np.zeros((T + 1, p.n)); X[0] = x1
    U = np.zeros((T, p.m))
    ok = True
    for t in range(T):
        U[t] = Ubar[t] if policy is None else Ubar[t] + policy[0] * policy[3][t] + policy[2][t] @ (X[t] - policy[1][t])
        o, X[t + 1] = p.step(X[t], U[t])
        ok = ok and bool(o)
    return X, U, ok


def solve(p, x1, U0, alphas=tuple(2.0 ** -i for i in range(11)), reg0=1e-6, c1=1e-4, obj_tol=1e-6, con_tol=1e-3, rho_init=1.0, rho_scale=10.0,
          rho_max=1e8, max_iter=50, max_al_iter=1, rollout_fn=None):
    """-> dict(X, U, J, violation, al_done, done, log) with log = one dict per iteration: step (index into alphas, -1 none), alpha, reg
    (after the iteration), rho, J (after), dJ, expected, reg_used (the regularisation the accepted Riccati pass ran with)"""
    roll = rollout_fn if rollout_fn is not None else (lambda x1_, Ub, pol=None: rollout(p, x1_, Ub, pol))
    x1 = np.asarray(x1, dtype=np.float64); U0 = np.asarray(U0, dtype=np.float64)
    T = U0.shape[0]
    X, U, _ = roll(x1, U0)
    lam = _Mult(p, T)
    rho = rho_init if p.constrained else 0.0
    reg, done, al_done = reg0, False, False
    A, Bm = p.linearise(X[:-1], U)
    log = []
    viol = 0.0
    for al in range(max_al_iter):
        J = merit(p, X, U, lam, rho)
        for it in range(max_iter):
            if done:
                break
            quad = expansion(p, X, U, lam, rho)
            r, res = reg, None
            while True:
                res = _backward_pd(A, Bm, quad, r)
                if res is not None or r >= 1e6:
                    break
                r = min(max(r, 1e-8) * 10.0, 1e6)
            step, Jn, exp_, cand = -1, J, 0.0, None
            if res is not None:
                K, k, dV = res
                for ia, a in enumerate(alphas):
                    Xc, Uc, okc = roll(x1, U, (a, X, K, k))
                    Jc = merit(p, Xc, Uc, lam, rho)
                    e = a * dV[0] + a * a * dV[1]
                    if okc and np.isfinite(Jc) and Jc <= J + c1 * e:
                        step, Jn, exp_, cand = ia, Jc, e, (Xc, Uc)
                        break
            dJ = J - Jn
            if step >= 0:
                X, U = cand
                J = Jn
                A, Bm = p.linearise(X[:-1], U)
                reg = max(reg / 5.0, reg0)
                done = dJ < obj_tol
            else:
                reg = min(reg * 10.0, 1e6)
                done = reg >= 1e6
            log.append(dict(al=al, step=step, alpha=(alphas[step] if step >= 0 else 0.0), reg=reg, rho=rho, J=J, dJ=dJ, expected=exp_, reg_used=r))
        if not p.constrained:
            break
        viol = violation(p, X, U)
        if viol < con_tol:
            al_done, done = True, True
            break
        if al + 1 == max_al_iter:
            break
        cg, cs, ct = _rows(p, X, U)
        if cg is not None:
            lam.goal = lam.goal + rho * cg
        if ct is not None:
            l = lam.term + rho * ct; kk = p.terminal[2]
            l[:kk] = np.where(l[:kk] > 0.0, l[:kk], 0.0)
            lam.term = l
        if cs is not None:
            l = lam.stage + rho * cs; kk = p.stage[3]
            l[:, :kk] = np.where(l[:, :kk] > 0.0, l[:, :kk], 0.0)
            lam.stage = l
        rho = min(rho * rho_scale, rho_max)
        reg, done = reg0, False
    return dict(X=X, U=U, J=J, violation=viol, al_done=al_done, done=done, log=log, rho=rho, reg=reg)


# ---- the oracle's dynamics as the two callables of a Problem ---------------------------------------------------------------------------
def mechanical_dynamics(sim):
    """f, fx, fu of oracle/ip_oracle.c for a mechanical model (src/dynamics.jl:81-128)"""
    from . import oracle as O

    def step(x, u):
        ok, d, it = O.f(sim, x, u)
        return ok, d

    def linearise(Xk, Uk):
        D, DX, DU, bad = O.step_grad_batch(sim, np.ascontiguousarray(Xk.T), np.ascontiguousarray(Uk.T))
        return np.moveaxis(DX, 2, 0).copy(), np.moveaxis(DU, 2, 0).copy()

    return step, linearise


def bundle_dynamics(sim, eta):
    """f of the oracle with fx_gb / fu_gb (src/gradient_bundle.jl:109-147) as linearisation: gradient! through the oracle's own
    N + 1 steps and least-squares fit (od_oracle_gradient_bundle) with the samples eta ((2nq + nu) x N)"""
    from . import oracle as O
    nq = O.dims(sim.model_id)["nq"]

    def step(x, u):
        ok, d, it = O.f(sim, x, u)
        return ok, d

    def linearise(Xk, Uk):
        T, n, m = Xk.shape[0], Xk.shape[1], Uk.shape[1]
        A = np.zeros((T, n, n)); Bm = np.zeros((T, n, m))
        for t in range(T):
            ok, dz = O.gradient_bundle(sim, eta, Xk[t, :nq], Xk[t, nq:], Uk[t])
            A[t, :nq, nq:] = np.eye(nq)
            A[t, nq:, :] = dz[:, :n]
            Bm[t, nq:, :] = dz[:, n:]
        return A, Bm

    return step, linearise


def rocket_dynamics(h=0.05, u_max=12.5, project=True):
    """f_rocket(_proj), fx / fu_rocket(_proj) of the oracle (src/models/rocket/dynamics.jl:101-268); + a rollout through its batched form"""
    from . import oracle as O

    def step(x, u):
        if project:
            ok, y, dx, du = O.rocket_proj(h, u_max, x, u)
            return ok, y
        ok, y, dz, it = O.rocket(h, x, u, False)
        return ok, y

    def linearise(Xk, Uk):
        T = Xk.shape[0]
        A = np.zeros((T, 12, 12)); Bm = np.zeros((T, 12, 3))
        if project:
            for t in range(T):
                ok, y, A[t], Bm[t] = O.rocket_proj(h, u_max, Xk[t], Uk[t])
        else:
            Y, DZ, st, it = O.rocket_batch(h, Xk.T, Uk.T, True)
            A[:] = np.moveaxis(DZ[:, :12], 2, 0); Bm[:] = np.moveaxis(DZ[:, 12:15], 2, 0)
        return A, Bm

    def roll(x1, Ubar, policy=None):
        pol = None
        if policy is not None:
            a, Xb, K, k = policy
            pol = (a, Xb.T[:, :, None], np.moveaxis(K, 0, 2)[:, :, :, None], k.T[:, :, None])
        Xr, Ua, st = O.rocket_rollout(h, u_max, x1[:, None], Ubar.T[:, :, None], project=project, policy=pol)
        return Xr[:, :, 0].T.copy(), Ua[:, :, 0].T.copy(), bool((st == 0x11).all())

    return step, linearise, roll


# ------------------------------------------------------------------------------------------------------------------------------------
# The same augmented-Lagrangian iLQR for stages of DIFFERENT dimensions, general stage costs and general (nonlinear) constraints -- what
# iLQR.solver(model, obj, cons) of IterativeLQR.jl accepts and examples/hopper.jl uses (a first stage R^8 x R^10 -> R^16, then
# R^16 x R^2 -> R^16; nonlinear foot-position constraints at the first stage, a terminal constraint that couples the last state with
# the initial configurations carried along in it): the independent checker of od_ilqr_* with a parameter stage
# (od_ilqr_set_parameter_stage).  Same rules as `solve` above; the stages are given as callables.
class Stage:
    """one stage t < T:  step(x, u) -> (converged, y);  jac(x, u) -> (A (ny, nx), B (ny, nu));
    cost(x, u) -> (l, lx, lu, lxx, luu, lux);  con(x, u) -> (c, cx, cu) or None, the leading n_ineq rows inequalities (<= 0)"""

    def __init__(self, step, jac, cost, con=None, n_ineq=0):
        self.step, self.jac, self.cost, self.con, self.n_ineq = step, jac, cost, con, n_ineq


def solve_stages(stages, terminal_cost, terminal_con, nt_ineq, x1, U0, alphas=tuple(2.0 ** -i for i in range(11)), reg0=1e-6, c1=1e-4,
                 obj_tol=1e-6, con_tol=1e-3, rho_init=1.0, rho_scale=10.0, rho_max=1e8, max_iter=50, max_al_iter=1):
    """stages: list of T Stage objects; terminal_cost(x) -> (l, lx, lxx); terminal_con(x) -> (c, cx) or None; U0: list of T control
    vectors.  -> dict(X (list), U (list), J, objective, violation, al_done, done, log)"""
    T = len(stages)

    def roll(Ubar, policy=None):
        X, U, ok = [np.asarray(x1, float)], [], True
        for t in range(T):
            u = Ubar[t] if policy is None else Ubar[t] + policy[0] * policy[3][t] + policy[2][t] @ (X[t] - policy[1][t])
            o, y = stages[t].step(X[t], u)
            ok = ok and bool(o)
            U.append(u); X.append(y)
        return X, U, ok

    def cons(X, U):
        """per stage (c, cx, cu) or None; terminal (c, cx) or None"""
        cs = [st.con(X[t], U[t]) if st.con is not None else None for t, st in enumerate(stages)]
        ct = terminal_con(X[T]) if terminal_con is not None else None
        return cs, ct

    constrained = terminal_con is not None or any(st.con is not None for st in stages)
    X, U, _ = roll([np.asarray(u, float) for u in U0])
    cs, ct = cons(X, U)
    lam_s = [None if c is None else np.zeros(c[0].size) for c in cs]
    lam_t = None if ct is None else np.zeros(ct[0].size)
    rho = rho_init if constrained else 0.0

    def merit(X, U, with_mult=True):
        J = sum(stages[t].cost(X[t], U[t])[0] for t in range(T)) + terminal_cost(X[T])[0]
        if not with_mult or not constrained:
            return float(J)
        cs, ct = cons(X, U)
        for t, c in enumerate(cs):
            if c is not None:
                a = _active(c[0], lam_s[t], stages[t].n_ineq)
                J += lam_s[t] @ c[0] + 0.5 * rho * (c[0][a] ** 2).sum()
        if ct is not None:
            a = _active(ct[0], lam_t, nt_ineq)
            J += lam_t @ ct[0] + 0.5 * rho * (ct[0][a] ** 2).sum()
        return float(J)

    def viol(X, U):
        cs, ct = cons(X, U)
        v = 0.0
        for t, c in enumerate(cs):
            if c is not None:
                k = stages[t].n_ineq
                v = max(v, np.maximum(c[0][:k], 0.0).max(initial=0.0), np.abs(c[0][k:]).max(initial=0.0))
        if ct is not None:
            v = max(v, np.maximum(ct[0][:nt_ineq], 0.0).max(initial=0.0), np.abs(ct[0][nt_ineq:]).max(initial=0.0))
        return float(v)

    def backward(AB, reg):
        cs, ct = cons(X, U)
        l, Vx, Vxx = terminal_cost(X[T])
        Vx, Vxx = Vx.copy(), Vxx.copy()
        if ct is not None:
            ra = np.where(_active(ct[0], lam_t, nt_ineq), rho, 0.0)
            Vx = Vx + ct[1].T @ (lam_t + ra * ct[0]); Vxx = Vxx + ct[1].T @ (ra[:, None] * ct[1])
        K, k, dV = [None] * T, [None] * T, np.zeros(2)
        for t in range(T - 1, -1, -1):
            A, Bm = AB[t]
            _, lx, lu, lxx, luu, lux = stages[t].cost(X[t], U[t])
            if cs[t] is not None:
                c, cx, cu = cs[t]
                ra = np.where(_active(c, lam_s[t], stages[t].n_ineq), rho, 0.0)
                w = lam_s[t] + ra * c
                lx = lx + cx.T @ w; lu = lu + cu.T @ w
                lxx = lxx + cx.T @ (ra[:, None] * cx); luu = luu + cu.T @ (ra[:, None] * cu); lux = lux + cu.T @ (ra[:, None] * cx)
            Qx = lx + A.T @ Vx; Qu = lu + Bm.T @ Vx
            Qxx = lxx + A.T @ Vxx @ A; Quu = luu + Bm.T @ Vxx @ Bm; Qux = lux + Bm.T @ Vxx @ A
            Qr = Quu + reg * np.eye(Quu.shape[0])
            try:
                if not np.isfinite(Qr).all():
                    return None
                np.linalg.cholesky(0.5 * (Qr + Qr.T))
            except np.linalg.LinAlgError:
                return None
            K[t] = -np.linalg.solve(Qr, Qux); k[t] = -np.linalg.solve(Qr, Qu)
            dV += [k[t] @ Qu, 0.5 * k[t] @ Quu @ k[t]]
            Vx = Qx + K[t].T @ Quu @ k[t] + K[t].T @ Qu + Qux.T @ k[t]
            Vxx = Qxx + K[t].T @ Quu @ K[t] + K[t].T @ Qux + Qux.T @ K[t]
            Vxx = 0.5 * (Vxx + Vxx.T)
        return K, k, dV

    reg, done, al_done, log, v = reg0, False, False, [], 0.0
    AB = [stages[t].jac(X[t], U[t]) for t in range(T)]
    for al in range(max_al_iter):
        J = merit(X, U)
        for it in range(max_iter):
            if done:
                break
            r, res = reg, None
            while True:
                res = backward(AB, r)
                if res is not None or r >= 1e6:
                    break
                r = min(max(r, 1e-8) * 10.0, 1e6)
            step, Jn, cand = -1, J, None
            if res is not None:
                K, k, dV = res
                for ia, a in enumerate(alphas):
                    Xc, Uc, okc = roll(U, (a, X, K, k))
                    Jc = merit(Xc, Uc)
                    if okc and np.isfinite(Jc) and Jc <= J + c1 * (a * dV[0] + a * a * dV[1]):
                        step, Jn, cand = ia, Jc, (Xc, Uc)
                        break
            dJ = J - Jn
            if step >= 0:
                X, U = cand
                J = Jn
                AB = [stages[t].jac(X[t], U[t]) for t in range(T)]
                reg = max(reg / 5.0, reg0)
                done = dJ < obj_tol
            else:
                reg = min(reg * 10.0, 1e6)
                done = reg >= 1e6
            log.append(dict(al=al, step=step, reg=reg, rho=rho, J=J, dJ=dJ, reg_used=r))
        if not constrained:
            break
        v = viol(X, U)
        if v < con_tol:
            al_done, done = True, True
            break
        if al + 1 == max_al_iter:
            break
        cs, ct = cons(X, U)
        for t, c in enumerate(cs):
            if c is not None:
                l = lam_s[t] + rho * c[0]; kk = stages[t].n_ineq
                l[:kk] = np.where(l[:kk] > 0.0, l[:kk], 0.0)
                lam_s[t] = l
        if ct is not None:
            l = lam_t + rho * ct[0]
            l[:nt_ineq] = np.where(l[:nt_ineq] > 0.0, l[:nt_ineq], 0.0)
            lam_t = l
        rho = min(rho * rho_scale, rho_max)
        reg, done = reg0, False
    return dict(X=X, U=U, J=J, objective=merit(X, U, False), violation=v, al_done=al_done, done=done, log=log, rho=rho, reg=reg)


def hopper_gait_stages(sim, h=0.05, T=20, foot_radius=0.05, r_cost=0.1, q_cost=0.1):
    """examples/hopper.jl as shipped (GAIT 1): stage 1 R^8 x R^10 -> R^16 (f1 / f1x / f1u :52-101: the controls carry the initial
    configurations theta = [q1; q2]), stages t >= 2 R^16 x R^2 -> R^16 (ft / ftx / ftu :103-160), obj1 / objt / objT :207-226,
    stage1_con / staget_con / terminal_con :234-266.  -> (stages, terminal_cost, terminal_con, nt_ineq, x1, U0)"""
    from . import oracle as O
    nq, nu = 4, 2
    q1 = np.array([0.0, 0.5 + foot_radius, 0.0, 0.5]); q_ref = np.array([0.5, 0.75 + foot_radius, 0.0, 0.25])
    x1 = np.concatenate([q1, q1]); x_ref = np.concatenate([q_ref, q_ref])
    w = np.array([1.0, 10.0, 1.0, 10.0] * 2)
    foot = lambda q: np.array([q[0] + q[3] * np.sin(q[2]), q[1] - q[3] * np.cos(q[2])])
    dfoot = lambda q: np.array([[1.0, 0.0, q[3] * np.cos(q[2]), np.sin(q[2])], [0.0, 1.0, q[3] * np.sin(q[2]), -np.cos(q[2])]])

    def step8(x8, u2):
        ok, d, it = O.f(sim, x8, u2)
        return ok, d

    def jac8(x8, u2):
        return O.fx(sim, x8, u2)[1], O.fu(sim, x8, u2)[1]

    def f1(x, u):
        ok, d = step8(u[nu:nu + 8], u[:nu])
        return ok, np.concatenate([d, u[nu:nu + 8]])

    def f1jac(x, u):
        A8, B8 = jac8(u[nu:nu + 8], u[:nu])
        A = np.zeros((16, 8)); Bm = np.zeros((16, 10))
        Bm[:8, :nu] = B8; Bm[:8, nu:] = A8; Bm[8:, nu:] = np.eye(8)
        return A, Bm

    def ft(x, u):
        ok, d = step8(x[:8], u)
        return ok, np.concatenate([d, x[8:]])

    def ftjac(x, u):
        A8, B8 = jac8(x[:8], u)
        A = np.zeros((16, 16)); A[:8, :8] = A8; A[8:, 8:] = np.eye(8)
        Bm = np.zeros((16, 2)); Bm[:8] = B8
        return A, Bm

    R1 = np.concatenate([r_cost * np.ones(nu), 1.0e-1 * np.ones(nq), 1.0e-5 * np.ones(nq)])

    def obj1(x, u):
        dx = x - x_ref
        return (0.5 * dx @ (w * dx) + 0.5 * u @ (R1 * u), w * dx, R1 * u, np.diag(w), np.diag(R1), np.zeros((10, 8)))

    wt = np.concatenate([q_cost * w, np.zeros(8)]); xr16 = np.concatenate([x_ref, np.zeros(8)])

    def objt(x, u):
        dx = x - xr16
        return (0.5 * dx @ (wt * dx) + 0.5 * r_cost * u @ u, wt * dx, r_cost * u, np.diag(wt), r_cost * np.eye(2), np.zeros((2, 16)))

    wT = np.concatenate([np.ones(8), np.zeros(8)])

    def objT(x):
        dx = x - xr16
        return 0.5 * dx @ (wT * dx), wT * dx, np.diag(wT)

    def con1(x, u):
        qa, qb = u[nu:nu + nq], u[nu + nq:nu + 2 * nq]
        c = np.concatenate([-10.0 - u[:nu], u[:nu] - 10.0, qa - x1[:nq], foot(qa) - foot(x1[:nq]), foot(qb) - foot(x1[nq:])])
        cu = np.zeros((12, 10))
        cu[0:2, 0:2] = -np.eye(2); cu[2:4, 0:2] = np.eye(2); cu[4:8, 2:6] = np.eye(4); cu[8:10, 2:6] = dfoot(qa); cu[10:12, 6:10] = dfoot(qb)
        return c, np.zeros((12, 8)), cu

    def cont(x, u):
        return np.concatenate([-10.0 - u, u - 10.0]), np.zeros((4, 16)), np.vstack([-np.eye(2), np.eye(2)])

    def conT(x):
        th = x[8:]
        c = np.concatenate([[0.5 - (x[0] - th[0]), 0.5 - (x[4] - th[4])], x[[1, 2, 3]] - th[[1, 2, 3]], x[[5, 6, 7]] - th[[5, 6, 7]]])
        cx = np.zeros((8, 16))
        cx[0, 0], cx[0, 8] = -1.0, 1.0
        cx[1, 4], cx[1, 12] = -1.0, 1.0
        for k, i in enumerate([1, 2, 3, 5, 6, 7]):
            cx[2 + k, i], cx[2 + k, 8 + i] = 1.0, -1.0
        return c, cx

    stages = [Stage(f1, f1jac, obj1, con1, 4)] + [Stage(ft, ftjac, objt, cont, 4) for _ in range(T - 1)]
    ustand = np.array([0.0, 9.81 * 3.0 * 0.5 * h])
    U0 = [np.concatenate([ustand, x1])] + [ustand.copy() for _ in range(T - 1)]
    return stages, objT, conT, 2, x1, U0
