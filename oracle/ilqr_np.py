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
