"""Comparison baseline of the reference, CPU side -- TEST INFRASTRUCTURE ONLY.

`examples/comparisons/acrobot/acrobot.jl:30-36` solves the acrobot swing-up with iLQR around a simulator whose Jacobians are
FINITE DIFFERENCES of its step (MuJoCo + FiniteDiff there), as the yardstick for the implicit gradients of the paper's method.
MuJoCo is absent here (and so is Ipopt for `examples/comparisons/hopper.jl`, the direct method): those two legs cannot be built.
What can be built is the comparison itself on the CPU oracle: the same task (`examples/acrobot.jl:15-111`: T = 101, h = 0.05,
x1 = 0, x_T = [pi, 0, pi, 0], stage cost 0.05 |v1|^2 + 0.5 u^2, terminal equality constraint by augmented Lagrangian), the same
iLQR, with the dynamics Jacobians taken either

  * "implicit": fx / fu of the oracle (implicit-function theorem at kappa_grad = 1e-3, src/dynamics.jl:96-128), or
  * "fd":       central differences of the oracle's f (kappa_eval = 1e-4, src/dynamics.jl:81-94), step 1e-6,

and the outcomes side by side: both must reach the goal to con_tol with comparable objectives (tests/test_fd_validator.py).
The iLQR below is the single-trajectory numpy form of optimization_dynamics_amd/ilqr.py (same regularisation schedule, Armijo
test, multiplier update), its backward pass is oracle/ilqr_np.py."""
import math

import numpy as np

from . import ilqr_np
from . import oracle as O


def acrobot_task(h=0.05, T=100):
    I2 = np.eye(2)
    Q = 0.1 / h ** 2 * np.block([[I2, -I2], [-I2, I2]])          # 1/2 0.1 v1'v1, v1 = (q2 - q1) / h   (examples/acrobot.jl:49-61)
    R = np.eye(1)
    xT = np.array([math.pi, 0.0, math.pi, 0.0])                  # :40-46
    x1 = np.zeros(4)
    U0 = 1.0e-3 * np.random.default_rng(1).normal(size=(T, 1))   # :90-91
    return Q, R, xT, x1, U0


def make_dynamics(jacobians, h=0.05, fd_step=1e-6):
    sim = O.make_sim("acrobot_impact", h, kappa_tol=1e-4, kappa_grad_tol=1e-3)

    def f(x, u):
        ok, d, it = O.f(sim, x, u)
        return d

    if jacobians == "implicit":
        def jac(x, u):
            return O.fx(sim, x, u)[1], O.fu(sim, x, u)[1]
    else:
        def jac(x, u):
            n, m = x.size, u.size
            A, B = np.zeros((n, n)), np.zeros((n, m))
            for j in range(n):
                e = np.zeros(n); e[j] = fd_step
                A[:, j] = (f(x + e, u) - f(x - e, u)) / (2 * fd_step)
            for j in range(m):
                e = np.zeros(m); e[j] = fd_step
                B[:, j] = (f(x, u + e) - f(x, u - e)) / (2 * fd_step)
            return A, B
    return f, jac


def solve(jacobians="implicit", max_iter=50, max_al_iter=20, con_tol=1e-3, obj_tol=1e-5, rho_init=1.0, rho_scale=10.0,
          reg0=1e-6, c1=1e-4, alphas=tuple(2.0 ** -i for i in range(11)), verbose=False):
    Q, R, xT, x1, U = acrobot_task()
    T, n, m = U.shape[0], 4, 1
    f, jac = make_dynamics(jacobians)

    def rollout(U, K=None, k=None, Xb=None, Ub=None, alpha=0.0):
        X = np.zeros((T + 1, n)); X[0] = x1
        Un = np.zeros_like(U)
        for t in range(T):
            Un[t] = U[t] if K is None else Ub[t] + alpha * k[t] + K[t] @ (X[t] - Xb[t])
            X[t + 1] = f(X[t], Un[t])
        return X, Un

    def cost(X, U, lam, rho):
        J = 0.5 * sum(X[t] @ Q @ X[t] + U[t] @ R @ U[t] for t in range(T)) + 0.5 * X[T] @ Q @ X[T]
        c = X[T] - xT
        return J + lam @ c + 0.5 * rho * c @ c

    X, U = rollout(U)
    lam, rho = np.zeros(n), rho_init
    iters = 0
    for al in range(max_al_iter):
        J = cost(X, U, lam, rho)
        reg = reg0
        AB = [jac(X[t], U[t]) for t in range(T)]
        for it in range(max_iter):
            iters += 1
            A = np.stack([a for a, b in AB]); Bm = np.stack([b for a, b in AB])
            c = X[T] - xT
            Vx = Q @ X[T] + lam + rho * c
            Vxx = Q + rho * np.eye(n)
            lxx = np.repeat(Q[None], T, 0); luu = np.repeat(R[None], T, 0); lux = np.zeros((T, m, n))
            lx = X[:T] @ Q.T; lu = U @ R.T
            r = reg
            while True:
                try:
                    K, k, dV = ilqr_np.backward(A, Bm, lxx, luu, lux, lx, lu, Vxx, Vx, r)
                    ok = np.isfinite(K).all()
                except np.linalg.LinAlgError:
                    ok = False
                if ok or r >= 1e6:
                    break
                r = min(max(r, 1e-8) * 10.0, 1e6)
            took = False
            if ok:
                for a in alphas:
                    Xc, Uc = rollout(U, K, k, X, U, a)
                    Jc = cost(Xc, Uc, lam, rho)
                    if np.isfinite(Jc) and Jc <= J + c1 * (a * dV[0] + a * a * dV[1]):
                        took = True
                        break
            if took:
                dJ = J - Jc
                X, U, J = Xc, Uc, Jc
                AB = [jac(X[t], U[t]) for t in range(T)]
                reg = max(reg / 5.0, reg0)
                if dJ < obj_tol:
                    break
            else:
                reg = min(reg * 10.0, 1e6)
                if reg >= 1e6:
                    break
        viol = np.abs(X[T] - xT).max()
        if verbose:
            print("%s al %d: iterations %d  J %.4f  violation %.2e" % (jacobians, al, iters, cost(X, U, 0 * lam, 0.0), viol))
        if viol < con_tol:
            break
        lam = lam + rho * (X[T] - xT)
        rho *= rho_scale
    return dict(X=X, U=U, objective=cost(X, U, 0 * lam, 0.0), violation=viol, iterations=iters, al_rounds=al + 1)


if __name__ == "__main__":
    import time
    O.build()
    for j in ("implicit", "fd"):
        t0 = time.time()
        r = solve(j, verbose=True)
        print(j, "objective %.4f violation %.2e iterations %d (%.1f s)" % (r["objective"], r["violation"], r["iterations"], time.time() - t0))
