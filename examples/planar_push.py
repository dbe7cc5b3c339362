"""examples/planar_push.jl of the reference, on the MI355X engine: a pusher translates or rotates a block on a frictional
plane (T = 26, h = 0.1, kappa_eval 1e-4, kappa_grad 1e-2); objective, constraints and solver options of
examples/planar_push.jl:18-131.  GB = True differentiates with the gradient bundle (fx_gb / fu_gb, N = 50, eps = 1e-4)
instead of the implicit-function gradient.  `python examples/planar_push.py [translate|rotate] [P] [gb]`."""
import math
import sys

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # run from anywhere
import optimization_dynamics_amd as od  # noqa: E402
from optimization_dynamics_amd import gradient_bundle as gbm
from optimization_dynamics_amd import ilqr_al as iLQR


class BundleStage:
    """iLQR.Dynamics(f, fx_gb, fu_gb, ...) (examples/planar_push.jl:28-32 with GB = true)"""

    def __init__(self, im, gb):
        self.im, self.gb = im, gb

    def step(self, x, u):
        return self.im.step(x.contiguous(), u.contiguous())[0]

    def step_grad(self, x, u):
        nq, nu = self.im.model.nq, self.im.model.nu
        P = x.shape[-1]
        d = self.im.step(x.contiguous(), u.contiguous())[0]
        dz, st = gbm.gradient_batch(self.im, self.gb, x.contiguous(), u.contiguous())      # (nq, 2nq+nu, P)
        dx = torch.zeros(2 * nq, 2 * nq, P, dtype=torch.float64, device=x.device)
        du = torch.zeros(2 * nq, nu, P, dtype=torch.float64, device=x.device)
        dx[:nq, nq:] = torch.eye(nq, dtype=torch.float64, device=x.device)[:, :, None]
        dx[nq:, :] = dz[:, :2 * nq]
        du[nq:, :] = dz[:, 2 * nq:]
        return d, dx, du


def problem(mode="rotate", P=1, T=26, h=0.1, GB=False, device="cuda", lib=None, **opts):
    info = gbm.GradientBundle(od.planarpush, N=50, eps=1.0e-4, seed=0) if GB else None
    im = od.ImplicitDynamics(od.planarpush, h, r_tol=1.0e-8, kappa_eval_tol=1.0e-4, kappa_grad_tol=1.0e-2, nc=1, nb=9,
                             info=info, device=device, lib=lib)                         # planar_push.jl:21-22
    nq, nx, nu = 5, 10, 2
    dev = im.device
    t64 = lambda a: torch.tensor(a, dtype=torch.float64, device=dev)
    r_dim = 0.1
    if mode == "translate":                                                             # planar_push.jl:37-44
        q0 = [0.0, 0.0, 0.0, -r_dim - 1.0e-8, 0.0]
        goal = (1.0, 0.0, 0.0)
    else:                                                                               # planar_push.jl:45-54
        q0 = [0.0, 0.0, 0.0, -r_dim - 1.0e-8, -0.01]
        goal = (0.5, 0.5, 0.5 * math.pi)
    qT = [goal[0], goal[1], goal[2], goal[0] - r_dim, goal[1] - r_dim]
    xT = t64(qT + qT)
    wv = t64([1.0, 1.0, 1.0, 0.1, 0.1])
    wx = t64([1.0, 1.0, 1.0, 0.1, 0.1] * 2)
    ru = 1.0e-1 if mode == "translate" else 1.0e-2

    def objT(x, u):                                                                     # planar_push.jl:72-83
        v1 = (x[nq:] - x[:nq]) / h
        e = x - xT
        return 0.5 * (wv * v1) @ v1 + 0.5 * (wx * e) @ e

    def objt(x, u):                                                                     # planar_push.jl:57-70
        return objT(x, u) + 0.5 * ru * (u @ u)

    ul, uu = t64([-5.0, -5.0]), t64([5.0, 5.0])                                         # planar_push.jl:90-91
    goal_rows = [0, 1, 2, 5, 6, 7]
    stage = BundleStage(im, info) if GB else iLQR.ImplicitStage(im)
    costs = [iLQR.Cost(objt)] * (T - 1) + [iLQR.Cost(objT)]
    cont = iLQR.Constraint(lambda x, u: torch.cat([ul - u, u - uu]), idx_ineq=range(2 * nu))   # planar_push.jl:93-98
    conT = iLQR.Constraint(lambda x, u: (x - xT)[goal_rows])                                     # planar_push.jl:100-104
    o = dict(alpha_min=1.0e-5, obj_tol=1.0e-3, grad_tol=1.0e-3, max_iter=10, max_al_iter=10, con_tol=0.005,
             rho_init=1.0, rho_scale=10.0)                                              # planar_push.jl:117-128
    o.update(opts)
    solver = iLQR.Solver([stage] * (T - 1), costs, [cont] * (T - 1) + [conT], nx, nu, im=im, **o)
    U0 = torch.zeros(nu, T - 1, P, dtype=torch.float64, device=dev)                     # planar_push.jl:111
    U0[0, :4] = 1.0
    if mode != "translate":
        U0[0, 4:9] = 0.5
    x1 = t64(q0 + q0)[:, None].repeat(1, P)
    return solver, x1, U0, xT


if __name__ == "__main__":
    import time
    mode = sys.argv[1] if len(sys.argv) > 1 else "rotate"
    P = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    solver, x1, U0, xT = problem(mode, P, GB=len(sys.argv) > 3, verbose=True)
    t0 = time.time()
    X, U = solver.solve(x1, U0)
    torch.cuda.synchronize()
    print("iterations %d, %.2f s" % (solver.iters, time.time() - t0))
    print("objective", solver.objective(X, U).cpu().numpy())
    print("goal error (block pose at T)", (X[5:8, -1] - xT[5:8, None]).abs().max(0).values.cpu().numpy())
    print("max constraint violation", solver.violation(X, U).cpu().numpy())
