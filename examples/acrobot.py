"""examples/acrobot.jl of the reference, on the MI355X engine: swing-up of an acrobot with joint limits through
impacts (T = 101, h = 0.05, kappa_eval 1e-4, kappa_grad 1e-3; objective, terminal constraint and solver options of
examples/acrobot.jl:15-111).  `python examples/acrobot.py [P]` solves P copies (perturbed initial controls)."""
import math
import sys

import numpy as np
import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # run from anywhere
import optimization_dynamics_amd as od  # noqa: E402
from optimization_dynamics_amd import ilqr_al as iLQR


def problem(P=1, T=101, h=0.05, device="cuda", lib=None, seed=1, **opts):
    im = od.ImplicitDynamics(od.acrobot_impact, h, r_tol=1.0e-8, kappa_eval_tol=1.0e-4, kappa_grad_tol=1.0e-3,
                             device=device, lib=lib)                                   # acrobot.jl:19-23
    nq, nx, nu = 2, 4, 1
    dev = im.device
    xT = torch.tensor([math.pi, 0.0, math.pi, 0.0], dtype=torch.float64, device=dev)    # acrobot.jl:40-46

    def objt(x, u):                                                                     # acrobot.jl:49-61
        v1 = (x[nq:] - x[:nq]) / h
        return 0.5 * 0.1 * (v1 @ v1) + 0.5 * (u @ u)

    def objT(x, u):                                                                     # acrobot.jl:63-73
        v1 = (x[nq:] - x[:nq]) / h
        return 0.5 * 0.1 * (v1 @ v1)

    stage = iLQR.ImplicitStage(im)
    costs = [iLQR.Cost(objt)] * (T - 1) + [iLQR.Cost(objT)]
    cons = [iLQR.Constraint()] * (T - 1) + [iLQR.Constraint(lambda x, u: x - xT)]       # acrobot.jl:80-88
    o = dict(alpha_min=1.0e-5, obj_tol=1.0e-5, grad_tol=1.0e-5, max_iter=50, max_al_iter=20, con_tol=0.001,
             rho_init=1.0, rho_scale=10.0)                                              # acrobot.jl:98-108
    o.update(opts)
    solver = iLQR.Solver([stage] * (T - 1), costs, cons, nx, nu, im=im, **o)
    rng = np.random.default_rng(seed)                                                   # acrobot.jl:90-91
    U0 = torch.tensor(1.0e-3 * rng.normal(size=(nu, T - 1, P)), device=dev)
    x1 = torch.zeros(nx, P, dtype=torch.float64, device=dev)
    return solver, x1, U0, xT


if __name__ == "__main__":
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    solver, x1, U0, xT = problem(P, verbose=True)
    import time
    t0 = time.time()
    X, U = solver.solve(x1, U0)
    torch.cuda.synchronize()
    print("iterations %d, %.2f s" % (solver.iters, time.time() - t0))
    print("objective", solver.objective(X, U).cpu().numpy())
    print("terminal constraint violation", (X[:, -1] - xT[:, None]).abs().max(0).values.cpu().numpy())
