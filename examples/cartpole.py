"""examples/cartpole.jl of the reference, on the MI355X engine: swing-up of a cartpole with Coulomb friction at both joints
(MODE = "friction", mu = 0.35) or without (MODE = "frictionless"); T = 51, h = 0.05; objective, terminal constraint and
solver options of examples/cartpole.jl:15-98.  `python examples/cartpole.py [friction|frictionless] [P]`."""
import math
import sys

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # run from anywhere
import optimization_dynamics_amd as od  # noqa: E402
from optimization_dynamics_amd import ilqr_al as iLQR


def problem(mode="friction", P=1, T=51, h=0.05, device="cuda", lib=None, **opts):
    if mode == "friction":                                                              # cartpole.jl:18-22
        od.cartpole_friction.friction[:] = [0.35, 0.35]
        im = od.ImplicitDynamics(od.cartpole_friction, h, r_tol=1.0e-8, kappa_eval_tol=1.0e-4, kappa_grad_tol=1.0e-3,
                                 no_impact=True, device=device, lib=lib)
    else:                                                                               # cartpole.jl:26-28
        im = od.ImplicitDynamics(od.cartpole_frictionless, h, r_tol=1.0e-8, kappa_eval_tol=1.0, kappa_grad_tol=1.0,
                                 no_impact=True, no_friction=True, device=device, lib=lib)
    nx, nu = 4, 1
    dev = im.device
    xT = torch.tensor([0.0, math.pi, 0.0, math.pi], dtype=torch.float64, device=dev)    # cartpole.jl:42-48
    stage = iLQR.ImplicitStage(im)
    costs = [iLQR.Cost(lambda x, u: u @ u)] * (T - 1) + [iLQR.Cost(lambda x, u: (x - xT) @ (x - xT))]   # cartpole.jl:51-61
    cons = [iLQR.Constraint()] * (T - 1) + [iLQR.Constraint(lambda x, u: x - xT)]       # cartpole.jl:68-76
    o = dict(alpha_min=1.0e-5, obj_tol=1.0e-5, grad_tol=1.0e-3, max_iter=100, max_al_iter=20, con_tol=0.005,
             rho_init=1.0, rho_scale=10.0)                                              # cartpole.jl:85-95
    o.update(opts)
    solver = iLQR.Solver([stage] * (T - 1), costs, cons, nx, nu, im=im, **o)
    U0 = torch.zeros(nu, T - 1, P, dtype=torch.float64, device=dev)                     # cartpole.jl:78
    U0[:, 0] = -1.5
    x1 = torch.zeros(nx, P, dtype=torch.float64, device=dev)
    return solver, x1, U0, xT


if __name__ == "__main__":
    import time
    mode = sys.argv[1] if len(sys.argv) > 1 else "friction"
    P = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    solver, x1, U0, xT = problem(mode, P, verbose=True)
    t0 = time.time()
    X, U = solver.solve(x1, U0)
    torch.cuda.synchronize()
    print("iterations %d, %.2f s" % (solver.iters, time.time() - t0))
    print("objective", solver.objective(X, U).cpu().numpy())
    print("terminal constraint violation", (X[:, -1] - xT[:, None]).abs().max(0).values.cpu().numpy())
