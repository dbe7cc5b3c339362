"""examples/rocket.jl of the reference, on the MI355X engine: soft landing of a rocket with the thrust-cone constraint either
projected inside the dynamics (MODE = "projection": f_rocket_proj and its implicit gradients) or as stage constraints
(MODE = "nominal"); T = 61, h = 0.05, u_max = 12.5; objective, constraints and solver options of examples/rocket.jl:15-137.
`python examples/rocket.py [projection|nominal] [P]`."""
import math
import sys

import numpy as np
import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # run from anywhere
import optimization_dynamics_amd as od  # noqa: E402
from optimization_dynamics_amd import ilqr_al as iLQR


class RocketStage:
    """iLQR.Dynamics(f_rocket(_proj), fx_rocket(_proj), fu_rocket(_proj), ...) (examples/rocket.jl:29-41)"""

    def __init__(self, info, project):
        self.info, self.project = info, project

    def step(self, x, u):
        return self.info.solve(x.contiguous(), u.contiguous(), project=self.project, grads=False)[0].double()

    def step_grad(self, x, u):
        Y, DX, DU, UP, st = self.info.solve(x.contiguous(), u.contiguous(), project=self.project, grads=True)
        return Y.double(), DX.double(), DU.double()


def mrp_of(Rm):
    """modified Rodrigues parameters of a rotation matrix (Rotations.MRP: q_vec / (1 + q_w))"""
    w = 0.5 * math.sqrt(max(0.0, 1.0 + Rm[0, 0] + Rm[1, 1] + Rm[2, 2]))
    v = np.array([Rm[2, 1] - Rm[1, 2], Rm[0, 2] - Rm[2, 0], Rm[1, 0] - Rm[0, 1]]) / (4.0 * w)
    return v / (1.0 + w)


def rot_z(a):
    return np.array([[math.cos(a), -math.sin(a), 0.0], [math.sin(a), math.cos(a), 0.0], [0.0, 0.0, 1.0]])


def rot_y(a):
    return np.array([[math.cos(a), 0.0, math.sin(a)], [0.0, 1.0, 0.0], [-math.sin(a), 0.0, math.cos(a)]])


def problem(mode="projection", P=1, T=61, h=0.05, u_max=12.5, device="cuda", lib=None, seed=1, **opts):
    info = od.RocketInfo(od.rocket, u_max, h, device=device, lib=lib)                   # rocket.jl:16-23
    dev = info.device
    t64 = lambda a: torch.tensor(np.asarray(a, dtype=np.float64), device=dev)
    nx, nu, length = 12, 3, 1.0                                                         # src/models/rocket/model.jl:35-48
    x1 = np.zeros(nx); x1[:3] = [2.5, 2.5, 10.0]                                        # rocket.jl:44-50
    x1[3:6] = mrp_of(rot_z(0.25 * math.pi) @ rot_y(-0.5 * math.pi)); x1[8] = -1.0
    xT = np.zeros(nx); xT[2] = length                                                   # rocket.jl:52-55
    xT[3:6] = mrp_of(rot_z(0.25 * math.pi) @ rot_y(0.0))
    xTt = t64(xT)
    wq = t64(h * np.r_[1.0e-1 * np.ones(3), 1.0e-5 * np.ones(3), 1.0e-1 * np.ones(3), 1000.0 * np.ones(3)])
    wr = t64(h * np.array([1000.0, 1000.0, 100.0]))

    def objt(x, u):                                                                     # rocket.jl:58-65
        e = x - xTt
        return 0.5 * (wq * e) @ e + 0.5 * (wr * u) @ u

    def objT(x, u):                                                                     # rocket.jl:67-73
        e = x - xTt
        return 0.5 * h * 1000.0 * (e @ e)

    if mode == "projection":                                                            # rocket.jl:82-100
        stage_con, n_ineq = (lambda x, u: (length - x[2]).reshape(1)), 1
    else:
        one = t64([1.0])[0]
        stage_con = lambda x, u: torch.stack([-one - u[0], u[0] - one, -one - u[1], u[1] - one, -u[2], u[2] - u_max, length - x[2]])
        n_ineq = 7

    def terminal_con(x, u):                                                             # rocket.jl:102-110
        return torch.cat([torch.stack([-0.5 - x[0], x[0] - 0.5, -0.75 - x[1], x[1] - 0.75]), (x - xTt)[2:]])

    stage = RocketStage(info, mode == "projection")
    costs = [iLQR.Cost(objt)] * (T - 1) + [iLQR.Cost(objT)]
    cons = [iLQR.Constraint(stage_con, idx_ineq=range(n_ineq))] * (T - 1) + [iLQR.Constraint(terminal_con, idx_ineq=range(4))]
    o = dict(alpha_min=1.0e-5, obj_tol=1.0e-3, grad_tol=1.0e-3, max_iter=100, max_al_iter=15, con_tol=0.005,
             rho_init=1.0, rho_scale=10.0)                                              # rocket.jl:123-134
    o.update(opts)
    # the Riccati kernel is reached through any handle of the library: the rocket's own
    solver = iLQR.Solver([stage] * (T - 1), costs, cons, nx, nu, im=od.RocketDynamics(info, mode == "projection"), **o)
    rng = np.random.default_rng(seed)                                                   # rocket.jl:116-117
    U0 = t64(1.0e-3 * rng.normal(size=(nu, T - 1, P)))
    return solver, t64(x1)[:, None].repeat(1, P), U0, xTt


if __name__ == "__main__":
    import time
    mode = sys.argv[1] if len(sys.argv) > 1 else "projection"
    P = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    solver, x1, U0, xT = problem(mode, P, verbose=True)
    t0 = time.time()
    X, U = solver.solve(x1, U0)
    torch.cuda.synchronize()
    print("iterations %d, %.2f s" % (solver.iters, time.time() - t0))
    print("objective", solver.objective(X, U).cpu().numpy())
    print("terminal error (rows 3..12)", (X[2:, -1] - xT[2:, None]).abs().max(0).values.cpu().numpy())
    if mode == "projection":
        Up = solver.stages[0].info.project(U.reshape(3, -1), grads=False)[0]
        print("thrust cone holds for the applied controls:", bool((torch.hypot(Up[0], Up[1]) <= Up[2] + 1e-2).all()))   # rocket.jl:151
