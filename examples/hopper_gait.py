"""examples/hopper.jl of the reference, on the MI355X engine: a periodic hopping gait found by optimising the controls
AND the initial configurations (T = 21, h = 0.05, kappa_eval 1e-4, kappa_grad 1e-3; objective, constraints and solver
options of examples/hopper.jl:176-290, GAIT 1).

The reference's first stage has its own dimensions (x in R^8 -> R^16, u in R^10) and later stages carry the initial
configurations theta along (x in R^16, u in R^2).  Here every stage uses n = 16, m = 10: the first stage ignores its
state, later stages ignore (and lightly penalise) controls 3..10.  The stage Jacobians are the exact ones of these
maps; the reference's f1u sets d(q2 slot)/d(theta_1) = I and omits d(theta)/d(theta) (examples/hopper.jl:93-99), which
reads like an indexing slip and is not reproduced.
`python examples/hopper_gait.py [P]` solves P copies.
`python examples/hopper_gait.py device [P]` solves the example AS SHIPPED -- initial configurations optimised through the parameter stage
(od_ilqr_set_parameter_stage), the generated foot-position constraint, the terminal constraint coupled with theta -- by od_ilqr_solve: the
whole solve on the GPU, no host round trip per iteration.  `python examples/hopper_gait.py device-fixed [P]`: the same gait with the
initial configurations held at the standing pose (uniform stages only)."""
import sys

import numpy as np
import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # run from anywhere
import optimization_dynamics_amd as od  # noqa: E402
from optimization_dynamics_amd import ilqr_al as iLQR

NQ, NU, N, M = 4, 2, 16, 10


class Stage1:
    """f1 / f1x / f1u (examples/hopper.jl:52-101): u = [u1; q1; q2] -> [q2; q3; q1; q2]"""

    def __init__(self, im):
        self.im = im

    def step(self, x, u):
        d = self.im.step(u[NU:NU + 2 * NQ].contiguous(), u[:NU].contiguous())[0]
        return torch.cat([d, u[NU:NU + 2 * NQ]], 0)

    def step_grad(self, x, u):
        P = x.shape[-1]
        d, DX, DU, st, it = self.im.step_grad(u[NU:NU + 2 * NQ].contiguous(), u[:NU].contiguous())
        dx = torch.zeros(N, N, P, dtype=torch.float64, device=x.device)
        du = torch.zeros(N, M, P, dtype=torch.float64, device=x.device)
        du[:2 * NQ, :NU] = DU                       # [0; dq3/du1]
        du[:2 * NQ, NU:] = DX                       # [[0 I]; [dq3/dq1 dq3/dq2]]
        du[2 * NQ:, NU:] = torch.eye(2 * NQ, dtype=torch.float64, device=x.device)[:, :, None]
        return torch.cat([d, u[NU:NU + 2 * NQ]], 0), dx, du


class StageT:
    """ft / ftx / ftu (examples/hopper.jl:103-160): x = [q1; q2; theta], u = [u1; unused] -> [q2; q3; theta]"""

    def __init__(self, im):
        self.im = im

    def step(self, x, u):
        d = self.im.step(x[:2 * NQ].contiguous(), u[:NU].contiguous())[0]
        return torch.cat([d, x[2 * NQ:]], 0)

    def step_grad(self, x, u):
        P = x.shape[-1]
        d, DX, DU, st, it = self.im.step_grad(x[:2 * NQ].contiguous(), u[:NU].contiguous())
        dx = torch.zeros(N, N, P, dtype=torch.float64, device=x.device)
        du = torch.zeros(N, M, P, dtype=torch.float64, device=x.device)
        dx[:2 * NQ, :2 * NQ] = DX
        dx[2 * NQ:, 2 * NQ:] = torch.eye(2 * NQ, dtype=torch.float64, device=x.device)[:, :, None]
        du[:2 * NQ, :NU] = DU
        return torch.cat([d, x[2 * NQ:]], 0), dx, du


def kinematics_foot(q):                              # RoboDojo hopper: [q1 + q4 sin q3; q2 - q4 cos q3]
    return torch.stack([q[0] + q[3] * torch.sin(q[2]), q[1] - q[3] * torch.cos(q[2])])


def problem(P=1, T=21, h=0.05, device="cuda", lib=None, **opts):
    od.hopper.friction[:] = [0.5, 0.5]
    im = od.ImplicitDynamics(od.hopper, h, r_tol=1.0e-8, kappa_eval_tol=1.0e-4, kappa_grad_tol=1.0e-3, device=device, lib=lib)
    dev = im.device
    t64 = lambda a: torch.tensor(a, dtype=torch.float64, device=dev)
    foot_radius, gravity, mass_body = 0.05, 9.81, 3.0
    q1 = [0.0, 0.5 + foot_radius, 0.0, 0.5]                                             # hopper.jl:176-184
    q_ref = [0.5, 0.75 + foot_radius, 0.0, 0.25]
    x1 = t64(q1 + q1)
    x_ref = t64(q_ref + q_ref)
    r_cost, q_cost = 1.0e-1, 1.0e-1                                                     # GAIT 1, hopper.jl:190-194
    w8 = t64([1.0, 10.0, 1.0, 10.0, 1.0, 10.0, 1.0, 10.0])
    wu1 = t64([r_cost] * NU + [1.0e-1] * NQ + [1.0e-5] * NQ)
    wut = t64([r_cost] * NU + [1.0] * (2 * NQ))          # controls 3..10 are padding after the first stage

    def obj1(x, u):                                                                     # hopper.jl:203-208 (its x is x1, a constant)
        e = x1 - x_ref
        return 0.5 * (w8 * e) @ e + 0.5 * (wu1 * u) @ u

    def objt(x, u):                                                                     # hopper.jl:210-215
        e = x[:2 * NQ] - x_ref
        return 0.5 * q_cost * (w8 * e) @ e + 0.5 * (wut * u) @ u

    def objT(x, u):                                                                     # hopper.jl:217-221
        e = x[:2 * NQ] - x_ref
        return 0.5 * e @ e

    ul, uu = t64([-10.0, -10.0]), t64([10.0, 10.0])                                     # hopper.jl:229-230
    kf1, kf2 = kinematics_foot(x1[:NQ]), kinematics_foot(x1[NQ:])

    def stage1_con(x, u):                                                               # hopper.jl:232-243
        return torch.cat([ul - u[:NU], u[:NU] - uu, u[NU:NU + NQ] - x1[:NQ],
                          kinematics_foot(u[NU:NU + NQ]) - kf1, kinematics_foot(u[NU + NQ:NU + 2 * NQ]) - kf2])

    def staget_con(x, u):                                                               # hopper.jl:245-250
        return torch.cat([ul - u[:NU], u[:NU] - uu])

    def terminal_con(x, u):                                                             # hopper.jl:252-261
        x_travel = 0.5
        th = x[2 * NQ:]
        return torch.cat([torch.stack([x_travel - (x[0] - th[0]), x_travel - (x[NQ] - th[NQ])]),
                          x[1:NQ] - th[1:NQ], x[NQ + 1:2 * NQ] - th[NQ + 1:2 * NQ]])

    s1, st = Stage1(im), StageT(im)
    c1, ct, cT = iLQR.Cost(obj1), iLQR.Cost(objt), iLQR.Cost(objT)
    con1 = iLQR.Constraint(stage1_con, idx_ineq=range(4))
    cont = iLQR.Constraint(staget_con, idx_ineq=range(4))
    conT = iLQR.Constraint(terminal_con, idx_ineq=range(2))
    o = dict(alpha_min=1.0e-5, obj_tol=1.0e-3, grad_tol=1.0e-3, max_iter=10, max_al_iter=15, con_tol=0.001,
             rho_init=1.0, rho_scale=10.0)                                              # hopper.jl:273-282
    o.update(opts)
    solver = iLQR.Solver([s1] + [st] * (T - 2), [c1] + [ct] * (T - 2) + [cT], [con1] + [cont] * (T - 2) + [conT], N, M, im=im, **o)
    U0 = torch.zeros(M, T - 1, P, dtype=torch.float64, device=dev)                      # u_stand, hopper.jl:270
    U0[1] = gravity * mass_body * 0.5 * h
    U0[NU:, 0] = x1[:, None]
    X1 = torch.zeros(N, P, dtype=torch.float64, device=dev)
    X1[:2 * NQ] = x1[:, None]
    return solver, X1, U0


def problem_device_full(P=1, T=21, h=0.05, device="cuda", lib=None):
    """examples/hopper.jl as shipped for od_ilqr_solve (parameter stage): (ILQR solver, theta0, U0, options)"""
    from optimization_dynamics_amd import ilqr as IL
    od.hopper.friction[:] = [0.5, 0.5]
    im = od.ImplicitDynamics(od.hopper, h, r_tol=1.0e-8, kappa_eval_tol=1.0e-4, kappa_grad_tol=1.0e-3, device=device, lib=lib)
    foot_radius, gravity, mass_body = 0.05, 9.81, 3.0
    q1 = np.array([0.0, 0.5 + foot_radius, 0.0, 0.5]); q_ref = np.array([0.5, 0.75 + foot_radius, 0.0, 0.25])
    x1v, x_ref = np.concatenate([q1, q1]), np.concatenate([q_ref, q_ref])
    w = np.array([1.0, 10.0, 1.0, 10.0] * 2)
    obj = IL.QuadraticObjective(0.1 * np.diag(w), 0.1 * np.eye(NU), np.eye(2 * NQ), x_ref=x_ref, device=im.device)              # objt / objT, hopper.jl:212-226
    obj.set_constraints(stage=(np.zeros((4, 2 * NQ)), np.vstack([-np.eye(NU), np.eye(NU)]), np.full(4, 10.0), 4))               # control limits, :236-238,251-254
    Ctx = np.zeros((8, 2 * NQ)); Cth = np.zeros((8, 2 * NQ)); dt = np.zeros(8)                                                  # terminal_con, :256-262
    Ctx[0, 0], Cth[0, 0], dt[0] = -1.0, 1.0, -0.5
    Ctx[1, NQ], Cth[1, NQ], dt[1] = -1.0, 1.0, -0.5
    for k, i in enumerate([1, 2, 3, NQ + 1, NQ + 2, NQ + 3]):
        Ctx[2 + k, i], Cth[2 + k, i] = 1.0, -1.0
    obj.set_parameter_stage(np.array([1.0e-1] * NQ + [1.0e-5] * NQ), constraint="hopper_foot", p=x1v, terminal=(Ctx, Cth, dt, 2),      # obj1's weights on theta (:209), stage1_con (:240-247)
                            cost_const=0.5 * float((x1v - x_ref) @ (w * (x1v - x_ref))))
    U0 = np.zeros((NU, T - 1, P)); U0[1] = gravity * mass_body * 0.5 * h                                                        # hopper.jl:270
    U0[:, :, 1:] += 1e-2 * np.random.default_rng(1).normal(size=(NU, T - 1, P - 1))
    solver = IL.ILQR(im, obj, T - 1, alphas=tuple(2.0 ** -i for i in range(17)))
    opts = dict(max_iter=10, max_al_iter=15, con_tol=1.0e-3, obj_tol=1.0e-3, rho_init=1.0, rho_scale=10.0)                      # hopper.jl:273-282
    dev = im.device
    return solver, torch.tensor(np.repeat(x1v[:, None], P, axis=1), device=dev), torch.tensor(U0, device=dev), opts


def problem_device(P=1, T=21, h=0.05, device="cuda", lib=None):
    """the gait with theta = x1 for od_ilqr_solve: (ILQR solver, x1, U0, options)"""
    from optimization_dynamics_amd import ilqr as IL
    od.hopper.friction[:] = [0.5, 0.5]
    im = od.ImplicitDynamics(od.hopper, h, r_tol=1.0e-8, kappa_eval_tol=1.0e-4, kappa_grad_tol=1.0e-3, device=device, lib=lib)
    foot_radius, gravity, mass_body = 0.05, 9.81, 3.0
    q1 = np.array([0.0, 0.5 + foot_radius, 0.0, 0.5]); q_ref = np.array([0.5, 0.75 + foot_radius, 0.0, 0.25])
    x1v, x_ref = np.concatenate([q1, q1]), np.concatenate([q_ref, q_ref])
    obj = IL.QuadraticObjective(0.1 * np.diag([1.0, 10.0, 1.0, 10.0] * 2), 0.1 * np.eye(NU), np.eye(2 * NQ), x_ref=x_ref, device=im.device)   # hopper.jl:210-221
    Cs = np.zeros((4, 2 * NQ)); Ds = np.vstack([-np.eye(NU), np.eye(NU)]); ds = np.full(4, 10.0)                       # hopper.jl:245-250
    Ct = np.zeros((8, 2 * NQ)); dt = np.zeros(8)                                                                       # hopper.jl:252-261 with theta = x1
    Ct[0, 0] = -1.0; dt[0] = -(0.5 + x1v[0]); Ct[1, NQ] = -1.0; dt[1] = -(0.5 + x1v[NQ])
    for k, i in enumerate([1, 2, 3, NQ + 1, NQ + 2, NQ + 3]):
        Ct[2 + k, i] = 1.0; dt[2 + k] = x1v[i]
    obj.set_constraints(stage=(Cs, Ds, ds, 4), terminal=(Ct, dt, 2))
    U0 = np.zeros((NU, T - 1, P)); U0[1] = gravity * mass_body * 0.5 * h                                                # hopper.jl:270
    U0[:, :, 1:] += 1e-2 * np.random.default_rng(1).normal(size=(NU, T - 1, P - 1))
    solver = IL.ILQR(im, obj, T - 1, alphas=tuple(2.0 ** -i for i in range(17)))                                        # alpha_min 1e-5
    opts = dict(max_iter=10, max_al_iter=15, con_tol=1.0e-3, obj_tol=1.0e-3, rho_init=1.0, rho_scale=10.0)              # hopper.jl:273-282
    dev = im.device
    return solver, torch.tensor(np.repeat(x1v[:, None], P, axis=1), device=dev), torch.tensor(U0, device=dev), opts


if __name__ == "__main__":
    import time
    if len(sys.argv) > 1 and sys.argv[1] in ("device", "device-fixed"):
        P = int(sys.argv[2]) if len(sys.argv) > 2 else 1
        solver, x1, U0, opts = (problem_device_full if sys.argv[1] == "device" else problem_device)(P)
        solver.solve(x1, U0, **dict(opts, max_iter=1, max_al_iter=1)); solver._dev = None      # buffers, lazy loads
        torch.cuda.synchronize(); t0 = time.time()
        X, U, J, hist = solver.solve(x1, U0, **opts)
        torch.cuda.synchronize(); dt = time.time() - t0
        info = solver._dev.info(); fl, viol, rho = solver._dev.status()
        print("device-resident solve, %d problem(s): %d iterations, %d multiplier rounds, %.3f s" % (P, info.iterations, info.al_iterations, dt))
        print("max constraint violation %.2e, travel %.3f m, objective %.3f .. %.3f" % (viol.max().item(), (X[NQ, -1] - X[NQ, 0]).min().item(), J.min().item(), J.max().item()))
        if sys.argv[1] == "device":
            print("optimised initial configurations theta = [q1; q2] of problem 0:", np.array2string(X[:, 0, 0].cpu().numpy(), precision=4))
        print("configurations q_t of problem 0 (x, z, angle, leg):")
        print(np.array2string(X[NQ:2 * NQ, :, 0].T.cpu().numpy()[::4], precision=3))
        sys.exit(0)
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    solver, x1, U0 = problem(P, verbose=True)
    t0 = time.time()
    X, U = solver.solve(x1, U0)
    torch.cuda.synchronize()
    print("iterations %d, %.2f s" % (solver.iters, time.time() - t0))
    print("objective", solver.objective(X, U).cpu().numpy())
    print("max constraint violation", solver.violation(X, U).cpu().numpy())
    q = X[NQ:2 * NQ, :, 0].T.cpu().numpy()
    print("configurations q_t of copy 0 (x, z, angle, leg):")
    print(np.array2string(q[::4], precision=3))
