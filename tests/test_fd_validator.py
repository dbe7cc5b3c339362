"""SURVEY.md 8(f).4, the part that can be built here: the reference's comparison of implicit gradients against FINITE-DIFFERENCE
Jacobians inside iLQR (examples/comparisons/acrobot/acrobot.jl:30-36; MuJoCo there, the CPU oracle's own step here --
oracle/fd_validator.py) on the acrobot swing-up of examples/acrobot.jl, and the same task through the device-resident solver."""
import math

import numpy as np
import pytest
import torch


def _device_solution(lib, device, B):
    import optimization_dynamics_amd as od
    from optimization_dynamics_amd import ilqr as IL
    h, T = 0.05, 100
    im = od.ImplicitDynamics(od.acrobot_impact, h, r_tol=1e-8, kappa_eval_tol=1e-4, kappa_grad_tol=1e-3, device=device, lib=lib)
    I2 = np.eye(2)
    Q = 0.1 / h ** 2 * np.block([[I2, -I2], [-I2, I2]])
    xT = np.array([math.pi, 0.0, math.pi, 0.0])
    obj = IL.QuadraticObjective(Q, np.eye(1), Q, x_ref=np.zeros(4), goal_idx=[0, 1, 2, 3], goal=xT, device=device)
    U0 = np.stack([1e-3 * np.random.default_rng(1 + b).normal(size=(1, T)) for b in range(B)], axis=-1)     # (trajectory 0: the validator's)
    sol = IL.ILQR(im, obj, T)
    X, U, J, hist = sol.solve(torch.zeros(4, B, dtype=torch.float64, device=device), torch.tensor(U0, device=device),
                              max_iter=50, max_al_iter=20, con_tol=1e-3, obj_tol=1e-5)
    viol = (X[:, -1] - torch.tensor(xT, device=device)[:, None]).abs().max(0).values
    return obj.value(X, U).cpu().numpy(), viol.cpu().numpy(), sol._dev.info()


def test_finite_difference_jacobians_against_implicit_gradients(oracle, emu_lib):
    from oracle import fd_validator as V
    imp = V.solve("implicit")
    fd = V.solve("fd")
    # both reach the goal of examples/acrobot.jl to its con_tol (:104), with comparable objectives: the implicit gradient at
    # kappa_grad = 1e-3 is a usable search direction, as is the finite difference of the kappa_eval = 1e-4 step
    assert imp["violation"] < 1e-3 and fd["violation"] < 1e-3, (imp["violation"], fd["violation"])
    assert abs(imp["objective"] - fd["objective"]) < 0.1 * fd["objective"], (imp["objective"], fd["objective"])
    print("acrobot swing-up, CPU oracle: implicit gradients J = %.3f in %d iterations; finite differences J = %.3f in %d iterations"
          % (imp["objective"], imp["iterations"], fd["objective"], fd["iterations"]))
    # the device-resident solver (host build of the same sources) on the same task and initial controls
    J, viol, info = _device_solution(emu_lib, "cpu", 2)
    assert viol.max() < 1e-3 and info.al_done == 1
    assert abs(J[0] - imp["objective"]) < 0.1 * imp["objective"], (J[0], imp["objective"])


@pytest.mark.gpu
def test_acrobot_swing_up_on_the_device(oracle, gpu_lib):
    """examples/acrobot.jl (T = 101, terminal equality constraint by augmented Lagrangian, options of :98-108) through od_ilqr_solve,
    64 problems: every one reaches the goal to con_tol; and problem 0 (the validator's initial controls) against the numpy AL-iLQR on
    the CPU oracle with implicit gradients (oracle/fd_validator.py::solve -> the same loop as oracle/ilqr_np.py::solve): the same
    task solved to the same constraint tolerance with an objective within 5 % -- the decisions of the first ~50 iterations are
    compared one by one in tests/test_ilqr.py::test_solver_decisions_against_the_numpy_oracle_gpu, after which a joint-limit impact
    amplifies the 1e-12 between the two implementations of the step"""
    from oracle import fd_validator as V
    J, viol, info = _device_solution(gpu_lib, "cuda:0", 64)
    assert viol.max() < 1e-3 and info.al_done == 1, (viol.max(), info.al_done)
    imp = V.solve("implicit")
    assert imp["violation"] < 1e-3
    assert abs(J[0] - imp["objective"]) < 0.05 * imp["objective"], (J[0], imp["objective"])
    # the other 63 start from other random controls of the same size: the same swing-up, objectives in a band around the validator's
    assert (np.abs(J - imp["objective"]) < 0.25 * imp["objective"]).all(), (J.min(), J.max(), imp["objective"])


def test_direct_method_against_ilqr_on_the_hopper_gait(oracle):
    """SURVEY.md 8(f).4, the direct-method leg (examples/comparisons/hopper.jl:57-162,290-303,318-357) on the CPU oracle: the hopper's
    gait task as one nonlinear programme over configurations, controls, contact impulses and complementarity slacks with the oracle's
    residual as constraints (scipy trust-constr standing in for Ipopt, which is absent like MuJoCo), started from the iLQR solution,
    priced under the iLQR cost beside it (oracle/direct_validator.py).  The two methods agree on the task: the direct solution is
    feasible, its objective within a few per cent of the iLQR solution's -- it may not be beaten by much by a method that is handed the
    iLQR answer as a start --, and its controls drive the time-stepping simulator along the same motion."""
    from oracle import direct_validator as D
    r = D.compare(maxiter=400)
    print("hopper gait: iLQR J = %.4f (violation %.1e, %d iterations); direct J = %.4f (equality rows %.1e, slack max %.1e, %d iterations, optimality %.1e)"
          % (r["ilqr_objective"], r["ilqr_violation"], r["ilqr_iterations"], r["direct_objective"], r["direct_equality_violation"], r["direct_slack_max"],
             r["direct_iterations"], r["direct_optimality"]))
    assert r["ilqr_violation"] < 1e-3 and r["travel_ilqr"] >= 0.5 - 1e-3
    assert r["direct_equality_violation"] < 1e-5 and r["direct_inequality_violation"] < 1e-5 and r["direct_terminal_violation"] < 1e-5
    assert r["direct_slack_max"] < 1e-3                              # complementarity at the time-stepping simulator's own kappa_tol = 1e-4 level
    assert r["travel_direct"] >= 0.5 - 1e-5
    assert abs(r["direct_objective"] - r["ilqr_objective"]) < 0.05 * r["ilqr_objective"], (r["direct_objective"], r["ilqr_objective"])
    assert r["direct_objective"] > 0.9 * r["ilqr_objective"]
    assert r["rollout_converged"] and r["rollout_of_direct_controls_state_diff"] < 2e-2, r["rollout_of_direct_controls_state_diff"]
