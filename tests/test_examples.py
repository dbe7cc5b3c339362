"""The reference's example problems driven end to end through the engine (examples/*.py mirror examples/*.jl):
augmented-Lagrangian iLQR (optimization_dynamics_amd.ilqr_al) around od_step / od_step_grad / od_ilqr_backward.
IterativeLQR.jl is un-vendored, so these are property tests: constraints met to the examples' con_tol, cost reduced,
the gait periodic and travelling."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))


def _check_gait(solver, X, U, x1):
    viol = solver.violation(X, U)
    assert viol.max().item() <= 1.0e-3                                     # con_tol of examples/hopper.jl:279
    q1_first, q2_first = U[2:6, 0], U[6:10, 0]                             # optimised initial configurations
    th = X[8:, -1]
    assert (th[:4] - q1_first).abs().max().item() < 1e-12 and (th[4:] - q2_first).abs().max().item() < 1e-12
    xT = X[:8, -1]
    assert ((xT[0] - th[0]) >= 0.5 - 1e-3).all() and ((xT[4] - th[4]) >= 0.5 - 1e-3).all()        # travels half a metre
    assert (xT[1:4] - th[1:4]).abs().max().item() < 1e-3 and (xT[5:8] - th[5:8]).abs().max().item() < 1e-3   # periodic
    assert (U[:2].abs() <= 10.0 + 1e-3).all()                              # control limits
    # every knot of the solution is a converged step of the engine: re-rolling the controls reproduces the states
    Xr = solver.rollout(x1, U)
    assert (Xr - X).abs().max().item() < 1e-9


def test_hopper_gait_cpu(emu_lib):
    import hopper_gait
    solver, x1, U0 = hopper_gait.problem(1, device="cpu", lib=emu_lib)
    J0 = solver.objective(solver.rollout(x1, U0), U0)
    X, U = solver.solve(x1, U0)
    _check_gait(solver, X, U, x1)
    assert torch.isfinite(solver.objective(X, U)).all() and J0.isfinite().all()


@pytest.mark.gpu
def test_hopper_gait_gpu(gpu_lib):
    import hopper_gait
    solver, x1, U0 = hopper_gait.problem(3, device="cuda:0", lib=gpu_lib)
    U0[1, :, 1] *= 1.05                                                     # three slightly different starts
    U0[1, :, 2] *= 0.95
    X, U = solver.solve(x1, U0)
    _check_gait(solver, X, U, x1)


@pytest.mark.gpu
def test_acrobot_swing_up_gpu(gpu_lib):
    import acrobot
    solver, x1, U0, xT = acrobot.problem(1, device="cuda:0", lib=gpu_lib)
    X, U = solver.solve(x1, U0)
    assert (X[:, -1] - xT[:, None]).abs().max().item() <= 1.0e-3            # con_tol of examples/acrobot.jl:104
    assert (X[1].abs() <= np.pi / 2 + 1e-3).all() and (X[3].abs() <= np.pi / 2 + 1e-3).all()     # joint limits held throughout
    assert solver.objective(X, U).item() < 500.0


@pytest.mark.gpu
@pytest.mark.parametrize("GB", [False, True])
def test_planar_push_rotate_gpu(gpu_lib, GB):
    """examples/planar_push.jl, MODE = :rotate, with the implicit-function gradient and with the gradient bundle"""
    import planar_push
    solver, x1, U0, xT = planar_push.problem("rotate", 1, GB=GB, device="cuda:0", lib=gpu_lib)
    X, U = solver.solve(x1, U0)
    assert solver.violation(X, U).max().item() <= 5.0e-3                    # con_tol of examples/planar_push.jl:124
    assert (X[5:8, -1] - xT[5:8, None]).abs().max().item() <= 5.0e-3        # block at (0.5, 0.5, pi/2)
    assert (U.abs() <= 5.0 + 5e-3).all()


@pytest.mark.gpu
def test_cartpole_frictionless_swing_up_gpu(gpu_lib):
    import cartpole
    solver, x1, U0, xT = cartpole.problem("frictionless", 2, device="cuda:0", lib=gpu_lib)
    U0[:, 0, 1] = -1.4
    X, U = solver.solve(x1, U0)
    assert (X[:, -1] - xT[:, None]).abs().max().item() <= 5.0e-3            # con_tol of examples/cartpole.jl:92


def test_hopper_gait_device_script_cpu(emu_lib):
    """examples/hopper_gait.py `device` mode: the example as shipped through od_ilqr_solve (host build here)"""
    import hopper_gait
    solver, x1, U0, opts = hopper_gait.problem_device_full(1, device="cpu", lib=emu_lib)
    X, U, J, hist = solver.solve(x1, U0, **opts)
    fl, viol, rho = solver._dev.status()
    assert viol.max().item() < 1e-3 and (X[4, -1] - X[4, 0]).min().item() >= 0.5 - 1e-3
    assert solver._dev.info().iterations == 33
