"""The device-pointer entry points of the C ABI only enqueue kernels on the handle's stream -- no allocation, no synchronisation,
no host read-back -- so a caller can record them in a HIP graph and replay it on new contents of the same buffers (an iLQR
loop's rollout, linearisation and backward pass).  Checked here by doing exactly that: capture, overwrite
the inputs in place, replay, compare bit for bit with a direct call."""
import numpy as np
import pytest
import torch

import parity_checks as P
import workloads as W

pytestmark = pytest.mark.gpu


def _capture_replay(run, mutate, outputs):
    """run() -> tuple of tensors (the same buffers every call); mutate() changes the inputs in place"""
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        run()                                   # warm-up on the capture stream (lazy module loads are not capturable)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        out = run()
    first = [t.clone() for t in outputs(out)]
    mutate()
    g.replay()
    torch.cuda.synchronize()
    replayed = [t.clone() for t in outputs(out)]
    direct = [t.clone() for t in outputs(run())]
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(replayed, direct))
    assert any(not torch.equal(a, b) for a, b in zip(replayed, first))        # (the replay did see the new inputs)


@pytest.mark.parametrize("B", [256, 5000])          # 16 lanes per problem | 8 lanes per problem
def test_rollout_compact_in_a_graph(gpu_lib, B):
    im = P.make_im("hopper", gpu_lib, "cuda:0")
    x1, U = W.hopper_rollout_inputs(B, 12, seed=0, u_sigma=1.0)
    x1d, Ud = torch.tensor(x1, device="cuda:0"), torch.tensor(U, device="cuda:0")
    bufs = im.rollout_compact(x1d, Ud)[-1]
    _capture_replay(lambda: im.rollout_compact(x1d, Ud, out=bufs), lambda: Ud.add_(0.05), lambda o: (o[0], o[1], o[2], o[3]))


@pytest.mark.parametrize("name", ["acrobot_impact", "planar_push", "cartpole_friction"])
def test_step_grad_in_a_graph(gpu_lib, name):
    im = P.make_im(name, gpu_lib, "cuda:0")
    X, U = W.knots(name, 777, seed=5)
    Xd, Ud = torch.tensor(X, device="cuda:0"), torch.tensor(U, device="cuda:0")
    _capture_replay(lambda: im.step_grad(Xd, Ud), lambda: Ud.mul_(0.9), lambda o: o)


def test_bundle_in_a_graph(gpu_lib):
    import optimization_dynamics_amd as od
    im = P.make_im("planar_push", gpu_lib, "cuda:0")
    gb = od.GradientBundle(od.planarpush, N=64, eps=1e-4, seed=0)
    X, U = W.knots("planar_push", 20, seed=2)
    Xd, Ud = torch.tensor(X, device="cuda:0"), torch.tensor(U, device="cuda:0")
    _capture_replay(lambda: od.gradient_batch(im, gb, Xd, Ud), lambda: Ud.mul_(0.9),
                    lambda o: tuple(t for t in (o if isinstance(o, (tuple, list)) else (o,)) if torch.is_tensor(t)))


def test_rocket_and_riccati_in_a_graph(gpu_lib):
    import ilqr_checks as C
    import optimization_dynamics_amd as od
    dyn, obj, x1, U0 = C.rocket_problem(gpu_lib, "cuda:0", 64, 10, dtype=torch.float32, seed=1)
    x1t, Ut = torch.tensor(x1, device="cuda:0"), torch.tensor(U0, device="cuda:0")
    solver = od.ILQR(dyn, obj, 10)
    lam = torch.zeros(12, 64, dtype=torch.float64, device="cuda:0")

    def run():
        X, A, Bm, st, _, _ = dyn.rollout(x1t, Ut)                       # od_rocket_rollout + od_rocket on every knot
        K, k, dV, bst = solver.backward(A, Bm, obj.expansion(X, Ut.double(), lam, 1.0), 1e-6)      # od_ilqr_backward
        J = obj.value(X, Ut.double())                                                               # od_quad_cost
        return X, A, Bm, K, k, dV, J
    _capture_replay(run, lambda: Ut.add_(0.02), lambda o: o)


@pytest.mark.parametrize("problem, dtype", [("rocket", torch.float32), ("rocket", torch.float64), ("cartpole", torch.float64)])
def test_ilqr_iteration_in_a_graph(gpu_lib, problem, dtype):
    """od_ilqr_iterate only enqueues kernels (no host synchronisation, no allocation, every decision on the device): ONE iteration
    recorded in a HIP graph and replayed n times gives bit for bit the cost history, trajectories and gains of n direct iterations"""
    import ilqr_checks as C
    import optimization_dynamics_amd as od
    B, T, n_it = 96, 20, 6
    if problem == "rocket":
        dyn, obj, x1, U0 = C.rocket_problem(gpu_lib, "cuda:0", B, T, dtype=dtype, seed=2)
    else:
        dyn, obj, x1, U0 = C.cartpole_problem(gpu_lib, "cuda:0", B, T, seed=2)
    x1t, Ut = torch.tensor(x1, device="cuda:0"), torch.tensor(U0, device="cuda:0")
    sol = od.ILQR(dyn, obj, T)
    direct = sol.device_solver(B, max_iter=n_it, obj_tol=0.0)
    direct.init(x1t, Ut)
    direct.iterate(n_it)
    want = direct.get(gains=True) + (direct.history(),)
    assert want[-1].shape[0] == n_it
    rec = sol.device_solver(B, max_iter=n_it, obj_tol=0.0)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        rec.init(x1t, Ut)
        rec.iterate(1)                           # warm-up on the capture stream
        rec.init(x1t, Ut)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            rec.iterate(1)
        # (capture does not execute: the solver still stands at its initial trajectory)
        for _ in range(n_it):
            g.replay()
        torch.cuda.synchronize()
        got = rec.get(gains=True) + (rec.history(),)
        torch.cuda.synchronize()
    for a, b in zip(got, want):
        assert torch.equal(a, b)


def test_ilqr_iteration_with_parameter_stage_in_a_graph(gpu_lib):
    """the parameter stage (examples/hopper.jl:52-99,234-266 through od_ilqr_set_parameter_stage: embedded first stage, generated
    constraint function, coupled terminal rows) adds kernels only: one recorded iteration replayed n times == n direct iterations,
    bit for bit, the optimised parameters (slot 0 of the trajectory) included"""
    import ilqr_checks as C
    import optimization_dynamics_amd as od
    B, n_it = 24, 7
    im, obj, x1, U0, x1v, T, opts = C.hopper_example_full(gpu_lib, "cuda:0", B)
    x1t, Ut = torch.tensor(x1, device="cuda:0"), torch.tensor(U0, device="cuda:0")
    sol = od.ILQR(im, obj, T)
    direct = sol.device_solver(B, max_iter=n_it, obj_tol=0.0)
    direct.init(x1t, Ut)
    direct.iterate(n_it)
    want = direct.get(gains=True) + (direct.history(),)
    assert want[-1].shape[0] == n_it
    assert (want[0][:, 0] - x1t).abs().max().item() > 1e-4, "the parameters did not move: the stage is not active"
    rec = sol.device_solver(B, max_iter=n_it, obj_tol=0.0)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        rec.init(x1t, Ut)
        rec.iterate(1)
        rec.init(x1t, Ut)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            rec.iterate(1)
        for _ in range(n_it):
            g.replay()
        torch.cuda.synchronize()
        got = rec.get(gains=True) + (rec.history(),)
        torch.cuda.synchronize()
    for a, b in zip(got, want):
        assert torch.equal(a, b)


def test_ilqr_iteration_with_gradient_bundle_in_a_graph(gpu_lib):
    """od_ilqr_set_gradient_bundle (examples/planar_push.jl with GB = true): the bundle's N + 1 steps per knot, the least-squares fit and
    the scatter into fx / fu are kernels on the solver's stream like the rest -- one recorded iteration replayed n times == n direct ones"""
    import ilqr_checks as C
    import optimization_dynamics_amd as od
    from optimization_dynamics_amd import gradient_bundle as gbm
    B, n_it = 16, 6
    im, obj, x1, U0, xT, T, opts = C.planar_push_example(gpu_lib, "cuda:0", "rotate", B)
    gb = gbm.GradientBundle(od.planarpush, N=50, eps=1.0e-4, seed=3)
    x1t, Ut = torch.tensor(x1, device="cuda:0"), torch.tensor(U0, device="cuda:0")
    sol = od.ILQR(im, obj, T, bundle=gb)
    direct = sol.device_solver(B, max_iter=n_it, obj_tol=0.0)
    direct.init(x1t, Ut)
    direct.iterate(n_it)
    want = direct.get(gains=True) + (direct.history(),)
    assert want[-1].shape[0] == n_it
    rec = sol.device_solver(B, max_iter=n_it, obj_tol=0.0)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        rec.init(x1t, Ut)
        rec.iterate(1)
        rec.init(x1t, Ut)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            rec.iterate(1)
        for _ in range(n_it):
            g.replay()
        torch.cuda.synchronize()
        got = rec.get(gains=True) + (rec.history(),)
        torch.cuda.synchronize()
    for a, b in zip(got, want):
        assert torch.equal(a, b)
