"""CPU tier: the product's device code + C ABI compiled for the host (tests/host_emu, a test
harness: stand-in hip_runtime.h that runs each lane as a loop iteration) against the oracle.
Checks the solver logic (static sparse KKT elimination vs the oracle's dense partial-pivot LU,
fused kappa_eval/kappa_grad prefix loop) and all host-side plumbing without a GPU."""
import numpy as np
import pytest
import torch

import parity_checks as P
import workloads as W

MECH = ["acrobot_impact", "acrobot_nominal", "cartpole_friction", "cartpole_frictionless", "hopper", "planar_push"]


@pytest.mark.parametrize("name", MECH)
def test_step_grad_parity(oracle, emu_lib, name):
    P.check_step_grad(oracle, emu_lib, "cpu", name, 512 if name != "planar_push" else 256)


@pytest.mark.parametrize("name", ["acrobot_impact", "hopper"])
def test_step_and_compact_outputs_consistent(emu_lib, name):
    P.check_step_only_and_compact(emu_lib, "cpu", name, 64)


@pytest.mark.parametrize("name", ["cartpole_friction", "hopper", "planar_push"])
def test_batch_major_layout_identical(emu_lib, name):
    P.check_layouts(emu_lib, "cpu", name, 96)


def test_hopper_rollout_parity(oracle, emu_lib):
    P.check_rollout(oracle, emu_lib, "cpu", 48, 30)


def test_cartpole_plumbing_config_single_rollout(oracle, emu_lib):
    """BASELINE config 1: cartpole with joint friction, T=51, single rollout, x1 = 0, u_1 = -1.5
    (examples/cartpole.jl:15-21,41-46,78)."""
    im = P.make_im("cartpole_friction", emu_lib, "cpu")
    T = 50
    U = np.zeros((1, T, 1)); U[0, 0, 0] = -1.5
    X, A, Bm, st, it, _ = im.rollout(torch.zeros(4, 1, dtype=torch.float64), torch.tensor(U))
    Xo, Ao, Bo, bad = oracle.rollout(P.make_sim(oracle, "cartpole_friction"), np.zeros((4, 1)), U)
    assert bad == 0 and ((st.numpy() & 3) == 3).all()
    assert np.abs(X.numpy() - Xo).max() < 1e-9
    assert W.grad_rel_err(A.numpy().reshape(16, T), Ao.reshape(16, T)).max() < 1e-6


@pytest.mark.parametrize("name,N", [("planar_push", 64), ("hopper", 50)])
def test_gradient_bundle(oracle, emu_lib, name, N):
    P.check_bundle(oracle, emu_lib, "cpu", name, 6, N)


def test_ls_known_answer_on_device_kernel(emu_lib):
    P.check_ls_kat(emu_lib, "cpu")


def test_rocket_f64(oracle, emu_lib):
    P.check_rocket(oracle, emu_lib, "cpu", 32)


def test_rocket_sweep_emulated(oracle, emu_lib):
    """the GPU tier's rocket sweep (test_gpu_parity_sweep.py::test_rocket_parity_sweep) at 2048 knots on the host build of the same sources"""
    for dtype in (torch.float64, torch.float32):
        P.check_rocket_sweep(oracle, emu_lib, "cpu", 2048, 202, dtype)


def test_rocket_f32(oracle, emu_lib):
    P.check_rocket(oracle, emu_lib, "cpu", 16, dtype=torch.float32)


@pytest.mark.parametrize("name", ["acrobot_impact", "cartpole_friction", "hopper", "planar_push", "acrobot_nominal"])
def test_step_full_contact_forces(oracle, emu_lib, name):
    P.check_step_full(oracle, emu_lib, "cpu", name, 48)


def test_raw_interior_point_solve(oracle, emu_lib):
    P.check_ip_solve(oracle, emu_lib, "cpu")


def test_raw_interior_point_solve_f32(oracle, emu_lib):
    P.check_ip_solve(oracle, emu_lib, "cpu", dtype=torch.float32)


def test_live_setters(emu_lib):
    P.check_live_setters(emu_lib, "cpu")


def test_soc_projection(oracle, emu_lib):
    P.check_soc_projection(oracle, emu_lib, "cpu", 48)


@pytest.mark.parametrize("name", ["acrobot_impact", "cartpole_friction"])
def test_reference_signature_callbacks(oracle, emu_lib, name):
    P.check_scalar_callbacks(oracle, emu_lib, "cpu", name)


def test_finite_undercut_runs_two_passes(oracle, emu_lib):
    """with cones and a finite undercut the centering floor depends on kappa_tol: the eval and grad
    simulators are no longer prefix-related and the library must run them separately"""
    name = "acrobot_impact"
    X, U = W.knots(name, 64, seed=61)
    im = P.make_im(name, emu_lib, "cpu", options=dict(undercut=5.0))
    D, DX, DU, st, it = im.step_grad(torch.tensor(X), torch.tensor(U))
    Do, DXo, DUo, bad = oracle.step_grad_batch(P.make_sim(oracle, name, undercut=5.0), X, U)
    assert np.abs(D.numpy() - Do).max() < 1e-6
    rel = W.grad_rel_err(np.concatenate([DX.numpy(), DU.numpy()], 1), np.concatenate([DXo, DUo], 1))
    assert np.median(rel) < 1e-9 and (rel < 1e-4).mean() > 0.98
    # status / iterations merge the two solves: bit 1 from the eval solve, bit 2 from the grad solve
    assert ((st.numpy() & 3) == 3).mean() > 0.98 and (it.numpy()[1] > 0).all() and (it.numpy()[1] <= it.numpy()[0]).mean() > 0.9


def test_friction_vector_is_live(emu_lib):
    """cartpole_friction.friction .= [...] after construction must take effect (examples/cartpole.jl:21)"""
    from optimization_dynamics_amd import models
    im = P.make_im("cartpole_friction", emu_lib, "cpu")
    X, U = W.knots("cartpole_friction", 8, seed=71)
    D1, _, _ = im.step(torch.tensor(X), torch.tensor(U))
    models.cartpole_friction.friction[:] = [0.0, 0.0]
    D2, _, _ = im.step(torch.tensor(X), torch.tensor(U))
    models.cartpole_friction.friction[:] = [0.35, 0.35]
    assert not torch.equal(D1, D2)


def test_empty_batch_and_bad_arguments(emu_lib):
    import ctypes as C
    im = P.make_im("hopper", emu_lib, "cpu")
    assert emu_lib.cdll.od_step(im._h, 0, 0, 0, 0, 0, 0) == 0          # empty batch is a no-op
    assert emu_lib.cdll.od_step(im._h, 4, 0, 0, 0, 0, 0) == -1         # null input
    assert b"null input" in emu_lib.cdll.od_last_error()
    assert emu_lib.cdll.od_rocket(im._h, 4, 0, 1, 1, 0, 0, 0, 0, 0) == -2   # wrong model
    fr = (C.c_double * 3)(0.1, 0.2, 0.3)
    assert emu_lib.cdll.od_set_friction(im._h, fr, 3) == -1


@pytest.mark.parametrize("B", [1, 3, 63, 65, 257])
def test_ragged_batch_sizes_and_launch_configs(oracle, emu_lib, B):
    """batches that do not fill a wavefront / workgroup, under every launch mapping, give identical results"""
    name = "acrobot_impact"
    X, U = W.knots(name, B, seed=81)
    im = P.make_im(name, emu_lib, "cpu")
    im.set_cooperative(1)          # the lane-per-problem kernels: bitwise identical under every mapping (tests/test_coop.py has the other)
    ref = None
    for ppw, wpb in [(0, 0), (1, 1), (4, 4), (16, 4), (64, 1), (64, 4)]:
        im.set_launch_config(ppw, wpb)
        D, DX, DU, st, it = im.step_grad(torch.tensor(X), torch.tensor(U))
        cur = (D.clone(), DX.clone(), DU.clone(), st.clone(), it.clone())
        if ref is None:
            ref = cur
        else:
            assert all(torch.equal(a, b) for a, b in zip(ref, cur))
    Do, DXo, DUo, bad = oracle.step_grad_batch(P.make_sim(oracle, name), X, U)
    ok = P.comparable_states(oracle, name, X, U, ref[0].numpy(), Do, (ref[3].numpy() & 3) == 3)
    assert (not ok.any()) or np.abs(ref[0].numpy() - Do)[:, ok].max() < 1e-6


def test_single_step_rollout_and_horizon_one(oracle, emu_lib):
    im = P.make_im("hopper", emu_lib, "cpu")
    x1, U = W.hopper_rollout_inputs(5, 1, seed=9, u_sigma=0.2)
    X, A, Bm, st, it, _ = im.rollout(torch.tensor(x1), torch.tensor(U))
    D, DX, DU, s1, i1 = im.step_grad(torch.tensor(x1), torch.tensor(U[:, 0]))
    assert torch.equal(X[:, 1], D) and torch.equal(A[:, :, 0], DX) and torch.equal(Bm[:, :, 0], DU)
    Xs = im.rollout(torch.tensor(x1), torch.tensor(U), grads=False)[0]
    assert torch.equal(Xs, X)


def test_nonconvergence_is_reported_not_raised(emu_lib):
    """max_iter = 2: the solve cannot converge; like the reference (Bool status, result still copied out)
    the call succeeds and the status bits say so"""
    im = P.make_im("hopper", emu_lib, "cpu", options=dict(max_iter=2))
    X, U = W.knots("hopper", 16, seed=5)
    D, DX, DU, st, it = im.step_grad(torch.tensor(X), torch.tensor(U))
    assert ((st.numpy() & 3) == 0).all() and (it.numpy() == 2).all()
    assert torch.isfinite(D).all()


def test_launch_config_validation(emu_lib):
    im = P.make_im("hopper", emu_lib, "cpu")
    assert emu_lib.cdll.od_set_launch_config(im._h, 3, 0) == -1
    assert emu_lib.cdll.od_set_launch_config(im._h, 128, 0) == -1
    assert emu_lib.cdll.od_set_launch_config(im._h, 16, 2) == -1
    assert emu_lib.cdll.od_set_launch_config(im._h, 16, 4) == 0


def test_device_sincos_accuracy(emu_lib):
    """od_math.h::od_sincos (the device replacement for sin/cos of joint angles) against a 200-bit
    reference: <= 1.5 ulp over the range the path sees, graceful beyond."""
    import ctypes
    import mpmath as mp
    mp.mp.prec = 200
    so = emu_lib.cdll
    rng = np.random.default_rng(11)
    xs = np.concatenate([rng.uniform(-4, 4, 3000), rng.uniform(-1e3, 1e3, 3000), rng.uniform(-1e6, 1e6, 2000),
                         np.arange(-40, 41) * (np.pi / 4), np.arange(-40, 41) * (np.pi / 2) + 1e-9,
                         [0.0, 1e-300, -1e-8, 1e-8]])
    s = np.empty_like(xs)
    c = np.empty_like(xs)
    P = ctypes.POINTER(ctypes.c_double)
    so.od_emu_sincos(xs.ctypes.data_as(P), ctypes.c_long(xs.size), s.ctypes.data_as(P), c.ctypes.data_as(P))
    worst = 0.0
    for x, si, ci in zip(xs, s, c):
        for got, ref in ((si, mp.sin(mp.mpf(float(x)))), (ci, mp.cos(mp.mpf(float(x))))):
            ulp = np.spacing(abs(float(ref))) if ref != 0 else 5e-324
            worst = max(worst, float(abs(mp.mpf(float(got)) - ref) / ulp))
    assert worst <= 1.5, worst
    # far outside the working range: bounded, finite garbage is acceptable, NaN only for non-finite input
    big = np.array([1e12, -3e15, 1e300])
    sb = np.empty(3); cb = np.empty(3)
    so.od_emu_sincos(big.ctypes.data_as(P), ctypes.c_long(3), sb.ctypes.data_as(P), cb.ctypes.data_as(P))
    assert np.all(np.isfinite(sb[:2])) and np.all(np.abs(sb[:2]) <= 1.0 + 1e-6)


def test_tail_lu_against_numpy(emu_lib):
    """od_math.h::od_lu_factor / od_lu_solve (the runtime-pivoted dense tail): random, badly scaled and
    permutation-heavy 6x6 systems against numpy; an all-zero column is reported and leaves finite output."""
    import ctypes
    P = ctypes.POINTER(ctypes.c_double)
    fn = emu_lib.cdll.od_emu_lu6
    fn.restype = ctypes.c_int
    rng = np.random.default_rng(3)
    x = np.empty(6)
    for trial in range(200):
        A = rng.normal(size=(6, 6))
        if trial % 3 == 1:
            A *= 10.0 ** rng.integers(-20, 20, size=(1, 6))         # column scales as in the KKT tail
        if trial % 3 == 2:
            A = A[rng.permutation(6)] + np.diag(rng.normal(size=6) * 1e-8)
        b = rng.normal(size=6)
        Af = np.asfortranarray(A)
        ok = fn(Af.ctypes.data_as(P), b.ctypes.data_as(P), x.ctypes.data_as(P))
        assert ok == 1
        ref = np.linalg.solve(A, b)
        assert np.allclose(x, ref, rtol=1e-9 * max(1.0, np.linalg.cond(A / np.abs(A).max(0)) / 1e4), atol=0), trial
    A = rng.normal(size=(6, 6)); A[:, 0] = 0.0
    Af = np.asfortranarray(A)
    ok = fn(Af.ctypes.data_as(P), b.ctypes.data_as(P), x.ctypes.data_as(P))
    assert ok == 0 and np.all(np.isfinite(x)) and x[0] == 0.0


def test_rollout_with_finite_undercut(oracle, emu_lib):
    P.check_rollout_finite_undercut(oracle, emu_lib, "cpu")


@pytest.mark.parametrize("name", ["acrobot_impact", "hopper", "cartpole_friction", "planar_push"])
def test_lane_cooperation_in_lockstep_rows(oracle, emu_lib, name):
    """The lane cooperation of the lane-per-problem kernels (od_solver.h: cone / orthant step-length tests shared out
    over the 16/ppw copies of a problem, parallel line search; DPP row rotations on the device) on the CPU: the host
    build runs the 16 threads of a row as 16 host threads that meet at every rotation (tests/host_emu/hip/hip_runtime.h),
    and every mapping must give what the sequential build gives -- states, status and iteration counts bit for bit."""
    B = 21 if name == "planar_push" else 45                          # ragged: the last row carries fewer problems
    X, U = W.knots(name, 4099 if name == "acrobot_impact" else 8 * B, seed=17)
    if name == "acrobot_impact":                                       # keep the hardest knots (jams: cooperative backtracking)
        im0 = P.make_im(name, emu_lib, "cpu")
        it0 = im0.step_grad(torch.tensor(X), torch.tensor(U))[4][0].numpy()
        order = np.argsort(-it0)
        X, U = np.ascontiguousarray(X[:, order[:B]]), np.ascontiguousarray(U[:, order[:B]])
    else:
        X, U = np.ascontiguousarray(X[:, :B]), np.ascontiguousarray(U[:, :B])
    im = P.make_im(name, emu_lib, "cpu")
    im.set_cooperative(1)
    im.set_launch_config(16, 4)
    ref = [t.clone() for t in im.step_grad(torch.tensor(X), torch.tensor(U))]
    emu_lib.cdll.od_emu_set_lockstep(1)
    try:
        for ppw, wpb in [(1, 4), (2, 4), (4, 1)]:
            im.set_launch_config(ppw, wpb)
            cur = im.step_grad(torch.tensor(X), torch.tensor(U))
            assert torch.equal(ref[3], cur[3]) and torch.equal(ref[4], cur[4]), ppw
            assert torch.equal(ref[0], cur[0]), ppw
            # gradients: bit for bit where the recorded iterates are (the same gradient pass); where the cone variables of the
            # iterate differ in their last bits between the cooperating and the sequential mapping (hopper, one knot of seed set
            # 15: 7e-8 relative), both within what the conditioning of THAT knot explains of the exact gradient (binary128
            # arbiter at this mapping's iterates) -- as tests/test_gpu_parity.py::test_launch_mappings_agree does on the device
            if not (torch.equal(ref[1], cur[1]) and torch.equal(ref[2], cur[2])):
                nq = X.shape[0] // 2
                G = lambda c: np.concatenate([c[1].numpy()[nq:], c[2].numpy()[nq:]], 1)
                P.assert_grad_conditioned(oracle, im, name, X, U, G(cur), G(ref), ((cur[3] & 3) == 3).numpy(), "mapping %d" % ppw)
    finally:
        emu_lib.cdll.od_emu_set_lockstep(0)


@pytest.mark.parametrize("name", MECH)
def test_solutions_zero_the_hand_written_residuals(oracle, emu_lib, name):
    """oracle/models_np.py (src/models/<model>/model.jl restated by hand) at the solutions the product's kernels return (host build)"""
    P.check_solutions_against_hand_written_residuals(oracle, emu_lib, "cpu", name, B=256)


def test_rocket_solutions_zero_the_hand_written_residuals(oracle, emu_lib):
    P.check_rocket_solutions_against_hand_written_residuals(oracle, emu_lib, "cpu", B=256)


def test_every_knot_of_hopper_rollouts_host_build(oracle, emu_lib):
    """the CPU twin of test_gpu_parity.py::test_headline_rollout_every_knot_passes_the_stopping_test: 1024 rollouts x 100 knots on the host
    build -- 1e-6 between the rollout kernel and the independent-knot kernel wherever both stop at the same iteration, stopping ties
    identified by their iteration counts, every solution under the oracle's residual"""
    P.check_every_knot(oracle, emu_lib, "cpu", 1024, 100)

