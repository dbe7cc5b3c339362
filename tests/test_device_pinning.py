"""A handle belongs to the HIP device that was current in od_create, and every entry point runs there whatever the calling
thread's current device is (include/od_mi355x.h "Devices"; csrc/od_capi.hip::OnDevice).  CPU tier: the host build of the product
sources against the emulated HIP runtime (tests/host_emu), which keeps a per-thread current device, records the device every
kernel launch saw and lets a test declare which device a stream belongs to.  GPU tier: what one visible device allows -- the
handle reports its device, a foreign stream is refused only when it is foreign."""
import ctypes as C

import numpy as np
import pytest
import torch

import parity_checks as P
import workloads as W


@pytest.fixture()
def two_devices(emu_lib):
    cd = C.CDLL(emu_lib.path)
    cd.od_emu_set_device_count(2)
    cd.od_emu_set_device(0)
    yield cd
    cd.od_emu_set_device(0)
    cd.od_emu_set_device_count(1)


def test_entry_points_run_on_the_device_of_their_handle(emu_lib, two_devices):
    emu = two_devices
    emu.od_emu_set_device(1)
    im = P.make_im("hopper", emu_lib, "cpu")                 # od_create under current device 1
    dev = C.c_int(-1)
    emu_lib.check(emu_lib.cdll.od_get_device(im._h, C.byref(dev)))
    assert dev.value == 1
    X, U = W.knots("hopper", 8, seed=3)
    ref = [t.clone() for t in im.step_grad(torch.tensor(X), torch.tensor(U))]
    assert emu.od_emu_last_launch_device() == 1
    # the caller moves on to device 0: the handle's launches still happen on device 1, and the caller's device is put back
    emu.od_emu_set_device(0)
    got = im.step_grad(torch.tensor(X), torch.tensor(U))
    assert emu.od_emu_last_launch_device() == 1, "the launch ran on the caller's current device, not on the handle's"
    assert emu.od_emu_get_device() == 0, "the caller's current device was not restored"
    for a, b in zip(ref, got):
        assert torch.equal(a, b)
    # ... for rollouts, the host-vector callbacks and the device-resident solver alike
    x1, Ur = W.hopper_rollout_inputs(4, 3, seed=1)
    im.rollout(torch.tensor(x1), torch.tensor(Ur))
    assert emu.od_emu_last_launch_device() == 1 and emu.od_emu_get_device() == 0
    d = np.zeros(8)
    from optimization_dynamics_amd import dynamics as dyn
    dyn.f(d, im, X[:, 0], U[:, 0], None)
    assert emu.od_emu_last_launch_device() == 1 and emu.od_emu_get_device() == 0
    al = (C.c_double * 2)(1.0, 0.5)
    s = C.c_void_p()
    emu_lib.check(emu_lib.cdll.od_ilqr_create(im._h, 4, 3, 2, al, None, C.byref(s)))
    assert emu.od_emu_get_device() == 0
    assert emu_lib.cdll.od_ilqr_destroy(s) == 0
    # a second handle made now lives on device 0; the two do not disturb each other
    im0 = P.make_im("hopper", emu_lib, "cpu")
    emu_lib.check(emu_lib.cdll.od_get_device(im0._h, C.byref(dev)))
    assert dev.value == 0
    im0.step(torch.tensor(X), torch.tensor(U))
    assert emu.od_emu_last_launch_device() == 0
    im.step(torch.tensor(X), torch.tensor(U))
    assert emu.od_emu_last_launch_device() == 1 and emu.od_emu_get_device() == 0


def test_stream_of_another_device_is_refused(emu_lib, two_devices):
    emu = two_devices
    emu.od_emu_set_device(1)
    im = P.make_im("cartpole_friction", emu_lib, "cpu")
    emu.od_emu_set_device(0)
    emu.od_emu_register_stream(C.c_void_p(0x1000), 0)
    emu.od_emu_register_stream(C.c_void_p(0x2000), 1)
    rc = emu_lib.cdll.od_set_stream(im._h, C.c_void_p(0x1000))
    assert rc == -5 and b"device 0" in emu_lib.cdll.od_last_error()          # OD_ERR_WRONG_DEVICE
    assert emu_lib.cdll.od_set_stream(im._h, C.c_void_p(0x2000)) == 0
    assert emu_lib.cdll.od_set_stream(im._h, None) == 0
    assert emu.od_emu_get_device() == 0


def test_destroying_the_handle_before_its_solver_is_safe(emu_lib):
    """a garbage collector gives no order between the two finalisers (julia/OptimizationDynamicsMI355X.jl, ILQRSolver): od_destroy
    releases what its live solvers hold and detaches them; the solver object then answers with an error code and can still be
    destroyed"""
    from optimization_dynamics_amd import _lib
    cd = emu_lib.cdll
    h = C.c_void_p()
    o = emu_lib.default_options("cartpole_friction")
    emu_lib.check(cd.od_create(emu_lib.model_id("cartpole_friction"), _lib.OD_F64, C.byref(o), 0.05, C.byref(h)))
    al = (C.c_double * 2)(1.0, 0.5)
    s1, s2 = C.c_void_p(), C.c_void_p()
    emu_lib.check(cd.od_ilqr_create(h, 4, 3, 2, al, None, C.byref(s1)))
    emu_lib.check(cd.od_ilqr_create(h, 2, 5, 2, al, None, C.byref(s2)))
    assert cd.od_ilqr_destroy(s1) == 0                       # the usual order for one of them
    assert cd.od_destroy(h) == 0                             # ... and the handle before the other
    info = _lib.IlqrInfo()
    assert cd.od_ilqr_get_info(s2, C.byref(info)) == -1 and b"destroyed" in cd.od_last_error()
    assert cd.od_ilqr_iterate(s2, 1) == -1
    assert cd.od_ilqr_destroy(s2) == 0


def test_destroy_is_best_effort_when_the_device_guard_fails(emu_lib, two_devices):
    """od_destroy / od_ilqr_destroy run from finalisers that ignore the return code: if switching to the handle's device fails (the
    device is gone) they report the error but the host objects go and the solver leaves the handle's list -- round 5 returned early,
    leaked everything and left the solver in h->solvers for a second release.  Emulated: a handle on device 1, then one device visible."""
    emu = two_devices
    for order in ("solver_first", "handle_first"):
        emu.od_emu_set_device_count(2)
        emu.od_emu_set_device(1)
        im = P.make_im("cartpole_friction", emu_lib, "cpu")
        dev = C.c_int(-1)
        assert emu_lib.cdll.od_get_device(im._h, C.byref(dev)) == 0 and dev.value == 1
        al = (C.c_double * 2)(1.0, 0.5)
        s = C.c_void_p()
        assert emu_lib.cdll.od_ilqr_create(im._h, 4, 3, 2, al, None, C.byref(s)) == 0
        emu.od_emu_set_device(0)
        emu.od_emu_set_device_count(1)                       # device 1 is gone: hipSetDevice(1) fails
        h, im._h = im._h, None                               # (ours to destroy now)
        if order == "solver_first":
            assert emu_lib.cdll.od_ilqr_destroy(s) == -3 and b"device" in emu_lib.cdll.od_last_error()
            assert emu_lib.cdll.od_destroy(h) == -3
        else:
            assert emu_lib.cdll.od_destroy(h) == -3 and b"device" in emu_lib.cdll.od_last_error()
            assert emu_lib.cdll.od_ilqr_init(s, None, None) == -1        # detached: answers, does not touch freed memory
            assert emu_lib.cdll.od_ilqr_destroy(s) == 0


def test_parameter_stage_excludes_goals_and_plain_terminal_rows(emu_lib):
    """with a parameter stage the Riccati pass starts from the stage's own terminal model; goal components and terminal
    rows of od_ilqr_set_constraints would be charged by the merit only -- refused in either order of the calls"""
    from optimization_dynamics_amd import _lib
    im = P.make_im("hopper", emu_lib, "cpu")
    al = (C.c_double * 2)(1.0, 0.5)
    n, m = 8, 2
    I8 = (C.c_double * 64)(*np.eye(8).ravel()); I2 = (C.c_double * 4)(*np.eye(2).ravel()); xr = (C.c_double * 8)()
    w = (C.c_double * 8)(*([0.1] * 8))
    gi = (C.c_int * 1)(0); gv = (C.c_double * 1)(1.0)
    Ct = (C.c_double * 8)(*([1.0] + [0.0] * 7)); dt = (C.c_double * 1)(0.5)
    ps = _lib.IlqrParameterStage(constraint=-1, n_p=0, p=None, w_theta=w, cost_const=0.0, nt=0, nt_ineq=0, Ct_x=None, Ct_theta=None, dt=None)
    cd = emu_lib.cdll
    # goals first, then the stage
    s = C.c_void_p(); assert cd.od_ilqr_create(im._h, 2, 3, 2, al, None, C.byref(s)) == 0
    assert cd.od_ilqr_set_objective(s, I8, I2, I8, xr, 1, gi, gv) == 0
    assert cd.od_ilqr_set_parameter_stage(s, C.byref(ps)) == -2 and b"parameter stage" in cd.od_last_error()
    assert cd.od_ilqr_set_objective(s, I8, I2, I8, xr, 0, None, None) == 0
    assert cd.od_ilqr_set_constraints(s, 0, 0, None, None, None, 1, 0, Ct, dt) == 0
    assert cd.od_ilqr_set_parameter_stage(s, C.byref(ps)) == -2
    assert cd.od_ilqr_set_constraints(s, 0, 0, None, None, None, 0, 0, None, None) == 0
    assert cd.od_ilqr_set_parameter_stage(s, C.byref(ps)) == 0
    # the stage first, then goals / terminal rows
    assert cd.od_ilqr_set_objective(s, I8, I2, I8, xr, 1, gi, gv) == -2
    assert cd.od_ilqr_set_constraints(s, 0, 0, None, None, None, 1, 0, Ct, dt) == -2
    assert cd.od_ilqr_set_objective(s, I8, I2, I8, xr, 0, None, None) == 0          # without goals: fine
    assert cd.od_ilqr_set_parameter_stage(s, None) == 0                              # stage removed: goals are back
    assert cd.od_ilqr_set_objective(s, I8, I2, I8, xr, 1, gi, gv) == 0
    assert cd.od_ilqr_destroy(s) == 0


def test_constraints_can_be_replaced_without_growth(emu_lib):
    """od_ilqr_set_constraints allocates its buffers once: replacing the constraints many times reuses them, and what is in force
    after each call is what that call passed"""
    cd = emu_lib.cdll
    im = P.make_im("cartpole_friction", emu_lib, "cpu")
    al = (C.c_double * 2)(1.0, 0.5)
    s = C.c_void_p()
    emu_lib.check(cd.od_ilqr_create(im._h, 3, 4, 2, al, None, C.byref(s)))
    dp = lambda a: np.ascontiguousarray(a, dtype=np.float64).ctypes.data_as(C.POINTER(C.c_double))
    Q = np.eye(4); R = np.eye(1)
    emu_lib.check(cd.od_ilqr_set_objective(s, dp(Q), dp(R), dp(Q), dp(np.zeros(4)), 0, None, None))
    for k in range(40):
        ns = 1 + k % 3
        Cs = np.zeros((ns, 4)); Ds = np.ones((ns, 1)); ds = np.full(ns, 2.0 + k)
        emu_lib.check(cd.od_ilqr_set_constraints(s, ns, ns, dp(Cs.T.copy()), dp(Ds.T.copy()), dp(ds), 1, 0, dp(np.array([[1.0, 0, 0, 0]]).T.copy()), dp(np.array([0.5]))))
    x1 = torch.zeros(4, 3, dtype=torch.float64); U0 = torch.full((1, 4, 3), 0.1, dtype=torch.float64)
    emu_lib.check(cd.od_ilqr_init(s, x1.data_ptr(), U0.data_ptr()))
    emu_lib.check(cd.od_ilqr_set_constraints(s, 0, 0, None, None, None, 0, 0, None, None))
    assert cd.od_ilqr_destroy(s) == 0


@pytest.mark.gpu
def test_handle_reports_its_device_gpu(gpu_lib):
    im = P.make_im("hopper", gpu_lib, "cuda:0")
    dev = C.c_int(-1)
    gpu_lib.check(gpu_lib.cdll.od_get_device(im._h, C.byref(dev)))
    assert dev.value == torch.cuda.current_device() == 0
    s = torch.cuda.Stream(device="cuda:0")
    assert gpu_lib.cdll.od_set_stream(im._h, C.c_void_p(s.cuda_stream)) == 0          # a stream of the handle's own device
    X, U = W.knots("hopper", 64, seed=3)
    with torch.cuda.stream(s):
        D, st, it = im.step(torch.tensor(X), torch.tensor(U))
    s.synchronize()
    assert ((st & 1) == 1).double().mean().item() > 0.95


def test_parameter_stage_argument_checks(emu_lib):
    """od_ilqr_set_parameter_stage / od_constraint_* / od_ilqr_get_trace / od_soc_project_full: wrong arguments are error codes"""
    from optimization_dynamics_amd import _lib, models, rocket as rk
    cd = emu_lib.cdll
    assert cd.od_num_constraints() >= 1 and cd.od_constraint_id(b"hopper_foot") == 0 and cd.od_constraint_id(b"no_such_rows") == -1
    assert cd.od_constraint_name(0) == b"hopper_foot" and cd.od_constraint_name(99) is None
    nc, nx, npar = C.c_int(), C.c_int(), C.c_int()
    assert cd.od_constraint_dims(0, C.byref(nc), C.byref(nx), C.byref(npar)) == 0 and (nc.value, nx.value, npar.value) == (8, 8, 8)
    assert cd.od_constraint_dims(99, None, None, None) == -1
    al = (C.c_double * 2)(1.0, 0.5)
    dp = lambda a: np.ascontiguousarray(a, dtype=np.float64).ctypes.data_as(C.POINTER(C.c_double))
    w8, p8 = np.full(8, 0.1), np.zeros(8)
    # a hopper solver: the constraint fits (8 variables); wrong parameter count, unknown id, too many rows are refused
    im = P.make_im("hopper", emu_lib, "cpu")
    s = C.c_void_p()
    emu_lib.check(cd.od_ilqr_create(im._h, 2, 4, 2, al, None, C.byref(s)))
    q = _lib.IlqrParameterStage()
    q.constraint, q.n_p, q.p, q.w_theta = 0, 8, dp(p8), dp(w8)
    assert cd.od_ilqr_set_parameter_stage(s, C.byref(q)) == 0
    q.n_p = 3
    assert cd.od_ilqr_set_parameter_stage(s, C.byref(q)) == -1 and b"8 parameters" in cd.od_last_error()
    q.n_p, q.constraint = 8, 7
    assert cd.od_ilqr_set_parameter_stage(s, C.byref(q)) == -1
    q.constraint, q.nt = 0, 17
    assert cd.od_ilqr_set_parameter_stage(s, C.byref(q)) == -1
    q.nt, q.w_theta = 0, None
    assert cd.od_ilqr_set_parameter_stage(s, C.byref(q)) == -1
    assert cd.od_ilqr_set_parameter_stage(s, None) == 0                      # removes the stage
    assert cd.od_ilqr_get_trace(s, None, None, None, 0) == 0                 # (no iterations yet: no rows)
    info0 = _lib.IlqrInfo()
    assert cd.od_ilqr_get_info(s, C.byref(info0)) == 0 and info0.iterations == 0 and info0.al_iterations == 0    # (state block zeroed at creation)
    assert cd.od_ilqr_get_trace(s, None, None, None, 5) == 0 and cd.od_ilqr_get_history(s, None, 0) == 0
    assert cd.od_ilqr_get_trace(None, None, None, None, 0) < 0
    assert cd.od_ilqr_destroy(s) == 0
    # a cartpole solver: theta has 4 entries, the hopper's rows act on 8
    imc = P.make_im("cartpole_friction", emu_lib, "cpu")
    emu_lib.check(cd.od_ilqr_create(imc._h, 2, 4, 2, al, None, C.byref(s)))
    q = _lib.IlqrParameterStage()
    q.constraint, q.n_p, q.p, q.w_theta = 0, 8, dp(p8), dp(w8)
    assert cd.od_ilqr_set_parameter_stage(s, C.byref(q)) == -1 and b"theta has 4" in cd.od_last_error()
    q.constraint, q.n_p, q.p = -1, 0, None
    assert cd.od_ilqr_set_parameter_stage(s, C.byref(q)) == 0                # no generated rows: cost on theta only
    assert cd.od_ilqr_destroy(s) == 0
    # a rocket solver has no configurations to optimise
    info = rk.RocketInfo(models.rocket, 12.5, 0.05, device="cpu", lib=emu_lib)
    emu_lib.check(cd.od_ilqr_create(info._h, 2, 4, 2, al, None, C.byref(s)))
    assert cd.od_ilqr_set_parameter_stage(s, C.byref(q)) == -2               # OD_ERR_UNSUPPORTED
    # ... and no simulator step to sample: the gradient bundle is for the mechanical models
    eta = np.zeros((15, 5))
    assert cd.od_ilqr_set_gradient_bundle(s, 5, eta.ctypes.data_as(C.c_void_p)) == -2
    assert cd.od_ilqr_set_gradient_bundle(s, 0, None) == 0                   # (removing what is not there is fine)
    assert cd.od_ilqr_destroy(s) == 0
    assert cd.od_ilqr_set_gradient_bundle(None, 5, eta.ctypes.data_as(C.c_void_p)) == -1
    emu_lib.check(cd.od_ilqr_create(imc._h, 2, 4, 2, al, None, C.byref(s)))
    eta = np.zeros((5, 8)); eta[0, :] = 1e-4
    assert cd.od_ilqr_set_gradient_bundle(s, 8, eta.ctypes.data_as(C.c_void_p)) == 0
    assert cd.od_ilqr_set_gradient_bundle(s, 4, eta.ctypes.data_as(C.c_void_p)) == 0      # fewer samples reuse the arrays
    assert cd.od_ilqr_set_gradient_bundle(s, 8, None) == 0                                # NULL eta: implicit gradients again
    assert cd.od_ilqr_destroy(s) == 0
    # od_soc_project_full: empty batch is a no-op, null outputs are refused, a mechanical handle is unsupported
    u = torch.zeros(3, 4, dtype=torch.float64); z = torch.zeros(10, 4, dtype=torch.float64)
    assert cd.od_soc_project_full(info._h, 0, u.data_ptr(), z.data_ptr(), None, None, None) == 0
    assert cd.od_soc_project_full(info._h, 4, u.data_ptr(), None, None, None, None) == -1
    assert cd.od_soc_project_full(im._h, 4, u.data_ptr(), z.data_ptr(), None, None, None) == -2
    assert cd.od_soc_project_full(info._h, 4, u.data_ptr(), z.data_ptr(), None, None, None) == 0 and torch.isfinite(z).all()
