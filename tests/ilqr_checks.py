"""iLQR checks shared by the CPU (host-emulation) and GPU tiers."""
import numpy as np
import torch

import parity_checks as P
from optimization_dynamics_amd import ilqr as IL, models


def cartpole_problem(lib, device, B, T, seed=0):
    """cartpole with joint friction (examples/cartpole.jl:15-21 model/h/friction): move the cart 0.3 m with
    the pole hanging, quadratic costs -- a task iLQR solves in a handful of iterations"""
    im = P.make_im("cartpole_friction", lib, device)
    n, m = 4, 1
    Q = np.diag([0.0, 0.1, 0.0, 0.1]); R = np.diag([0.01]); QT = np.diag([1.0, 1.0, 1.0, 1.0]) * 100
    goal = np.array([0.3, 0.0, 0.3, 0.0])
    obj = IL.QuadraticObjective(Q, R, QT, x_ref=goal, goal_idx=[0, 1, 2, 3], goal=goal, device=device)
    rng = np.random.default_rng(seed)
    x1 = np.zeros((n, B)) + rng.normal(0, 0.01, (n, B)); x1[2:] = x1[:2]
    # start above the static-friction threshold (mu (mp+mc) g h = 0.2): a stuck cart has dx/du = 0
    U0 = 0.4 + 1e-2 * rng.normal(size=(m, T, B))
    return im, obj, x1, U0


def check_backward_and_forward(oracle, lib, device):
    from oracle import ilqr_np          # numpy checker (test infrastructure)
    B, T = 6, 12
    im, obj, x1, U0 = cartpole_problem(lib, device, B, T)
    solver = IL.ILQR(im, obj, T)
    x1t, Ut = torch.tensor(x1, device=device), torch.tensor(U0, device=device)
    X, A, Bm, st = solver.linearize(x1t, Ut)
    lam = torch.zeros(4, B, dtype=torch.float64, device=device)
    quad = obj.expansion(X, Ut, lam, 1.0)
    K, k, dV, bst = solver.backward(A, Bm, quad, 1e-6)
    assert (bst == 1).all()
    lxx, luu, lux, lx, lu, VxxT, VxT = [q.cpu().numpy() for q in quad]
    An, Bn = A.cpu().numpy(), Bm.cpu().numpy()
    n, m = 4, 1
    for b in range(B):
        Kr, kr, dVr = ilqr_np.backward(
            np.moveaxis(An[:, :, :, b], 2, 0), np.moveaxis(Bn[:, :, :, b], 2, 0),
            np.moveaxis(lxx[:, :, b].reshape(n, n, T, order="F"), 2, 0), np.moveaxis(luu[:, :, b].reshape(m, m, T, order="F"), 2, 0),
            np.moveaxis(lux[:, :, b].reshape(m, n, T, order="F"), 2, 0), lx[:, :, b].T, lu[:, :, b].T,
            VxxT[:, b].reshape(n, n, order="F"), VxT[:, b], 1e-6)
        Kd = np.moveaxis(K.cpu().numpy()[:, :, b].reshape(m, n, T, order="F"), 2, 0)
        assert np.abs(Kd - Kr).max() < 1e-8 * max(1.0, np.abs(Kr).max())
        assert np.abs(k.cpu().numpy()[:, :, b].T - kr).max() < 1e-8 * max(1.0, np.abs(kr).max())
        assert np.abs(dV.cpu().numpy()[:, b] - dVr).max() < 1e-8 * max(1.0, np.abs(dVr).max())
    # forward pass: alpha = 0 candidate reproduces the nominal, every candidate equals an open-loop
    # rollout of the controls it reports
    solver.alphas = torch.tensor([1.0, 0.25, 0.0], dtype=torch.float64, device=device)
    Xc, Uc, cst = solver.forward(x1t, X, Ut, K, k)
    na = 3
    assert (Xc[:, :, 2 * B:] - X).abs().max().item() < 1e-12 and (Uc[:, :, 2 * B:] - Ut).abs().max().item() < 1e-12
    Xo = im.rollout(x1t.repeat(1, na), Uc, grads=False)[0]
    assert (Xo - Xc).abs().max().item() < 1e-10
    # policy law
    t, a, b = 3, 1, 2
    Kt = K[:, t, b].view(n, m).T
    u_expect = Ut[:, t, b] + 0.25 * k[:, t, b] + Kt @ (Xc[:, t, a * B + b] - X[:, t, b])
    assert (u_expect - Uc[:, t, a * B + b]).abs().max().item() < 1e-12


def check_backward_trajectories_per_workgroup(lib, device, sizes=(2304, 4608, 9001), T=6):
    """od_ilqr_backward gives a workgroup 2 / 4 / 8 consecutive trajectories once the batch keeps 1024 workgroups busy (csrc/od_capi.hip:
    k_ilqr_backward_tb): gains, feed-forward terms, expected decrease and status identical to the one-trajectory-per-workgroup kernel
    (the same trajectories in batches of 500), ragged last workgroup included"""
    for B in sizes:
        im, obj, x1, U0 = cartpole_problem(lib, device, B, T, seed=3)
        solver = IL.ILQR(im, obj, T)
        x1t, Ut = torch.tensor(x1, device=device), torch.tensor(U0, device=device)
        X, A, Bm, st = solver.linearize(x1t, Ut)
        lam = torch.zeros(4, B, dtype=torch.float64, device=device)
        quad = obj.expansion(X, Ut, lam, 1.0)
        K, k, dV, bst = solver.backward(A, Bm, quad, 1e-6)
        assert (bst == 1).all() and torch.isfinite(K).all()
        for b0 in range(0, B, 500):
            sl = slice(b0, min(B, b0 + 500))
            Ks, ks, dVs, bs = solver.backward(A[..., sl], Bm[..., sl], tuple(q[..., sl].contiguous() for q in quad), 1e-6)
            assert torch.equal(Ks, K[..., sl]) and torch.equal(ks, k[..., sl]) and torch.equal(dVs, dV[:, sl]) and torch.equal(bs, bst[sl]), (B, b0)


def check_backward_sizes(lib, device, sizes=((12, 3), (8, 2), (4, 1), (10, 2), (6, 2), (16, 12)), batches=(5, 2304, 4608), T=5):
    """od_ilqr_backward on random well-posed data through the C ABI, every (n, m) instantiation of k_ilqr_backward_tb (the five models'
    sizes are template parameters, anything else takes the run-time form) at 1 / 2 / 4 trajectories per workgroup, against the numpy
    Riccati recursion (oracle/ilqr_np.py) on a sample of trajectories; ragged last workgroup included"""
    from oracle import ilqr_np
    from optimization_dynamics_amd.dynamics import _ptr
    im = P_make_im(lib, device)
    for n, m in sizes:
        for B in batches:
            rng = np.random.default_rng(1000 * n + m + B)
            A = np.eye(n)[:, :, None, None] + 0.1 * rng.normal(size=(n, n, T, B))
            Bm = rng.normal(size=(n, m, T, B))
            def spd(k, shape):
                G = rng.normal(size=(k, k) + shape)
                return np.einsum("ij...,kj...->ik...", G, G) + np.eye(k).reshape((k, k) + (1,) * len(shape))
            lxx, luu, Vxx = spd(n, (T, B)), spd(m, (T, B)), spd(n, (B,))
            lux = 0.1 * rng.normal(size=(m, n, T, B)); lx = rng.normal(size=(n, T, B)); lu = rng.normal(size=(m, T, B)); Vx = rng.normal(size=(n, B))
            col = lambda M: torch.tensor(np.ascontiguousarray(M.transpose((1, 0) + tuple(range(2, M.ndim))).reshape((-1,) + M.shape[2:])), device=device)
            dev = lambda M: torch.tensor(np.ascontiguousarray(M), device=device)
            K = torch.empty(m * n, T, B, dtype=torch.float64, device=device); k = torch.empty(m, T, B, dtype=torch.float64, device=device)
            dV = torch.empty(2, B, dtype=torch.float64, device=device); st = torch.empty(B, dtype=torch.int32, device=device)
            args = [col(A), col(Bm), col(lxx), col(luu), col(lux), dev(lx), dev(lu), col(Vxx), dev(Vx)]
            im._use_current_stream()
            # the workgroup kernels (matrices in LDS) first, then one trajectory per 16-lane DPP row where m <= 4 (mode 2), then the
            # default (mode 0): the matrix-core kernel for (12, 3) -- 4 / 8 / 16 trajectories per workgroup by batch size --, the row kernel else
            im.set_cooperative(1)
            lib.check(lib.cdll.od_ilqr_backward(im._h, B, T, n, m, *[_ptr(a) for a in args], 1e-6, _ptr(K), _ptr(k), _ptr(dV), _ptr(st)))
            assert (st == 1).all(), (n, m, B)
            Kl, kl, dVl = K.clone(), k.clone(), dV.clone()
            rel = lambda x, y: ((x - y).abs().max() / y.abs().max().clamp(min=1.0)).item()
            for mode in (2, 0):
                im.set_cooperative(mode)
                K.zero_(); k.zero_(); dV.zero_(); st.zero_()
                lib.check(lib.cdll.od_ilqr_backward(im._h, B, T, n, m, *[_ptr(a) for a in args], 1e-6, _ptr(K), _ptr(k), _ptr(dV), _ptr(st)))
                assert (st == 1).all(), (n, m, B, mode)
                assert rel(K, Kl) < 1e-10 and rel(k, kl) < 1e-10 and rel(dV, dVl) < 1e-10, (n, m, B, mode, rel(K, Kl), rel(k, kl), rel(dV, dVl))
            Kh, kh, dVh = K.cpu().numpy(), k.cpu().numpy(), dV.cpu().numpy()
            for b in sorted({0, 1, B // 2, B - 2, B - 1}):
                mv = lambda M: np.moveaxis(M[..., b], -1, 0)        # (r, c, T) -> (T, r, c)
                Ko, ko, dVo = ilqr_np.backward(mv(A), mv(Bm), mv(lxx), mv(luu), mv(lux), lx[:, :, b].T, lu[:, :, b].T, Vxx[:, :, b], Vx[:, b], 1e-6)
                Kd = Kh[:, :, b].reshape(n, m, T).transpose(2, 1, 0)   # stored col-major m x n per knot
                sc = max(1.0, np.abs(Ko).max())
                assert np.abs(Kd - Ko).max() < 1e-8 * sc and np.abs(kh[:, :, b].T - ko).max() < 1e-8 * sc, (n, m, B, b)
                assert np.abs(dVh[:, b] - dVo).max() < 1e-8 * max(1.0, np.abs(dVo).max())


def P_make_im(lib, device):
    import parity_checks as P
    return P.make_im("cartpole_friction", lib, device)


def check_solver_decreases_cost(lib, device, B=8, T=25):
    im, obj, x1, U0 = cartpole_problem(lib, device, B, T, seed=1)
    solver = IL.ILQR(im, obj, T)
    x1t, Ut = torch.tensor(x1, device=device), torch.tensor(U0, device=device)
    X0 = solver.linearize(x1t, Ut)[0]
    J0 = obj.value(X0, Ut)
    X, U, J, hist = solver.solve(x1t, Ut, max_iter=15, max_al_iter=1)
    Jf = obj.value(X, U)
    assert torch.isfinite(Jf).all()
    assert (Jf <= J0 + 1e-9).all() and (Jf < 0.2 * J0).float().mean().item() > 0.7
    # the returned trajectory is dynamically consistent with its controls
    Xr = im.rollout(x1t, U, grads=False)[0]
    assert (Xr - X).abs().max().item() < 1e-10
    return J0, Jf


def check_quad_cost(lib, device):
    """od_quad_cost (one pass over X and U) against the formula in torch, double and single precision inputs, dense Q / R / QT"""
    import parity_checks as P
    im = P.make_im("cartpole_friction", lib, device)
    for n, m, T, Pn in ((4, 1, 7, 33), (12, 3, 5, 1000), (16, 12, 3, 65)):
        rng = np.random.default_rng(n + m)
        sym = lambda k: (lambda G: G @ G.T + np.eye(k))(rng.normal(size=(k, k)))
        obj = IL.QuadraticObjective(sym(n), sym(m), sym(n), rng.normal(size=n), device=device)
        X = torch.tensor(rng.normal(size=(n, T + 1, Pn)), device=device); U = torch.tensor(rng.normal(size=(m, T, Pn)), device=device)
        ref = obj.value(X, U)                                   # (no handle: the torch formula)
        got = obj.value(X, U, im=im)
        assert ((got - ref).abs() <= 1e-12 * ref.abs().clamp(min=1.0)).all()
        got32 = obj.value(X.float(), U.float(), im=im)
        ref32 = IL.QuadraticObjective(obj.Q.cpu().numpy(), obj.R.cpu().numpy(), obj.QT.cpu().numpy(), obj.x_ref.cpu().numpy(), device=device).value(X.float().double(), U.float().double())
        assert ((got32 - ref32).abs() <= 1e-12 * ref32.abs().clamp(min=1.0)).all()


def check_reused_forward_states(lib, device, B=8, T=25):
    """the iteration linearises on the states its forward pass computed for the accepted candidate; rolling the accepted controls
    out a second time (reuse_forward_states=False) must give the same optimisation: costs equal to rounding, iteration by iteration"""
    im, obj, x1, U0 = cartpole_problem(lib, device, B, T, seed=1)
    x1t, Ut = torch.tensor(x1, device=device), torch.tensor(U0, device=device)
    a = IL.ILQR(im, obj, T).solve_stepwise(x1t, Ut, max_iter=10, max_al_iter=1)
    b = IL.ILQR(im, obj, T).solve_stepwise(x1t, Ut, max_iter=10, max_al_iter=1, reuse_forward_states=False)
    assert len(a[3]) == len(b[3])
    for ja, jb in zip(a[3], b[3]):
        assert ((ja - jb).abs() <= 1e-9 * jb.abs().clamp(min=1.0)).all()
    assert (a[0] - b[0]).abs().max().item() < 1e-8 and (a[1] - b[1]).abs().max().item() < 1e-8


def check_one_bad_trajectory_does_not_hurt_the_batch(lib, device, B=4, T=20):
    """a trajectory whose linearisation is not finite (NaN control: a failed contact solve looks the same) keeps its backward
    pass from factorising at any regularisation; the others must converge exactly as they do without it"""
    im, obj, x1, U0 = cartpole_problem(lib, device, B, T, seed=1)
    x1t = torch.tensor(x1, device=device)
    good = IL.ILQR(im, obj, T).solve(x1t, torch.tensor(U0, device=device), max_iter=8, max_al_iter=1)
    U0b = U0.copy(); U0b[:, 3, 1] = np.nan
    bad = IL.ILQR(im, obj, T).solve(x1t, torch.tensor(U0b, device=device), max_iter=8, max_al_iter=1)
    keep = [0, 2, 3]
    assert torch.isfinite(bad[2][keep]).all()
    assert torch.equal(bad[1][:, :, keep], good[1][:, :, keep]) and torch.equal(bad[2][keep], good[2][keep])


def check_backward_retry(lib, device, B=24, T=20, dtype=torch.float64):
    """a control cost that is indefinite until Quu is regularised by ~0.2: every trajectory's Riccati pass fails at the initial
    regularisation and repeats ITS recursion at 10 x reg inside the kernel (od_ilqr_solver.inc) until it factorises -- the matrix-core
    kernel (whole workgroup repeats, mode 0) against the one-trajectory-per-16-lanes kernel (mode 2): same regularisation reached, same
    gains to rounding; trajectory 1 carries a NaN control and must end with zero gains in both"""
    dyn, obj0, x1, U0 = rocket_problem(lib, device, B, T, dtype=dtype, seed=5)
    goal = np.zeros(12); goal[2] = 2.0
    obj = IL.QuadraticObjective(np.diag([1e-2] * 12), np.diag([-0.2, 1e-2, 1e-3]), np.diag([10.0] * 12), x_ref=goal, device=device)
    U0 = U0.copy(); U0[:, 3, 1] = np.nan
    x1t, Ut = torch.tensor(x1, device=device), torch.tensor(U0, device=device)
    out = {}
    for mode in (2, 0):
        dyn.info.set_cooperative(mode) if hasattr(dyn, "info") and hasattr(dyn.info, "set_cooperative") else lib.check(lib.cdll.od_set_cooperative(dyn._h, mode))
        d = IL.ILQR(dyn, obj, T).device_solver(B, max_iter=1, obj_tol=0.0)
        d.init(x1t, Ut); d.iterate(1)
        X, U, J, K, k = d.get(gains=True)
        out[mode] = (K.double().clone(), k.double().clone(), 0)
    lib.check(lib.cdll.od_set_cooperative(dyn._h, 0))
    (K2, k2, r2), (K0, k0, r0) = out[2], out[0]
    keep = [b for b in range(B) if b != 1]
    # (without regularisation beyond the initial 1e-6 no trajectory factorises: the one-pass entry point says so)
    sol = IL.ILQR(dyn, obj, T)
    Xn, An, Bn, _ = sol.linearize(x1t, Ut)
    bst = sol.backward(An, Bn, obj.expansion(Xn, Ut.double(), None, 0.0), 1e-6)[3]
    assert (bst[keep] == 0).all()
    assert torch.isfinite(K0[..., keep]).all() and K0[..., keep].abs().max() > 0
    tol = 1e-9 if dtype == torch.float64 else 1e-5
    assert ((K0 - K2)[..., keep].abs().max() <= tol * K2[..., keep].abs().max()).item() and ((k0 - k2)[..., keep].abs().max() <= tol * k2[..., keep].abs().max().clamp(min=1.0)).item()
    assert (K0[..., 1] == 0).all() and (k0[..., 1] == 0).all() and (K2[..., 1] == 0).all()


def rocket_problem(lib, device, B, T, dtype=torch.float64, seed=0):
    """rocket soft landing with the thrust-cone projection on the path (examples/rocket.jl:15-50 sizes:
    h=0.05, u_max=12.5); quadratic costs"""
    from optimization_dynamics_amd import rocket as rk
    info = rk.RocketInfo(models.rocket, 12.5, 0.05, dtype=dtype, device=device, lib=lib)
    dyn = rk.RocketDynamics(info, project=True)
    goal = np.zeros(12); goal[2] = 2.0
    Q = np.diag([1e-2] * 3 + [1e-1] * 3 + [1e-2] * 3 + [1e-1] * 3)
    R = np.diag([1e-2, 1e-2, 1e-3])
    QT = np.diag([10.0] * 12)
    obj = IL.QuadraticObjective(Q, R, QT, x_ref=goal, device=device)
    rng = np.random.default_rng(seed)
    x1 = np.zeros((12, B)); x1[0] = 1.0; x1[1] = 0.5; x1[2] = 4.0
    x1 += rng.normal(0, 0.05, (12, B)); x1[3:6] *= 0.2
    U0 = np.zeros((3, T, B)); U0[2] = 9.81 + 0.1 * rng.normal(size=(T, B))      # hover thrust (mass 1)
    U0[:2] = 1e-3 * rng.normal(size=(2, T, B))
    return dyn, obj, x1, U0


def check_rollout_against_oracle(oracle, dyn, x1, U0, X, A, Bm, dtype, ntraj=64, h=0.05, u_max=12.5):
    """a projected rocket rollout of the device (X, and its linearisation A, Bm) against the oracle on the first `ntraj` trajectories
    (the oracle's rollouts and solves run batched under OpenMP), in the handle's precision:
      * chained: the oracle's f_rocket_proj from x1 along the same controls, its projection with the rounding-decided places completed
        as exact arithmetic has them (`exact_boundary`, like the device since round 6: parity_checks.check_rocket_sweep) -- all T steps
        of every trajectory at 1e-6 in double (round 5, against the literal projection loop: 1e-3, the kappa_tol level of its
        line-search ties), 5e-6 in single (60 chained steps of float rounding);
      * knot by knot, no path dependence left: f_rocket_proj / fx of the device on its own rollout states (independent knots, the
        solve the linearisation comes from) against the oracle's dynamics step from the same state with the control the DEVICE
        projected to -- at the north_star's bars in both precisions (1e-6 / 1e-4: the single-precision handle finishes its dynamics
        steps in double, od_set_mixed_precision), on every converged knot of those trajectories"""
    T = U0.shape[1]
    nt = min(ntraj, x1.shape[1])
    Uq = U0.astype(np.float32).astype(np.float64) if dtype == torch.float32 else U0
    Xn, An = X.double().cpu().numpy()[:, :, :nt], A.double().cpu().numpy()[:, :, :, :nt]
    oracle.lib().od_oracle_set_exact_boundary(1)
    try:
        Xo, _, so = oracle.rocket_rollout(h, u_max, x1[:, :nt], Uq[:, :, :nt], project=True)
    finally:
        oracle.lib().od_oracle_set_exact_boundary(0)
    okt = (so == 0x11).all(0)                                   # trajectories whose every oracle solve converged
    assert okt.mean() > 0.9
    err = np.abs(Xn - Xo).max(0) / np.maximum(1.0, np.abs(Xo).max(0))          # (T+1, nt)
    chain_tol = 1e-6 if dtype == torch.float64 else 5e-6
    # (a projection that is ill-conditioned in its own rounding -- 0.05 % of apex-heavy controls, check_rocket_sweep -- parts a whole
    # trajectory from there on: at most one trajectory in 50 may, and only at the kappa_tol level)
    bad = okt & (err.max(0) >= chain_tol)
    assert bad.sum() <= nt // 50 and err[:, okt].max() < 1e-3, (str(dtype), int(bad.sum()), float(err[:, okt].max()))
    okt = okt & ~bad
    # independent knots: the device's own states, all T * nt of them in one launch
    Xk = np.ascontiguousarray(Xn[:, :T].reshape(12, T * nt))
    Uk = np.ascontiguousarray(Uq[:, :, :nt].reshape(3, T * nt))
    Y, DX, DU, UP, st = dyn.info.solve(torch.tensor(Xk), torch.tensor(Uk), project=True, grads=True)
    Y, DX, UP, st = Y.double().cpu().numpy(), DX.double().cpu().numpy(), UP.double().cpu().numpy(), st.cpu().numpy()
    Yo, DZo, sto, _ = oracle.rocket_batch(h, Xk, UP, True)
    ok = ((st & 0x33) == 0x33) & (sto == 1)
    assert ok.mean() > 0.99, ok.mean()
    rel = lambda a, b: np.abs(a - b).reshape(-1, T * nt).max(0) / np.maximum(1.0, np.abs(b).reshape(-1, T * nt).max(0))
    es, ex = rel(Y, Yo)[ok], rel(DX, DZo[:, :12])[ok]
    assert es.max() < 1e-6 and ex.max() < 1e-4, (str(dtype), float(es.max()), float(ex.max()))
    # (the rollout kernel and the independent-knot kernel are two compilations of the same solve; round 5 allowed them the kappa_tol
    # level of the projection's line-search ties, 1e-3 -- the ties are gone)
    Xn1 = Xn[:, 1:].reshape(12, T * nt)
    two = rel(Y, Xn1)[ok]
    assert (two > chain_tol).sum() <= max(1, ok.sum() // 2000) and two.max() <= 1e-3, (str(dtype), int((two > chain_tol).sum()), float(two.max()))
    # the linearisation IS that solve
    Ak = An.reshape(12, 12, T * nt)
    assert rel(DX, Ak)[ok].max() <= 1e-6
    return dict(trajectories=int(nt), knots=int(ok.sum()), chained_state_rel_max=float(err[:, okt].max()), knot_state_rel_max=float(es.max()), knot_fx_rel_max=float(ex.max()))


def check_rocket_ilqr(oracle, lib, device, B=4, T=20, dtype=torch.float64):
    dyn, obj, x1, U0 = rocket_problem(lib, device, B, T, dtype=dtype)
    x1t, Ut = torch.tensor(x1, device=device), torch.tensor(U0, device=device)
    # rollout == chained f_rocket_proj of the oracle, in either precision
    X, A, Bm, st, _, _ = dyn.rollout(x1t, Ut)
    check_rollout_against_oracle(oracle, dyn, x1, U0, X, A, Bm, dtype, ntraj=min(B, 64))
    # the time recursion without the projection (no path dependence at all): every step of the device's rollout against the
    # oracle's f_rocket from the device's own previous state -- 1e-6 in both precisions (single: the double-precision residual
    # refinement of the rollout kernels, csrc/od_units.h::rocket_refine64)
    from optimization_dynamics_amd import rocket as rk
    Xn, Un, sn = rk._rocket_rollout(dyn.info, x1t, Ut, False)
    Xn = Xn.double().cpu().numpy()
    Uq = U0.astype(np.float32).astype(np.float64) if dtype == torch.float32 else U0
    for b in range(min(B, 2)):
        for t in range(T):
            ok, y, dz, it = oracle.rocket(0.05, Xn[:, t, b], Uq[:, t, b], False)
            assert np.abs(Xn[:, t + 1, b] - y).max() < 1e-6 * max(1, np.abs(y).max()), (dtype, b, t, np.abs(Xn[:, t + 1, b] - y).max())
    solver = IL.ILQR(dyn, obj, T)
    J0 = obj.value(X, Ut.double())
    Xs, Us, J, hist = solver.solve(x1t, Ut, max_iter=10)
    Jf = obj.value(Xs, Us)
    assert torch.isfinite(Jf).all() and (Jf <= J0 + 1e-6).all() and (Jf < 0.8 * J0).float().mean().item() > 0.7
    return J0, Jf


def config5_problem(lib, device, B, dtype=torch.float64, T=61, seed=1):
    """BASELINE config 5 with the inputs of examples/rocket.jl, `:projection` mode: h = 0.05, T = 61 (60 steps), u_max = 12.5
    (:16-20), x1 = [2.5, 2.5, 10, MRP(RotZ(pi/4) RotY(-pi/2)), 0, 0, -1, 0, 0, 0] (:44-50), xT = [0, 0, 1, MRP(RotZ(pi/4)), 0...]
    (:52-55), the quadratic objective of :58-73 (stage weights h [0.1 x3, 1e-5 x3, 0.1 x3, 1000 x3] and h [1000, 1000, 100], terminal
    h 1000 I), initial controls 1e-3 randn (:116-117; trajectory b draws with seed + b, so trajectory 0 is the example's).
    The example's stage / terminal constraints are IterativeLQR's business and not part of the path."""
    import math
    from optimization_dynamics_amd import rocket as rk
    sys_path_examples()
    import rocket as ex
    h, u_max = 0.05, 12.5
    info = rk.RocketInfo(models.rocket, u_max, h, dtype=dtype, device=device, lib=lib)
    dyn = rk.RocketDynamics(info, project=True)
    x1 = np.zeros(12); x1[:3] = [2.5, 2.5, 10.0]
    x1[3:6] = ex.mrp_of(ex.rot_z(0.25 * math.pi) @ ex.rot_y(-0.5 * math.pi)); x1[8] = -1.0
    xT = np.zeros(12); xT[2] = 1.0
    xT[3:6] = ex.mrp_of(ex.rot_z(0.25 * math.pi) @ ex.rot_y(0.0))
    Q = np.diag(h * np.r_[1.0e-1 * np.ones(3), 1.0e-5 * np.ones(3), 1.0e-1 * np.ones(3), 1000.0 * np.ones(3)])
    R = np.diag(h * np.array([1000.0, 1000.0, 100.0]))
    QT = h * 1000.0 * np.eye(12)
    obj = IL.QuadraticObjective(Q, R, QT, x_ref=xT, device=device)
    U0 = np.stack([1.0e-3 * np.random.default_rng(seed + b).normal(size=(3, T - 1)) for b in range(B)], axis=-1)
    return dyn, obj, np.repeat(x1[:, None], B, axis=1), U0


def sys_path_examples():
    import os, sys
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples")
    if d not in sys.path:
        sys.path.insert(0, d)


def check_config5(oracle, lib, device, B=1024, iters=12):
    """BASELINE config 5 as stated -- rocket thrust-cone SOCP step inside the full iLQR loop with implicit gradients, T = 61 --
    in double and in single precision, against the oracle:
      * the nominal rollout and its linearisation against the oracle's f_rocket_proj / fx (check_rollout_against_oracle);
      * K, k, dV of od_ilqr_backward at (n, m) = (12, 3) on that linearisation against the numpy Riccati recursion;
      * the iteration on the device (od_ilqr_solve) against the loop composed on the host, cost by cost;
      * the single-precision cost history against the double-precision one."""
    from oracle import ilqr_np
    hist = {}
    T = 60
    for dtype in (torch.float64, torch.float32):
        dyn, obj, x1, U0 = config5_problem(lib, device, B, dtype=dtype)
        x1t, Ut = torch.tensor(x1, device=device), torch.tensor(U0, device=device)
        X, A, Bm, st, _, _ = dyn.rollout(x1t, Ut)
        need = 0x33
        assert ((st & 1) == 1).double().mean().item() > 0.999
        stats = check_rollout_against_oracle(oracle, dyn, x1, U0, X, A, Bm, dtype, ntraj=64)
        assert stats["trajectories"] >= min(B, 64)
        sol = IL.ILQR(dyn, obj, T)
        quad = obj.expansion(X, Ut, None, 0.0)
        K, k, dV, bst = sol.backward(A, Bm, quad, 1e-6)
        assert (bst == 1).all()
        lxx, luu, lux, lx, lu, VxxT, VxT = [q.cpu().numpy() for q in quad]
        An, Bn, Kn, kn, dVn = A.cpu().numpy(), Bm.cpu().numpy(), K.cpu().numpy(), k.cpu().numpy(), dV.cpu().numpy()
        n, m = 12, 3
        # ALL B trajectories against the vectorised numpy recursion (oracle/ilqr_np.py::backward_batch)
        col = lambda M, r, c: np.moveaxis(M.reshape(r, c, T, B, order="F"), (2, 3), (1, 0))          # (r*c, T, B) col-major -> (B, T, r, c)
        Kr, kr, dVr = ilqr_np.backward_batch(
            np.moveaxis(An, (2, 3), (1, 0)), np.moveaxis(Bn, (2, 3), (1, 0)), col(lxx, n, n), col(luu, m, m), col(lux, m, n),
            np.moveaxis(lx, (1, 2), (1, 0)), np.moveaxis(lu, (1, 2), (1, 0)), np.moveaxis(VxxT.reshape(n, n, B, order="F"), 2, 0), VxT.T, 1e-6)
        Kd = col(Kn, m, n)
        relb = lambda a, b: np.abs(a - b).reshape(B, -1).max(1) / np.maximum(1.0, np.abs(b).reshape(B, -1).max(1))
        eK, ek, eV = relb(Kd, Kr), relb(np.moveaxis(kn, (1, 2), (1, 0)), kr), relb(dVn.T, dVr)
        assert eK.max() < 1e-8 and ek.max() < 1e-8 and eV.max() < 1e-8, (str(dtype), float(eK.max()), float(ek.max()), float(eV.max()))
        kw = dict(max_iter=iters, max_al_iter=1, obj_tol=0.0)
        got = sol.solve(x1t, Ut, **kw)
        ref = IL.ILQR(dyn, obj, T).solve_stepwise(x1t, Ut, **kw)
        assert len(got[3]) == len(ref[3]) == iters
        for i, (ja, jb) in enumerate(zip(got[3], ref[3])):
            assert ((ja - jb).abs() <= 1e-9 * jb.abs().clamp(min=1.0)).all(), (dtype, i, (ja - jb).abs().max().item())
        J0 = obj.value(X.double(), Ut)
        assert torch.isfinite(got[2]).all() and (got[2] <= J0 + 1e-6).all() and (got[2] < 0.9 * J0).double().mean().item() > 0.9
        # (knots whose projection stalled on the boundary of the cone -- status bits 16 / 32 clear, ~0.2 % of the controls this close to
        # the apex of the cone, in the oracle alike, DESIGN.md section 7 -- or whose dynamics solve did not converge)
        nbad = sol._dev.info().bad_linearisations
        assert nbad <= max(1, T * B // 200), nbad
        hist[dtype] = torch.stack(got[3]).cpu().numpy()
    # single against double precision: the same optimisation (accept decisions of single trajectories may differ, the batch does not)
    h64, h32 = hist[torch.float64], hist[torch.float32]
    rel = np.abs(h32.mean(1) - h64.mean(1)) / np.abs(h64.mean(1))
    assert rel.max() < 2e-2, rel
    assert np.median(np.abs(h32[0] - h64[0]) / np.abs(h64[0])) < 1e-3
    return hist


def constrained_problem(lib, device, problem, B, T, dtype=torch.float64, seed=1):
    """the problems above with affine constraints of the kinds the reference's examples pose (examples/planar_push.jl:105-106: control
    bounds as stage inequalities; examples/rocket.jl:82-110: a state bound at every stage, a terminal box and terminal equalities)"""
    if problem == "cartpole":
        im, obj, x1, U0 = cartpole_problem(lib, device, B, T, seed=seed)
        obj = IL.QuadraticObjective(obj.Q.cpu().numpy(), obj.R.cpu().numpy(), obj.QT.cpu().numpy(), obj.x_ref.cpu().numpy(), device=device)
        n, m = 4, 1
        # -0.45 <= u <= 0.45 (two inequality rows: active -- the unconstrained optimum pushes with 0.49), cart position x2[0] <= 0.35 (one row); terminal: |pole angle| <= 0.05 (two inequality
        # rows on q2), cart at 0.3 (equalities on q1[0], q2[0])
        Cs = np.zeros((3, n)); Ds = np.array([[1.0], [-1.0], [0.0]]); Cs[2, 2] = 1.0
        ds = np.array([0.45, 0.45, 0.35])
        Ct = np.zeros((4, n)); Ct[0, 3] = 1.0; Ct[1, 3] = -1.0; Ct[2, 0] = 1.0; Ct[3, 2] = 1.0
        dt = np.array([0.05, 0.05, 0.3, 0.3])
        obj.set_constraints(stage=(Cs, Ds, ds, 3), terminal=(Ct, dt, 2))
        return im, obj, x1, U0
    dyn, obj, x1, U0 = rocket_problem(lib, device, B, T, dtype=dtype, seed=seed)
    n, m = 12, 3
    Cs = np.zeros((1, n)); Cs[0, 2] = -1.0                     # length - x[3] <= 0  (examples/rocket.jl:84)
    Ds = np.zeros((1, m)); ds = np.array([-1.0])
    Ct = np.zeros((6, n)); dt = np.zeros(6)
    Ct[0, 0] = -1.0; dt[0] = 0.5; Ct[1, 0] = 1.0; dt[1] = 0.5          # -0.5 <= x <= 0.5, -0.75 <= y <= 0.75 (:104-107)
    Ct[2, 1] = -1.0; dt[2] = 0.75; Ct[3, 1] = 1.0; dt[3] = 0.75
    Ct[4, 2] = 1.0; dt[4] = 2.0; Ct[5, 8] = 1.0; dt[5] = 0.0            # z = 2, vertical speed 0 (equalities)
    obj.set_constraints(stage=(Cs, Ds, ds, 1), terminal=(Ct, dt, 4))
    return dyn, obj, x1, U0


def check_device_iteration_constrained(lib, device, problem="cartpole", B=6, T=25, dtype=torch.float64, max_iter=10, max_al_iter=4, tol=1e-9, expect_feasible=True):
    """od_ilqr_solve with stage / terminal affine constraints (inequalities by active set) against the host-composed loop: iteration
    counts, costs iteration by iteration, trajectories, final violation"""
    im, obj, x1, U0 = constrained_problem(lib, device, problem, B, T, dtype=dtype)
    x1t, Ut = torch.tensor(x1, device=device), torch.tensor(U0, device=device)
    kw = dict(max_iter=max_iter, max_al_iter=max_al_iter, obj_tol=1e-7, con_tol=1e-4)
    ref = IL.ILQR(im, obj, T).solve_stepwise(x1t, Ut, **kw)
    sol = IL.ILQR(im, obj, T)
    got = sol.solve(x1t, Ut, **kw)
    info = sol._dev.info()
    assert len(got[3]) == len(ref[3]) == info.iterations, (len(got[3]), len(ref[3]), info.iterations)
    for i, (ja, jb) in enumerate(zip(got[3], ref[3])):
        assert ((ja - jb).abs() <= tol * jb.abs().clamp(min=1.0)).all(), (problem, i, (ja - jb).abs().max().item())
    sc = lambda t: t.abs().max().clamp(min=1.0).item()
    assert (got[0] - ref[0]).abs().max().item() < 1e3 * tol * sc(ref[0]) and (got[1] - ref[1]).abs().max().item() < 1e3 * tol * sc(ref[1])
    v_dev, v_ref = obj.violation(got[0], got[1]).max().item(), obj.violation(ref[0], ref[1]).max().item()
    assert abs(v_dev - v_ref) <= 1e-6 * max(1.0, v_ref)
    assert abs(info.max_violation - v_dev) <= 1e-12 * max(1.0, v_dev) or info.al_iterations == max_al_iter - 1      # (measured after the last round's update)
    if expect_feasible:
        # the multiplier rounds did their work: constraints met to con_tol where the unconstrained optimum violates them
        assert info.al_done == 1 and v_dev < 1e-4, (info.al_done, v_dev)
        free = IL.QuadraticObjective(obj.Q.cpu().numpy(), obj.R.cpu().numpy(), obj.QT.cpu().numpy(), obj.x_ref.cpu().numpy(), device=device)
        Xf, Uf, _, _ = IL.ILQR(im, free, T).solve(x1t, Ut, max_iter=30, obj_tol=1e-7)
        assert obj.violation(Xf, Uf).max().item() > 20.0 * v_dev
    return got, ref, v_dev


def rocket_example_problem(lib, device, B, dtype=torch.float64):
    """examples/rocket.jl, MODE = :projection, in full: the inputs of config5_problem plus the example's constraints -- at every stage
    length - x[3] <= 0 (:82-88), at the horizon -0.5 <= x[1] <= 0.5, -0.75 <= x[2] <= 0.75 and x[3:12] = xT[3:12] (:102-110) -- by
    augmented Lagrangian; solver options of :123-134 (alpha_min 1e-5, obj_tol 1e-3, max_iter 100, max_al_iter 15, con_tol 0.005,
    rho 1 x 10)"""
    dyn, obj, x1, U0 = config5_problem(lib, device, B, dtype=dtype)
    xT = obj.x_ref.cpu().numpy()
    n, m = 12, 3
    Cs = np.zeros((1, n)); Cs[0, 2] = -1.0
    Ds = np.zeros((1, m)); ds = np.array([-1.0])
    Ct = np.zeros((14, n)); dt = np.zeros(14)
    Ct[0, 0] = -1.0; dt[0] = 0.5; Ct[1, 0] = 1.0; dt[1] = 0.5
    Ct[2, 1] = -1.0; dt[2] = 0.75; Ct[3, 1] = 1.0; dt[3] = 0.75
    for k in range(10):
        Ct[4 + k, 2 + k] = 1.0; dt[4 + k] = xT[2 + k]
    obj.set_constraints(stage=(Cs, Ds, ds, 1), terminal=(Ct, dt, 4))
    opts = dict(max_iter=100, max_al_iter=15, con_tol=0.005, obj_tol=1.0e-3, rho_init=1.0, rho_scale=10.0)
    alphas = tuple(2.0 ** -i for i in range(17))          # down to alpha_min = 1e-5
    return dyn, obj, x1, U0, xT, opts, alphas


def check_rocket_example(lib, device, B=1, dtype=torch.float64):
    """the reference's rocket landing example, `:projection` mode with its constraints, solved by od_ilqr_solve: every problem reaches
    con_tol = 0.005 (the frozen host-side ilqr_al.py needs 484 iterations / 21.8 s for one problem, examples/README.md), the thrust
    cone holds for the applied controls (examples/rocket.jl:151), the rocket never goes below its own length"""
    dyn, obj, x1, U0, xT, opts, alphas = rocket_example_problem(lib, device, B, dtype=dtype)
    sol = IL.ILQR(dyn, obj, 60, alphas=alphas)
    X, U, J, hist = sol.solve(torch.tensor(x1, device=device), torch.tensor(U0, device=device), **opts)
    info = sol._dev.info()
    fl, viol, rho = sol._dev.status()
    assert info.al_done == 1 and (fl == 3).all(), (info.al_done, fl.cpu().numpy()[:8])
    assert viol.max().item() < 0.005 and abs(obj.violation(X, U).max().item() - viol.max().item()) < 1e-12
    assert (X[2] >= 1.0 - 0.005).all()                                          # stage constraint
    assert (X[0, -1].abs() <= 0.5 + 0.005).all() and (X[1, -1].abs() <= 0.75 + 0.005).all()
    assert (X[2:, -1] - torch.tensor(xT[2:], device=device)[:, None]).abs().max().item() < 0.005
    UP = dyn.info.project(U.reshape(3, -1), grads=False)[0].double()
    assert (torch.hypot(UP[0], UP[1]) <= UP[2] + 1e-2).all() and (UP[2] <= 12.5 + 1e-3).all()
    print("rocket example with its constraints, %d problem(s), %s: %d iterations, %d multiplier updates, violation %.2e, objective %.1f .. %.1f"
          % (B, str(dtype).split(".")[-1], info.iterations, info.al_iterations, viol.max().item(), obj.value(X, U).min().item(), obj.value(X, U).max().item()))
    return info


def rocket_example_nominal_problem(lib, device, B, dtype=torch.float64):
    """examples/rocket.jl with `MODE = :nominal` -- what the file runs as shipped (:11-12, the second assignment wins): f_rocket without
    the projection (:30-35), the thrust limits as stage constraints -1 <= u[1:2] <= 1, 0 <= u[3] <= u_max together with
    length - x[3] <= 0 (:89-99, seven inequalities), the terminal constraints and solver options of the `:projection` mode"""
    from optimization_dynamics_amd import rocket as rk
    dyn, obj, x1, U0 = config5_problem(lib, device, B, dtype=dtype)
    dyn = rk.RocketDynamics(dyn.info, project=False)
    xT = obj.x_ref.cpu().numpy()
    n, m = 12, 3
    Cs = np.zeros((7, n)); Ds = np.zeros((7, m)); ds = np.zeros(7)
    for j in range(2):
        Ds[2 * j, j], ds[2 * j] = -1.0, 1.0
        Ds[2 * j + 1, j], ds[2 * j + 1] = 1.0, 1.0
    Ds[4, 2], ds[4] = -1.0, 0.0
    Ds[5, 2], ds[5] = 1.0, 12.5
    Cs[6, 2], ds[6] = -1.0, -1.0
    Ct = np.zeros((14, n)); dt = np.zeros(14)
    Ct[0, 0] = -1.0; dt[0] = 0.5; Ct[1, 0] = 1.0; dt[1] = 0.5
    Ct[2, 1] = -1.0; dt[2] = 0.75; Ct[3, 1] = 1.0; dt[3] = 0.75
    for k in range(10):
        Ct[4 + k, 2 + k] = 1.0; dt[4 + k] = xT[2 + k]
    obj.set_constraints(stage=(Cs, Ds, ds, 7), terminal=(Ct, dt, 4))
    opts = dict(max_iter=100, max_al_iter=15, con_tol=0.005, obj_tol=1.0e-3, rho_init=1.0, rho_scale=10.0)
    alphas = tuple(2.0 ** -i for i in range(17))
    return dyn, obj, x1, U0, xT, opts, alphas


def check_rocket_example_nominal(lib, device, B=4, dtype=torch.float64, need=0.5):
    """the rocket example as shipped (`:nominal`) through od_ilqr_solve.  The landing has two basins from initial controls of 1e-3 randn
    (examples/rocket.jl:116-117; Julia's stream of seed 1 cannot be drawn here): most starts end at the landing with every constraint
    at con_tol, the others flip the rocket over in the FIRST multiplier round (rho = 1) and stay infeasible -- a property of the
    problem's landscape, each problem being an independent solve.  At least `need` of the starts must land, and for those: thrust
    limits, floor, terminal box and terminal state to con_tol, flags == 3"""
    dyn, obj, x1, U0, xT, opts, alphas = rocket_example_nominal_problem(lib, device, B, dtype=dtype)
    sol = IL.ILQR(dyn, obj, 60, alphas=alphas)
    X, U, J, hist = sol.solve(torch.tensor(x1, device=device), torch.tensor(U0, device=device), **opts)
    info = sol._dev.info()
    fl, viol, rho = sol._dev.status()
    ok = (fl == 3)
    assert ok.double().mean().item() >= need, fl.cpu().numpy()
    assert torch.equal(ok, viol < opts["con_tol"])
    assert (obj.violation(X, U) - viol).abs().max().item() < 1e-9 * max(1.0, viol.max().item())
    Xo, Uo = X[:, :, ok], U[:, :, ok]
    tol = opts["con_tol"]
    assert (Uo[:2].abs() <= 1.0 + tol).all() and (Uo[2] >= -tol).all() and (Uo[2] <= 12.5 + tol).all()
    assert (Xo[2, :-1] >= 1.0 - tol).all()
    assert (Xo[0, -1].abs() <= 0.5 + tol).all() and (Xo[1, -1].abs() <= 0.75 + tol).all()
    assert (Xo[2:, -1] - torch.tensor(xT[2:], device=device)[:, None]).abs().max().item() < tol
    cone = (torch.hypot(Uo[0], Uo[1]) <= Uo[2] + tol).double().mean().item()        # (:151: not a constraint of this mode)
    print("rocket example as shipped (:nominal), %d problem(s), %s: %d of them land, %d iterations, %d multiplier updates, objective %.1f .. %.1f, "
          "thrust inside the cone at %.0f %% of the knots" % (B, str(dtype).split(".")[-1], int(ok.sum()), info.iterations, info.al_iterations,
                                                           obj.value(Xo, Uo).min().item(), obj.value(Xo, Uo).max().item(), 100 * cone))
    return dict(problems=B, landed=int(ok.sum()), iterations=int(info.iterations), rounds=int(info.al_iterations),
                objective=[obj.value(Xo, Uo).min().item(), obj.value(Xo, Uo).max().item()], cone_fraction=cone)


def planar_push_example(lib, device, mode, B, seed=1):
    """examples/planar_push.jl (`:rotate` / `:translate`): h = 0.1, T = 26, kappa_eval 1e-4, kappa_grad 1e-2 (:18-22); objective
    1/2 v1'W v1 + 1/2 (x - xT)'Wx (x - xT) + 1/2 ru u'u per stage, without the control term at the horizon (:57-83) -- a quadratic
    form in x: 1/2 (x - x*)'Q(x - x*) + const with Q = Qv + Wx, x* = Q^-1 Wx xT; control limits -5 <= u <= 5 as stage inequalities
    (:90-99), goal on rows 1:3, 6:8 as terminal equalities (:101-104); initial controls of :111; options of :117-128.  Problem b > 0
    perturbs the initial controls by 1e-2 randn."""
    import math
    h, T, r_dim = 0.1, 25, 0.1
    im = P.make_im("planar_push", lib, device)
    if mode == "translate":
        q0 = [0.0, 0.0, 0.0, -r_dim - 1.0e-8, 0.0]; goal = (1.0, 0.0, 0.0); ru = 1.0e-1
        U0 = np.zeros((2, T)); U0[0, :4] = 1.0
    else:
        q0 = [0.0, 0.0, 0.0, -r_dim - 1.0e-8, -0.01]; goal = (0.5, 0.5, 0.5 * math.pi); ru = 1.0e-2
        U0 = np.zeros((2, T)); U0[0, :4] = 1.0; U0[0, 4:9] = 0.5
    qT = [goal[0], goal[1], goal[2], goal[0] - r_dim, goal[1] - r_dim]
    xT = np.array(qT + qT)
    W = np.diag([1.0, 1.0, 1.0, 0.1, 0.1]); Wx = np.diag([1.0, 1.0, 1.0, 0.1, 0.1] * 2)
    Qv = np.block([[W, -W], [-W, W]]) / h ** 2
    Q = Qv + Wx
    xs = np.linalg.solve(Q, Wx @ xT)
    obj = IL.QuadraticObjective(Q, ru * np.eye(2), Q, x_ref=xs, goal_idx=[0, 1, 2, 5, 6, 7], goal=xT[[0, 1, 2, 5, 6, 7]], device=device)
    Cs = np.zeros((4, 10)); Ds = np.vstack([-np.eye(2), np.eye(2)]); ds = np.array([5.0, 5.0, 5.0, 5.0])
    obj.set_constraints(stage=(Cs, Ds, ds, 4))
    rng = np.random.default_rng(seed)
    U = np.repeat(U0[:, :, None], B, axis=2)
    U[:, :, 1:] += 1e-2 * rng.normal(size=(2, T, B - 1))
    x1 = np.repeat(np.array(q0 + q0)[:, None], B, axis=1)
    opts = dict(max_iter=10, max_al_iter=10, con_tol=0.005, obj_tol=1.0e-3)
    return im, obj, x1, U, xT, T, opts


def cartpole_example(lib, device, mode, B, seed=1):
    """examples/cartpole.jl: swing-up, h = 0.05, T = 51, objective u'u per stage and (x - xT)'(x - xT) at the horizon (:50-60), terminal
    constraint x = xT = [0, pi, 0, pi] (:66-70), initial controls -1.5 at the first knot (:77), options of :83-94; `:frictionless` (the
    file's default: kappa 1.0) or `:friction` (joint friction 0.35, kappa_eval 1e-4 / kappa_grad 1e-3)"""
    import math
    T = 50
    name = "cartpole_friction" if mode == "friction" else "cartpole_frictionless"
    im = P.make_im(name, lib, device)
    if mode != "friction":
        im.set_options(kappa_eval_tol=1.0, kappa_grad_tol=1.0)
    xT = np.array([0.0, math.pi, 0.0, math.pi])
    obj = IL.QuadraticObjective(np.zeros((4, 4)), 2.0 * np.eye(1), 2.0 * np.eye(4), x_ref=xT, goal_idx=[0, 1, 2, 3], goal=xT, device=device)
    U0 = np.zeros((1, T, B)); U0[0, 0] = -1.5
    U0[:, :, 1:] += 1e-2 * np.random.default_rng(seed).normal(size=(1, T, B - 1))
    opts = dict(max_iter=100, max_al_iter=20, con_tol=0.005, obj_tol=1.0e-5)
    return im, obj, np.zeros((4, B)), U0, xT, T, opts


def check_reference_example(lib, device, which, B=1, need=1.0):
    """one of the reference's examples through od_ilqr_solve (the whole solve on the device): constraints to the example's con_tol
    on at least `need` of the problems, controls inside their limits, the returned trajectory consistent with its controls"""
    if which.startswith("planar_push"):
        im, obj, x1, U0, xT, T, opts = planar_push_example(lib, device, which.split(":")[1], B)
    else:
        im, obj, x1, U0, xT, T, opts = cartpole_example(lib, device, which.split(":")[1], B)
    sol = IL.ILQR(im, obj, T, alphas=tuple(2.0 ** -i for i in range(17)))
    x1t, Ut = torch.tensor(x1, device=device), torch.tensor(U0, device=device)
    X, U, J, hist = sol.solve(x1t, Ut, **opts)
    info = sol._dev.info()
    fl, viol, rho = sol._dev.status()
    okc = (viol < opts["con_tol"])
    assert okc.double().mean().item() >= need, (which, okc.double().mean().item(), viol.max().item())
    assert ((fl & 2) != 0).eq(okc).all()
    if obj.stage is not None:
        assert (U.abs() <= 5.0 + opts["con_tol"]).all()
    Xr = im.rollout(x1t, U, grads=False)[0]
    assert (Xr - X).abs().max().item() < 1e-9
    print("%s, %d problem(s): %d lockstep iterations, %d multiplier rounds, %.0f %% at con_tol, max violation %.2e, goal error %.2e"
          % (which, B, info.iterations, info.al_iterations, 100 * okc.double().mean().item(), viol.max().item(),
             (X[:, -1] - torch.tensor(xT, device=device)[:, None])[obj.goal_idx.cpu()].abs().max().item()))
    return info, viol


def hopper_example(lib, device, B, seed=1):
    """examples/hopper.jl, GAIT 1, with the initial configurations held at the example's standing pose: T = 21, h = 0.05, kappa_eval 1e-4,
    kappa_grad 1e-3 (:12-13, 42); objective of the stages t >= 2 and of the horizon (:205-216: 1/2 (x - x_ref)' 0.1 diag(1, 10, ...) (x - x_ref)
    + 1/2 0.1 u'u; 1/2 |x - x_ref|^2 at T) around q_ref = [0.5, 0.75 + r_foot, 0, 0.25] (:181); control limits -10 <= u <= 10 at every stage
    (:224-225, 241-246); the terminal constraint (:248-257) with theta = x1: travel x[1], x[5] >= 0.5 + theta (two inequalities), the other six
    configuration entries back at theta (equalities) -- one hop forward that ends in the pose it started from; standing controls (:270) as the
    initial guess; solver options of :277-287.  (The example also optimises theta, through a first stage of its own dimensions 8 -> 16 with
    nonlinear foot-position constraints (:227-239): that part is beyond the device solver's uniform stages, examples/hopper_gait.py does it
    with the host loop.)  Problem b > 0 perturbs the initial controls by 1e-2 randn."""
    from optimization_dynamics_amd.codegen.models import HOPPER_PARAMS as HP
    h, T = 0.05, 20
    im = P.make_im("hopper", lib, device)
    r = HP["foot_radius"]
    q1 = np.array([0.0, 0.5 + r, 0.0, 0.5]); q_ref = np.array([0.5, 0.75 + r, 0.0, 0.25])
    x1v, x_ref = np.concatenate([q1, q1]), np.concatenate([q_ref, q_ref])
    w = np.array([1.0, 10.0, 1.0, 10.0] * 2)
    obj = IL.QuadraticObjective(0.1 * np.diag(w), 0.1 * np.eye(2), np.eye(8), x_ref=x_ref, device=device)
    Cs = np.zeros((4, 8)); Ds = np.vstack([-np.eye(2), np.eye(2)]); ds = np.full(4, 10.0)
    Ct = np.zeros((8, 8)); dt = np.zeros(8)
    Ct[0, 0] = -1.0; dt[0] = -(0.5 + x1v[0])                    # x_travel - (x[1] - theta[1]) <= 0
    Ct[1, 4] = -1.0; dt[1] = -(0.5 + x1v[4])
    for k, i in enumerate([1, 2, 3, 5, 6, 7]):
        Ct[2 + k, i] = 1.0; dt[2 + k] = x1v[i]
    obj.set_constraints(stage=(Cs, Ds, ds, 4), terminal=(Ct, dt, 2))
    U0 = np.zeros((2, T, B)); U0[1] = HP["gravity"] * HP["mass_body"] * 0.5 * h
    U0[:, :, 1:] += 1e-2 * np.random.default_rng(seed).normal(size=(2, T, B - 1))
    x1 = np.repeat(x1v[:, None], B, axis=1)
    opts = dict(max_iter=10, max_al_iter=15, con_tol=1.0e-3, obj_tol=1.0e-3, rho_init=1.0, rho_scale=10.0)
    return im, obj, x1, U0, x1v, T, opts


def check_hopper_example(lib, device, B=1, need=1.0):
    """the hopper's gait problem (hopper_example) through od_ilqr_solve -- the headline's model inside the device-resident solver, its
    Riccati pass the 8 / 2 instantiation of the matrix-core kernel: constraints to the example's con_tol, controls inside their limits, the
    hopper one half metre further in the pose it started from, the returned trajectory consistent with its controls"""
    im, obj, x1, U0, x1v, T, opts = hopper_example(lib, device, B)
    sol = IL.ILQR(im, obj, T, alphas=tuple(2.0 ** -i for i in range(17)))
    x1t, Ut = torch.tensor(x1, device=device), torch.tensor(U0, device=device)
    X, U, J, hist = sol.solve(x1t, Ut, **opts)
    info = sol._dev.info()
    fl, viol, rho = sol._dev.status()
    okc = viol < opts["con_tol"]
    assert okc.double().mean().item() >= need, (okc.double().mean().item(), viol.max().item())
    assert ((fl & 2) != 0).eq(okc).all()
    assert (U.abs() <= 10.0 + opts["con_tol"]).all()
    good = okc.nonzero().reshape(-1)
    XT = X[:, -1, good]
    assert (XT[0] >= 0.5 - 1e-3).all() and (XT[4] >= 0.5 - 1e-3).all()
    keep = [1, 2, 3, 5, 6, 7]
    assert (XT[keep] - torch.tensor(x1v[keep], device=device)[:, None]).abs().max().item() < 1e-3
    Xr = im.rollout(x1t, U, grads=False)[0]
    assert (Xr - X).abs().max().item() < 1e-9
    assert (X[1] > 0.0).all() and (X[5] > 0.0).all()                      # the body stays above the ground
    print("hopper gait, %d problem(s): %d lockstep iterations, %d multiplier rounds, %.0f %% at con_tol, max violation %.2e, travel %.3f m, objective %.2f .. %.2f"
          % (B, info.iterations, info.al_iterations, 100 * okc.double().mean().item(), viol.max().item(), XT[4].min().item(), obj.value(X, U).min().item(), obj.value(X, U).max().item()))
    return info, viol


def check_batch_independence(lib, device, problem="cartpole", B=6, T=25, dtype=torch.float64, max_iter=10, max_al_iter=4, pick=(0, 3)):
    """the B problems of a solver are independent solves that share their launches: every trajectory has its own regularisation
    schedule, penalty and flags (csrc/od_ilqr_solver.inc::IlTraj), so what a problem converges to -- trajectory, controls, cost,
    flags, violation -- does not depend on what else is in the batch: solved alone it gives the same numbers, bit for bit (the
    kernels of a batch of one and of a batch of B are the same arithmetic; the iteration counts differ, the batch runs as long as
    its slowest member)"""
    im, obj, x1, U0 = constrained_problem(lib, device, problem, B, T, dtype=dtype)
    x1t, Ut = torch.tensor(x1, device=device), torch.tensor(U0, device=device)
    kw = dict(max_iter=max_iter, max_al_iter=max_al_iter, obj_tol=1e-7, con_tol=1e-4)
    sol = IL.ILQR(im, obj, T)
    X, U, J, hist = sol.solve(x1t, Ut, **kw)
    fl, viol, rho = sol._dev.status()
    for b in pick:
        one = IL.ILQR(im, obj, T)
        X1, U1, J1, h1 = one.solve(x1t[:, b:b + 1].contiguous(), Ut[:, :, b:b + 1].contiguous(), **kw)
        f1, v1, r1 = one._dev.status()
        assert torch.equal(X1[:, :, 0], X[:, :, b]) and torch.equal(U1[:, :, 0], U[:, :, b]) and torch.equal(J1[0], J[b]), (problem, b)
        assert f1[0] == fl[b] and torch.equal(v1[0], viol[b]) and torch.equal(r1[0], rho[b]), (problem, b)
    return fl, viol


def check_device_iteration(lib, device, problem="cartpole", B=6, T=15, dtype=torch.float64, max_iter=8, max_al_iter=2, seed=1, tol=1e-9):
    """od_ilqr_* (the whole iteration on the device, decisions included) against the same loop composed from the separate entry
    points with the decisions on the host (ILQR.solve_stepwise): the same costs iteration by iteration, the same number of
    iterations, the same trajectories.  The two differ in the summation order of the cost expansion only (a loop in k_il_expand, a
    GEMM in torch), so costs agree to rounding, not bit for bit; single precision: the candidates' states are single-precision
    numbers in both, the comparison is as tight."""
    if problem == "cartpole":
        im, obj, x1, U0 = cartpole_problem(lib, device, B, T, seed=seed)
    else:
        im, obj, x1, U0 = rocket_problem(lib, device, B, T, dtype=dtype, seed=seed)
    x1t, Ut = torch.tensor(x1, device=device), torch.tensor(U0, device=device)
    kw = dict(max_iter=max_iter, max_al_iter=max_al_iter, obj_tol=1e-7, con_tol=1e-4)
    ref = IL.ILQR(im, obj, T).solve_stepwise(x1t, Ut, **kw)
    sol = IL.ILQR(im, obj, T)
    got = sol.solve(x1t, Ut, **kw)
    info = sol._dev.info()
    assert len(got[3]) == len(ref[3]) == info.iterations, (len(got[3]), len(ref[3]), info.iterations)
    for i, (ja, jb) in enumerate(zip(got[3], ref[3])):
        assert ((ja - jb).abs() <= tol * jb.abs().clamp(min=1.0)).all(), (i, (ja - jb).abs().max().item())
    sc = lambda t: t.abs().max().clamp(min=1.0).item()
    assert (got[0] - ref[0]).abs().max().item() < 1e3 * tol * sc(ref[0]) and (got[1] - ref[1]).abs().max().item() < 1e3 * tol * sc(ref[1])
    assert ((got[2] - ref[2]).abs() <= tol * ref[2].abs().clamp(min=1.0)).all()
    # the pieces separately: init + n x iterate(1) reproduces solve (no augmented-Lagrangian round)
    d = sol.device_solver(B, max_iter=max_iter, max_al_iter=1, obj_tol=1e-7, con_tol=1e-4)
    d.init(x1t, Ut)
    d.iterate(3)
    d.iterate(2)
    h5 = d.history()
    assert h5.shape[0] == min(5, d.info().iterations)
    for i in range(h5.shape[0]):
        assert torch.equal(h5[i], got[3][i]), i
    return got, ref


def check_forward_pass_early_exit(lib, device, B=4, max_iter=6, scale=300.0):
    """the solver's forward pass ends a candidate at its first knot whose solve fails (PolicyArgs::stop_failed) -- such a rollout can
    never be accepted.  The cartpole with joint friction (examples/cartpole.jl `:friction`) started from wild controls (|u| ~ 300): the
    first policies throw the large step sizes into states whose solves do not converge.  od_ilqr_solve must reproduce, cost by cost,
    the loop composed on the host from od_rollout_policy, which rolls every candidate out to its horizon; and the first policy really
    has such candidates (counted with the public entry point).  -> their number"""
    im, obj, x1, U0, xT, T, opts = cartpole_example(lib, device, "friction", B)
    U0 = scale * np.random.default_rng(5).normal(size=U0.shape)
    kw = dict(opts, max_iter=max_iter, max_al_iter=1)
    x1t, Ut = torch.tensor(x1, device=device), torch.tensor(U0, device=device)
    alphas = tuple(2.0 ** -i for i in range(11))
    ref = IL.ILQR(im, obj, T, alphas=alphas).solve_stepwise(x1t, Ut, **kw)
    sol = IL.ILQR(im, obj, T, alphas=alphas)
    got = sol.solve(x1t, Ut, **kw)
    assert len(got[3]) == len(ref[3]) == sol._dev.info().iterations
    for i, (ja, jb) in enumerate(zip(got[3], ref[3])):
        assert ((ja - jb).abs() <= 1e-9 * jb.abs().clamp(min=1.0)).all(), (i, (ja - jb).abs().max().item())
    sc = max(1.0, ref[0].abs().max().item())
    assert (got[0] - ref[0]).abs().max().item() < 1e-6 * sc and (got[1] - ref[1]).abs().max().item() < 1e-6 * scale
    # the first policy of the solve, rolled out in full through the public entry point: how many candidates have a failed knot
    d = sol.device_solver(B, max_iter=1, obj_tol=0.0)
    d.init(x1t, Ut)
    X0 = d.get()[0].clone()
    d.iterate(1)
    K, k = d.get(gains=True)[3:]
    Xc, Uc, cst = sol.forward(x1t, X0, Ut, K, k)
    nbad = int(((cst & 1) == 0).any(0).sum().item())
    assert nbad > 0, "no candidate of the first forward pass has a failed knot: the check does not exercise the early exit"
    return nbad


# ---- od_ilqr_* against the INDEPENDENT numpy AL-iLQR of oracle/ilqr_np.py (driven by the oracle's dynamics), decision by decision ---------
def acrobot_example(lib, device, B, T=100, h=0.05, mode="impact"):
    """examples/acrobot.jl:15-111: swing-up, x1 = 0, x_T = [pi, 0, pi, 0] by augmented Lagrangian, 1/2 0.1 |v1|^2 + 1/2 u^2; trajectory b
    starts from controls 1e-3 randn(seed 1 + b) (:90-91: trajectory 0 is the example's).  mode "impact": joint limits (:19-23, kappa
    1e-4 / 1e-3); "nominal": the mode the file ends up in (:11-12,24-27: acrobot_nominal, no joint limits, kappa = 1)"""
    import math
    import optimization_dynamics_amd as od
    if mode == "impact":
        im = od.ImplicitDynamics(od.acrobot_impact, h, r_tol=1e-8, kappa_eval_tol=1e-4, kappa_grad_tol=1e-3, device=device, lib=lib)
    else:
        im = od.ImplicitDynamics(od.acrobot_nominal, h, r_tol=1e-8, kappa_eval_tol=1.0, kappa_grad_tol=1.0, device=device, lib=lib)
    I2 = np.eye(2)
    Q = 0.1 / h ** 2 * np.block([[I2, -I2], [-I2, I2]])
    xT = np.array([math.pi, 0.0, math.pi, 0.0])
    obj = IL.QuadraticObjective(Q, np.eye(1), Q, x_ref=np.zeros(4), goal_idx=[0, 1, 2, 3], goal=xT, device=device)
    U0 = np.stack([1e-3 * np.random.default_rng(1 + b).normal(size=(1, T)) for b in range(B)], axis=-1)
    return im, obj, np.zeros((4, B)), U0


ORACLE_CASES = {
    # name: (builder, T, solver options, J tolerance while the decisions agree, iterations that must agree (None: all), final tolerance)
    "cartpole": dict(T=15, kw=dict(max_iter=8, max_al_iter=2, obj_tol=1e-7, con_tol=1e-4), tolJ=1e-8, need=None),
    "cartpole_constrained": dict(T=25, kw=dict(max_iter=10, max_al_iter=4, obj_tol=1e-7, con_tol=1e-4), tolJ=1e-8, need=None),
    # the swing-up passes through joint-limit impacts: a contact-mode switch amplifies the 1e-12 between two implementations of one
    # solve to 1e-5 in a later cost (both stay valid solves of the task: same outcome)
    "acrobot": dict(T=100, kw=dict(max_iter=50, max_al_iter=20, obj_tol=1e-5, con_tol=1e-3), tolJ=1e-8, need=30),
    # examples/acrobot.jl AS SHIPPED (`:nominal`: no joint limits, smooth dynamics)
    "acrobot_nominal": dict(T=100, kw=dict(max_iter=50, max_al_iter=20, obj_tol=1e-5, con_tol=1e-3), tolJ=1e-8, need=None),
    "rocket": dict(T=20, kw=dict(max_iter=10, max_al_iter=1, obj_tol=1e-7, con_tol=1e-4), tolJ=1e-8, need=None),
    # with the thrust-cone projection on the path.  Round 5 compared against the LITERAL oracle projection, whose eps_min = 0 line search
    # compares rounding noise: costs agreed to kappa_tol level (2e-3) until an Armijo test landed on the other side, 8-10 of 10 iterations.
    # Round 6: device and oracle both complete the projection's rounding-decided places as exact arithmetic has them
    # (parity_checks.check_rocket_sweep, oracle `exact_boundary`): costs to 1e-8 while the decisions agree, and they agree in every
    # iteration on all but the odd problem whose projection is ill-conditioned in its own rounding (`all_frac`: the share of problems
    # that must agree to the end; the rest for `need` leading iterations)
    "rocket_projected": dict(T=20, kw=dict(max_iter=10, max_al_iter=1, obj_tol=1e-7, con_tol=1e-4), tolJ=1e-8, need=3, all_frac=0.74, exact_boundary=True),
}


def check_against_numpy_oracle(oracle, lib, device, case, B=8, seed=1):
    """od_ilqr_solve on B problems against oracle/ilqr_np.py::solve on each of them: the accepted step index, the regularisation after
    the iteration, the penalty and the cost, iteration by iteration (od_ilqr_get_trace / od_ilqr_get_history), then the final
    trajectory and flags.  -> statistics (asserted here)"""
    from oracle import ilqr_np as N
    from optimization_dynamics_amd import rocket as rk
    cfg = ORACLE_CASES[case]
    if cfg.get("exact_boundary") and not getattr(check_against_numpy_oracle, "_inside", False):
        oracle.lib().od_oracle_set_exact_boundary(1)
        check_against_numpy_oracle._inside = True
        try:
            return check_against_numpy_oracle(oracle, lib, device, case, B=B, seed=seed)
        finally:
            check_against_numpy_oracle._inside = False
            oracle.lib().od_oracle_set_exact_boundary(0)
    T, kw = cfg["T"], cfg["kw"]
    roll = None
    if case == "cartpole":
        im, obj, x1, U0 = cartpole_problem(lib, device, B, T, seed=seed)
        step, lin = N.mechanical_dynamics(P.make_sim(oracle, "cartpole_friction"))
    elif case == "cartpole_constrained":
        im, obj, x1, U0 = constrained_problem(lib, device, "cartpole", B, T, seed=seed)
        step, lin = N.mechanical_dynamics(P.make_sim(oracle, "cartpole_friction"))
    elif case == "acrobot":
        im, obj, x1, U0 = acrobot_example(lib, device, B, T)
        step, lin = N.mechanical_dynamics(oracle.make_sim("acrobot_impact", 0.05, kappa_tol=1e-4, kappa_grad_tol=1e-3))
    elif case == "acrobot_nominal":
        im, obj, x1, U0 = acrobot_example(lib, device, B, T, mode="nominal")
        step, lin = N.mechanical_dynamics(oracle.make_sim("acrobot_nominal", 0.05, kappa_tol=1.0, kappa_grad_tol=1.0))
    else:
        im, obj, x1, U0 = rocket_problem(lib, device, B, T, seed=seed)
        if case == "rocket":
            im = rk.RocketDynamics(im.info, project=False)
        step, lin, roll = N.rocket_dynamics(0.05, 12.5, project=(case == "rocket_projected"))
    sol = IL.ILQR(im, obj, T)
    X, U, J, hist = sol.solve(torch.tensor(x1, device=device), torch.tensor(U0, device=device), **kw)
    d = sol._dev
    sel, reg, rho = [a.cpu().numpy() for a in d.trace()]
    H = torch.stack(hist).cpu().numpy() if len(hist) else np.zeros((0, B))
    flags, viol, pen = [a.cpu().numpy() for a in d.status()]
    Xn, Un = X.cpu().numpy(), U.cpu().numpy()
    f = lambda t: None if t is None else t.cpu().numpy()
    stats = dict(case=case, problems=B, horizon=T, agreeing_iterations=[], iterations_device=[], iterations_oracle=[], cost_rel_max_while_agreeing=0.0,
                 final_cost_rel=[], final_state_diff=[], flags_equal=0)
    for b in range(B):
        p = N.Problem(step, lin, f(obj.Q), f(obj.R), f(obj.QT), f(obj.x_ref), goal_idx=f(obj.goal_idx), goal=f(obj.goal), stage=obj.stage, terminal=obj.terminal)
        r = N.solve(p, x1[:, b], U0[:, :, b].T, alphas=tuple(sol.alphas.cpu().tolist()), reg0=sol.reg, c1=sol.c1, rollout_fn=roll, **kw)
        L = r["log"]
        rows = [i for i in range(sel.shape[0]) if sel[i, b] != -2]              # the iterations trajectory b took part in
        nag = 0
        for l, i in zip(L, rows):
            same = l["step"] == sel[i, b] and abs(l["reg"] - reg[i, b]) <= 1e-12 * reg[i, b] and abs(l["rho"] - rho[i, b]) <= 1e-12 * max(1.0, rho[i, b])
            eJ = abs(l["J"] - H[i, b]) / max(1.0, abs(l["J"]))
            if not (same and eJ <= cfg["tolJ"]):
                break
            nag += 1
            stats["cost_rel_max_while_agreeing"] = max(stats["cost_rel_max_while_agreeing"], float(eJ))
        stats["agreeing_iterations"].append(nag); stats["iterations_device"].append(len(rows)); stats["iterations_oracle"].append(len(L))
        fl_o = (1 if r["done"] else 0) | (2 if r["al_done"] else 0)
        stats["flags_equal"] += int(fl_o == int(flags[b]))
        eX = float(np.abs(r["X"].T - Xn[:, :, b]).max())
        stats["final_state_diff"].append(eX)
        Jd, Jo = float(J[b].item()), r["J"]
        stats["final_cost_rel"].append(abs(Jd - Jo) / max(1e-12, abs(Jo)))
        if cfg["need"] is None:
            # every decision of every iteration, and the same trajectory at the end
            assert nag == len(L) == len(rows), (case, b, nag, len(L), len(rows))
            assert fl_o == int(flags[b]) and eX < 1e-6 * max(1.0, np.abs(r["X"]).max()), (case, b, fl_o, int(flags[b]), eX)
            assert abs(r["violation"] - viol[b]) <= 1e-8 + 1e-6 * abs(r["violation"]) and abs(r["rho"] - pen[b]) <= 1e-9 * max(1.0, pen[b])
        else:
            # the leading iterations decision by decision; then two valid solves of the same task: same outcome
            assert nag >= min(cfg["need"], len(L)), (case, b, nag)
            assert r["al_done"] == bool(flags[b] & 2) or not obj.constrained, (case, b)
            # (the swing-up has several local solutions -- one more pump of the lower link costs ~60 % more --, and which one a solve ends
            # in is decided after the paths have parted: same constraint flag, objectives of one order, iteration counts of one order)
            assert 0.4 * abs(Jo) <= abs(Jd) <= 2.5 * abs(Jo), (case, b, Jd, Jo)
            assert abs(len(L) - len(rows)) <= max(3, 0.5 * len(L)), (case, b, len(L), len(rows))
    if cfg.get("all_frac"):
        full = sum(int(a == o == d_) for a, o, d_ in zip(stats["agreeing_iterations"], stats["iterations_oracle"], stats["iterations_device"]))
        stats["problems_agreeing_in_every_decision"] = full
        assert full >= cfg["all_frac"] * B, (case, "problems agreeing in every decision", full, B)
    return stats


# ---- examples/planar_push.jl with GB = true: the gradient bundle as the solver's linearisation (od_ilqr_set_gradient_bundle) ------------------
def check_bundle_linearisation(oracle, lib, device, mode="rotate", B=4, N=50, n_oracle=1, need=1.0):
    """examples/planar_push.jl:15,22,29-30 with GB = true (GradientBundle(planarpush, N = 50, eps = 1e-4)) through od_ilqr_solve:
    (1) against the same loop composed on the host from the PUBLIC entry points (od_bundle_grad for every linearisation, decisions in
        torch): costs iteration by iteration, iteration counts, trajectories;
    (2) the first `n_oracle` problems against oracle/ilqr_np.py::solve with the ORACLE's bundle (its own N + 1 steps and fit) as
        linearisation, decision by decision for as long as the two agree -- a zero-order fit divides differences of steps by
        eps = 1e-4, so what two implementations of a step leave at r_tol = 1e-8 is ~1e-4 in the fitted Jacobians and an Armijo test
        can fall the other way early; then: same outcome;
    (3) the example's outcome: goal to con_tol, controls inside their limits."""
    from oracle import ilqr_np as N_
    from optimization_dynamics_amd import gradient_bundle as gbm
    import optimization_dynamics_amd as od
    im, obj, x1, U0, xT, T, opts = planar_push_example(lib, device, mode, B)
    gb = gbm.GradientBundle(od.planarpush, N=N, eps=1.0e-4, seed=3)
    alphas = tuple(2.0 ** -i for i in range(17))
    x1t, Ut = torch.tensor(x1, device=device), torch.tensor(U0, device=device)
    sol = IL.ILQR(im, obj, T, alphas=alphas, bundle=gb)
    X, U, J, hist = sol.solve(x1t, Ut, **opts)
    d = sol._dev
    info = d.info()
    fl, viol, rho = [a.cpu().numpy() for a in d.status()]
    sel, reg, rh = [a.cpu().numpy() for a in d.trace()]
    H = torch.stack(hist).cpu().numpy()
    okc = viol < opts["con_tol"]
    assert okc.mean() >= need, (okc.mean(), viol.max())
    assert (U[:, :, torch.tensor(okc, device=U.device)].abs() <= 5.0 + opts["con_tol"]).all()
    assert (im.rollout(x1t, U, grads=False)[0] - X).abs().max().item() < 1e-9
    # the implicit gradients give another path: the bundle really is what linearised
    Xi, Ui, Ji, hi = IL.ILQR(im, obj, T, alphas=alphas).solve(x1t, Ut, **opts)
    assert (torch.stack(hi)[0] - torch.stack(hist)[0]).abs().max().item() > 1e-9
    # (1) the host-composed loop with od_bundle_grad
    ref = IL.ILQR(im, obj, T, alphas=alphas, bundle=gb).solve_stepwise(x1t, Ut, **opts)
    assert len(ref[3]) == len(hist) == info.iterations, (len(ref[3]), len(hist), info.iterations)
    for i, (ja, jb) in enumerate(zip(hist, ref[3])):
        assert ((ja - jb).abs() <= 1e-9 * jb.abs().clamp(min=1.0)).all(), (i, (ja - jb).abs().max().item())
    assert (X - ref[0]).abs().max().item() < 1e-6 and (U - ref[1]).abs().max().item() < 1e-6
    stats = dict(mode=mode, problems=B, samples=N, iterations=int(info.iterations), rounds=int(info.al_iterations), fraction_at_con_tol=float(okc.mean()),
                 violation_max=float(viol.max()), agreeing_iterations=[], iterations_oracle=[], objective=[float(v) for v in J[: min(B, 4)].cpu().numpy()])
    # (2) the numpy oracle with the oracle's bundle
    sim = oracle.make_sim("planar_push", 0.1, kappa_tol=1e-4, kappa_grad_tol=1e-2)
    step, lin = N_.bundle_dynamics(sim, gb.eta)
    f = lambda t: None if t is None else t.cpu().numpy()
    for b in range(min(B, n_oracle)):
        p = N_.Problem(step, lin, f(obj.Q), f(obj.R), f(obj.QT), f(obj.x_ref), goal_idx=f(obj.goal_idx), goal=f(obj.goal), stage=obj.stage, terminal=obj.terminal)
        r = N_.solve(p, x1[:, b], U0[:, :, b].T, alphas=alphas, reg0=sol.reg, c1=sol.c1, **opts)
        L = r["log"]
        rows = [i for i in range(sel.shape[0]) if sel[i, b] != -2]
        nag = 0
        for l, i in zip(L, rows):
            same = l["step"] == sel[i, b] and abs(l["reg"] - reg[i, b]) <= 1e-12 * reg[i, b] and abs(l["rho"] - rh[i, b]) <= 1e-12 * max(1.0, rh[i, b])
            if not (same and abs(l["J"] - H[i, b]) <= 1e-4 * max(1.0, abs(l["J"]))):
                break
            nag += 1
        stats["agreeing_iterations"].append(nag); stats["iterations_oracle"].append(len(L))
        # (problem 0 is the example's own start; a perturbed start may meet an Armijo test within the fit's noise at once: compared by outcome)
        assert nag >= min(3 if b == 0 else 1, len(L)), (b, nag, len(L))
        stats.setdefault("same_outcome", []).append(bool(r["al_done"] == bool(fl[b] & 2)))
        if b == 0 or (r["al_done"] and (fl[b] & 2)):
            assert r["al_done"] == bool(fl[b] & 2), (b, r["al_done"], fl[b])
            assert abs(len(L) - len(rows)) <= max(3, len(L) // 2), (len(L), len(rows))
            assert abs(r["J"] - J[b].item()) <= 0.2 * abs(r["J"]), (r["J"], J[b].item())
    print("planar push %s with GB = true (N = %d), %d problem(s): %d iterations, %d multiplier rounds, %.0f %% at con_tol, max violation %.2e; "
          "oracle agrees on %s of %s iterations" % (mode, N, B, info.iterations, info.al_iterations, 100 * okc.mean(), viol.max(), stats["agreeing_iterations"], stats["iterations_oracle"]))
    return stats


# ---- examples/hopper.jl AS SHIPPED (the initial configurations optimised through a first stage of its own dimensions) on the device ------
GAITS = {1: (1.0e-1, 1.0e-1), 2: (1.0, 1.0), 3: (1.0e-3, 1.0e-1)}      # (r_cost, q_cost), examples/hopper.jl:190-203


def hopper_example_full(lib, device, B, seed=1, gait=1):
    """examples/hopper.jl:12-13,42-50,176-290, GAIT 1: T = 21, h = 0.05; u_1 = [u; theta] with theta = [q1; q2] the two initial
    configurations (od_ilqr_set_parameter_stage: slot 0 of the trajectory), obj1 / objt / objT (:207-226), stage1_con (control limits,
    q1 fixed, foot positions: the generated `hopper_foot`), staget_con (control limits), terminal_con (:256-262, couples x_T with theta).
    Problem 0 starts from the standing controls (:270), the others from perturbed ones."""
    HP = dict(foot_radius=0.05, gravity=9.81, mass_body=3.0)
    h, T = 0.05, 20
    im = P.make_im("hopper", lib, device)
    r = HP["foot_radius"]
    q1 = np.array([0.0, 0.5 + r, 0.0, 0.5]); q_ref = np.array([0.5, 0.75 + r, 0.0, 0.25])
    x1v, x_ref = np.concatenate([q1, q1]), np.concatenate([q_ref, q_ref])
    w = np.array([1.0, 10.0, 1.0, 10.0] * 2)
    r_cost, q_cost = GAITS[gait]
    obj = IL.QuadraticObjective(q_cost * np.diag(w), r_cost * np.eye(2), np.eye(8), x_ref=x_ref, device=device)
    Cs = np.zeros((4, 8)); Ds = np.vstack([-np.eye(2), np.eye(2)]); ds = np.full(4, 10.0)
    obj.set_constraints(stage=(Cs, Ds, ds, 4))
    Ctx = np.zeros((8, 8)); Cth = np.zeros((8, 8)); dt = np.zeros(8)
    Ctx[0, 0], Cth[0, 0], dt[0] = -1.0, 1.0, -0.5             # x_travel - (x[1] - theta[1]) <= 0
    Ctx[1, 4], Cth[1, 4], dt[1] = -1.0, 1.0, -0.5
    for k, i in enumerate([1, 2, 3, 5, 6, 7]):
        Ctx[2 + k, i], Cth[2 + k, i] = 1.0, -1.0
    w_theta = np.array([1.0e-1] * 4 + [1.0e-5] * 4)              # obj1's weights on u[3:10] (:209)
    c0 = 0.5 * float((x1v - x_ref) @ (w * (x1v - x_ref)))        # obj1's cost of the fixed x_1
    obj.set_parameter_stage(w_theta, constraint="hopper_foot", p=x1v, terminal=(Ctx, Cth, dt, 2), cost_const=c0)
    U0 = np.zeros((2, T, B)); U0[1] = HP["gravity"] * HP["mass_body"] * 0.5 * h
    U0[:, :, 1:] += 1e-2 * np.random.default_rng(seed).normal(size=(2, T, B - 1))
    x1 = np.repeat(x1v[:, None], B, axis=1)
    opts = dict(max_iter=10, max_al_iter=15, con_tol=1.0e-3, obj_tol=1.0e-3, rho_init=1.0, rho_scale=10.0)
    return im, obj, x1, U0, x1v, T, opts


def check_hopper_example_full(oracle, lib, device, B=1, n_oracle=1, gait=1):
    """examples/hopper.jl as shipped through od_ilqr_solve -- parameter stage, generated nonlinear constraint, coupled terminal rows --
    against oracle/ilqr_np.py::solve_stages on the reference's own formulation (stages of dimensions 8 / 10 -> 16 and 16 / 2 -> 16,
    constraint functions as the example writes them) with the oracle's dynamics: the decisions of every iteration on the first
    `n_oracle` problems (accepted step index, regularisation, penalty, merit), the optimised initial configurations, the
    constraints to con_tol on every problem"""
    from oracle import ilqr_np as N
    im, obj, x1, U0, x1v, T, opts = hopper_example_full(lib, device, B, gait=gait)
    alphas = tuple(2.0 ** -i for i in range(17))
    sol = IL.ILQR(im, obj, T, alphas=alphas)
    X, U, J, hist = sol.solve(torch.tensor(x1, device=device), torch.tensor(U0, device=device), **opts)
    d = sol._dev
    info = d.info()
    fl, viol, rho = [a.cpu().numpy() for a in d.status()]
    sel, reg, rh = [a.cpu().numpy() for a in d.trace()]
    H = torch.stack(hist).cpu().numpy()
    Xn, Un = X.cpu().numpy(), U.cpu().numpy()
    assert (viol < opts["con_tol"]).all() and ((fl & 2) != 0).all(), (viol.max(), fl)
    assert (np.abs(Un) <= 10.0 + opts["con_tol"]).all()
    th, xT = Xn[:, 0], Xn[:, -1]
    assert (xT[0] - th[0] >= 0.5 - 1e-3).all() and (xT[4] - th[4] >= 0.5 - 1e-3).all()          # half a metre further ...
    keep = [1, 2, 3, 5, 6, 7]
    assert np.abs(xT[keep] - th[keep]).max() < 1e-3                                              # ... in the pose the gait started from
    foot = lambda q: np.array([q[0] + q[3] * np.sin(q[2]), q[1] - q[3] * np.cos(q[2])])
    assert np.abs(th[:4] - x1v[:4, None]).max() < 1e-3 and np.abs(foot(th[4:8]) - foot(x1v[4:8])[:, None]).max() < 1e-3
    assert np.abs(th[4:8] - x1v[4:8, None]).max() > 1e-2, "the second initial configuration was not optimised"
    # the returned trajectory is the rollout of its controls from its theta
    Xr = im.rollout(X[:, 0].contiguous(), U, grads=False)[0]
    assert (Xr - X).abs().max().item() < 1e-9
    # od_ilqr_get's gains with a parameter stage: feedback on the mechanical state (none at knot 0), feed-forward of the mechanical controls
    Kg, kg = [a.cpu().numpy() for a in d.get(gains=True)[3:]]
    assert Kg.shape == (2 * 8, T, B) and kg.shape == (2, T, B) and np.isfinite(Kg).all() and np.isfinite(kg).all()
    assert (Kg[:, 0] == 0).all() and (np.abs(Kg[:, 1:]).max(axis=0) > 0).all()
    stats = dict(gait=gait, problems=B, iterations=int(info.iterations), rounds=int(info.al_iterations), violation_max=float(viol.max()),
                 objective=[float(v) for v in J[: min(B, 4)].cpu().numpy()], agreeing_iterations=[], iterations_oracle=[])
    sim = oracle.make_sim("hopper", 0.05, kappa_tol=1e-4, kappa_grad_tol=1e-3, friction=[0.5, 0.5])
    for b in range(min(B, n_oracle)):
        st, objT, conT, nti, x1o, U0o = N.hopper_gait_stages(sim, r_cost=GAITS[gait][0], q_cost=GAITS[gait][1])
        U0o = [np.concatenate([U0[:, 0, b], x1v])] + [U0[:, t, b] for t in range(1, T)]
        r = N.solve_stages(st, objT, conT, nti, x1o, U0o, alphas=alphas, reg0=sol.reg, c1=sol.c1, **{k: v for k, v in opts.items()})
        L = r["log"]
        rows = [i for i in range(sel.shape[0]) if sel[i, b] != -2]
        nag = 0
        for l, i in zip(L, rows):
            same = l["step"] == sel[i, b] and abs(l["reg"] - reg[i, b]) <= 1e-12 * reg[i, b] and abs(l["rho"] - rh[i, b]) <= 1e-12 * max(1.0, rh[i, b])
            if not (same and abs(l["J"] - H[i, b]) <= 1e-7 * max(1.0, abs(l["J"]))):
                break
            nag += 1
        stats["agreeing_iterations"].append(nag); stats["iterations_oracle"].append(len(L))
        # (a hopping gait switches contact modes: the 1e-12 between two implementations of one step can grow, see ORACLE_CASES)
        # problem 0 is the example itself (standing controls); the perturbed starts may meet an Armijo test or a contact switch within
        # rounding of its threshold early (device kernels and oracle associate their sums differently): the paths are then compared by
        # their outcome below, the number of agreeing iterations is recorded
        assert nag >= min(10 if b == 0 else 2, len(L)), (b, nag, len(L), len(rows))
        assert r["al_done"] and abs(len(L) - len(rows)) <= max(3, len(L) // 4), (len(L), len(rows))
        assert abs(r["J"] - float(J[b].item())) <= (1e-6 if nag == len(L) == len(rows) else 5e-2) * abs(r["J"]), (r["J"], float(J[b].item()))       # (the merit: J carries the multiplier terms)
        th_o = r["U"][0][2:]
        assert np.abs(th_o - th[:, b]).max() < 5e-2, (th_o, th[:, b])
        stats["theta_device"] = [float(v) for v in th[:, b]]; stats["theta_oracle"] = [float(v) for v in th_o]
        stats["objective_oracle"] = float(r["objective"])
    return stats


def check_converged_neighbours_do_not_disturb(lib, device, mode, B=7, which="hopper"):
    """a batch whose problems converge at DIFFERENT iterations under a cooperative rollout kernel (od_set_cooperative mode 2: 16 lanes per
    problem, 3: 8 lanes -- two problems share a DPP row): once a problem has converged its lanes leave the kernels (per-trajectory
    predicate) while its row partner goes on alone.  Every problem solved alone (a batch of one: no partner at all) must give the same
    trajectory, cost, flags and iteration history bit for bit -- a half-active row computes what a full one does
    (csrc/od_model_tu.inc::k_rollout_policy_coop3)."""
    if which == "hopper":
        im, obj, x1, U0, x1v, T, opts = hopper_example(lib, device, B, seed=3)
        U0[:, :, 2::2] += 0.3 * np.random.default_rng(9).normal(size=U0[:, :, 2::2].shape)     # every other problem starts far away: converges later
        alphas = tuple(2.0 ** -i for i in range(17))
    else:
        raise NotImplementedError(which)
    im.set_cooperative(mode)
    assert lib.cdll.od_uses_cooperative(im._h, B * len(alphas)) == 1
    sol = IL.ILQR(im, obj, T, alphas=alphas)
    x1t, Ut = torch.tensor(x1, device=device), torch.tensor(U0, device=device)
    X, U, J, hist = sol.solve(x1t, Ut, **opts)
    sel, reg, rho = sol._dev.trace()
    fl, viol, pen = sol._dev.status()
    its = [(sel[:, b] != -2).sum().item() for b in range(B)]
    assert len(set(its)) >= 2, ("the problems should converge at different iterations", its)
    for b in range(B):
        one = IL.ILQR(im, obj, T, alphas=alphas)
        X1, U1, J1, h1 = one.solve(x1t[:, b:b + 1].contiguous(), Ut[:, :, b:b + 1].contiguous(), **opts)
        s1, r1, p1 = one._dev.trace()
        f1, v1, q1 = one._dev.status()
        n1 = s1.shape[0]
        assert torch.equal(X1[:, :, 0], X[:, :, b]) and torch.equal(U1[:, :, 0], U[:, :, b]) and torch.equal(J1[0], J[b]), (mode, b)
        assert f1[0] == fl[b] and torch.equal(v1[0], viol[b])
        rows = (sel[:, b] != -2).nonzero().reshape(-1)
        assert rows.numel() == (s1[:, 0] != -2).sum().item()
        assert torch.equal(sel[rows, b], s1[s1[:, 0] != -2, 0]) and torch.equal(reg[rows, b], r1[s1[:, 0] != -2, 0]), (mode, b)
    im.set_cooperative(0)
    return its
