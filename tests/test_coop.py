"""Cooperative solve pass (csrc/od_coop.h: one problem per 16-lane DPP row, contacts and cones in their own lanes).
CPU tier: the row is emulated lane by lane (RowEmu, tests/host_emu) -- the same block algebra, routing tables and
reductions as on the device, so the lane bookkeeping is covered without a GPU.  GPU tier: the DPP instructions."""
import numpy as np
import pytest
import torch

import parity_checks as P
import workloads as W

COOP_MODELS = ["hopper", "acrobot_impact", "cartpole_friction"]


def test_models_with_cooperative_kernels(emu_lib):
    from optimization_dynamics_amd import models
    hop, acro = P.make_im("hopper", emu_lib, "cpu"), P.make_im("acrobot_impact", emu_lib, "cpu")
    uses = emu_lib.cdll.od_uses_cooperative
    assert uses(hop._h, 4096) == 1 and uses(hop._h, 16384) == 1 and uses(hop._h, 16385) == 0      # automatic: small batches (16 lanes per problem up to 4096, 8 lanes up to 16 384)
    assert uses(acro._h, 1024) == 1 and uses(acro._h, 4096) == 1 and uses(acro._h, 4097) == 0      # two contacts: up to 4096
    acro.set_cooperative(2); assert uses(acro._h, 65536) == 1
    hop.set_cooperative(1); assert uses(hop._h, 64) == 0
    hop.set_cooperative(0); hop.set_launch_config(16, 4); assert uses(hop._h, 64) == 0               # an explicit mapping wins
    im = P.make_im("planar_push", emu_lib, "cpu")
    im.set_cooperative(2)                     # 3-d cones: the 8-lane form (od_coop3.h, tests/test_coop3.py)
    assert uses(im._h, 64) == 1
    im = P.make_im("acrobot_nominal", emu_lib, "cpu")
    im.set_cooperative(2)                     # no cones, no cooperative kernels: silently the usual ones
    assert uses(im._h, 64) == 0
    im = P.make_im("cartpole_frictionless", emu_lib, "cpu")
    im.set_cooperative(2)
    X, U = W.knots("cartpole_frictionless", 32, seed=3)
    a = im.step(torch.tensor(X), torch.tensor(U))[0]
    im.set_cooperative(1)
    assert torch.equal(a, im.step(torch.tensor(X), torch.tensor(U))[0])
    assert emu_lib.cdll.od_set_cooperative(im._h, 4) == -1


@pytest.mark.parametrize("name", COOP_MODELS)
def test_coop_matches_lane_per_problem_emulated(emu_lib, name):
    P.check_coop_vs_serial(emu_lib, "cpu", name, 1024)


@pytest.mark.parametrize("B", [3, 1027, 2051])
def test_coop_rows_per_wavefront_emulated(emu_lib, B):
    """batches of <= 1024, <= 2048 and more problems run 1, 2 and 4 rows per wavefront: same results, problem by problem"""
    X, U = W.knots("hopper", 2051, seed=83)
    im = P.make_im("hopper", emu_lib, "cpu")
    im.set_cooperative(2)
    full = im.step_grad(torch.tensor(X), torch.tensor(U))
    part = im.step_grad(torch.tensor(np.ascontiguousarray(X[:, :B])), torch.tensor(np.ascontiguousarray(U[:, :B])))
    for a, b in zip(full, part):
        assert torch.equal(a[..., :B], b)


@pytest.mark.parametrize("name", COOP_MODELS)
def test_coop_against_oracle_emulated(oracle, emu_lib, name):
    # (automatic mode picks the cooperative kernels for hopper batches this small; the acrobot has them on request only)
    old = P.make_im

    def forced(*a, **k):
        im = old(*a, **k)
        im.set_cooperative(2)
        return im
    P.make_im = forced
    try:
        P.check_step_grad(oracle, emu_lib, "cpu", name, 512)
    finally:
        P.make_im = old


def test_coop_rollout_emulated(oracle, emu_lib):
    P.check_coop_rollout(oracle, emu_lib, "cpu", 32, 30)


def test_coop_finite_undercut_emulated(oracle, emu_lib):
    """two separate passes (eval, grad) through the cooperative kernel: status / iteration merge"""
    name = "hopper"
    X, U = W.knots(name, 128, seed=61)
    im = P.make_im(name, emu_lib, "cpu", options=dict(undercut=5.0))
    im.set_cooperative(2)
    a = [t.numpy() for t in im.step_grad(torch.tensor(X), torch.tensor(U))]
    im.set_cooperative(1)
    b = [t.numpy() for t in im.step_grad(torch.tensor(X), torch.tensor(U))]
    assert np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4])
    assert np.abs(a[0] - b[0]).max() < 1e-8


def test_coop_policy_rollout_emulated(emu_lib):
    P.check_coop_policy_rollout(emu_lib, "cpu")


@pytest.mark.gpu
def test_coop_policy_rollout(gpu_lib):
    P.check_coop_policy_rollout(gpu_lib, "cuda:0", B=16, T=20)


@pytest.mark.gpu
@pytest.mark.parametrize("name", COOP_MODELS)
def test_coop_matches_lane_per_problem(gpu_lib, name):
    P.check_coop_vs_serial(gpu_lib, "cuda:0", name, 8192)


@pytest.mark.gpu
def test_coop_rollout(oracle, gpu_lib):
    P.check_coop_rollout(oracle, gpu_lib, "cuda:0", 96, 40)


@pytest.mark.gpu
@pytest.mark.parametrize("B", [1, 3, 5, 63, 257, 1025, 2047, 2051])
def test_coop_ragged_batches(gpu_lib, B):
    """rows of a wavefront / wavefronts of a workgroup without a problem, with 1, 2 and 4 rows per wavefront"""
    X, U = W.knots("hopper", B, seed=81)
    im = P.make_im("hopper", gpu_lib, "cuda:0")
    Xd, Ud = torch.tensor(X, device="cuda:0"), torch.tensor(U, device="cuda:0")
    im.set_cooperative(2); a = im.step_grad(Xd, Ud)
    im.set_cooperative(1); b = im.step_grad(Xd, Ud)
    # (a knot whose violation sits within rounding of a tolerance may take one iteration more in one of the kernels)
    same = (a[3] == b[3]) & (a[4] == b[4]).all(0)
    assert same.float().mean().item() >= 0.999 and (a[0] - b[0])[:, same].abs().max().item() < 1e-8


EDGE_OPTIONS = [
    dict(max_iter=0), dict(max_iter=1), dict(max_iter=3), dict(max_ls=1), dict(max_ls=2),
    dict(kappa_grad_tol=1e-6), dict(kappa_grad_tol=1e-2, kappa_eval_tol=1e-6), dict(r_tol=1e-3), dict(r_tol=1e-13),
    dict(eps_min=0.0), dict(undercut=5.0), dict(gamma_reg=0.0), dict(kappa_reg=1.0),
]


# lowest agreement rate (status and both iteration counts equal, 96 knots) over 20 seeds x every cooperative model / form on the
# MI355X (tools/edge_rates.py -> profiles/r3_edge_option_agreement.json); every option not listed: 1.0 on every seed
MEASURED_MIN_AGREEMENT = {"r_tol=1e-13": 0.948, "eps_min=0": 0.927}


def min_agreement(kw, n=96):
    key = ",".join("%s=%g" % kv for kv in kw.items())
    # (n knots per call: one standard deviation of a rate p is sqrt(p (1 - p) / n) -- 0.022 for 0.95 and 96 knots, 0.027 for 64 -- and
    # the bar sits three of them below the measured minimum; seed offset 22 of the emulated tier gives 88 of 96 for r_tol = 1e-13,
    # seed offset 69 of the GPU tier 56 of 64 in the 8-lane form)
    if key not in MEASURED_MIN_AGREEMENT:
        return 0.98
    p = MEASURED_MIN_AGREEMENT[key]
    return p - 3.0 * (p * (1.0 - p) / n) ** 0.5


def _edge_check(lib, device, name, kw):
    """cooperative against lane-per-problem kernels under unusual solver options: same status and iteration counts,
    states equal to rounding -- including the solves that stop at max_iter or take no iteration at all"""
    X, U = W.knots(name, 96, seed=7)
    Xd, Ud = torch.tensor(X, device=device), torch.tensor(U, device=device)
    out = []
    for mode in (1, 2):
        im = P.make_im(name, lib, device)
        im.set_options(**kw)
        im.set_cooperative(mode)
        out.append([t.cpu().numpy() for t in im.step_grad(Xd, Ud)] + [im.step(Xd, Ud)[0].cpu().numpy()])
    ref, got = out
    same = (ref[3] == got[3]) & (ref[4] == got[4]).all(0)
    # (a residual tolerance at rounding level: the two association orders reach it an iteration apart on some knots)
    noise_level = kw.get("r_tol", 1) < 1e-10 or kw.get("eps_min", 1) == 0.0      # (or tau = 1: the acceptance test compares noise)
    assert same.mean() >= min_agreement(kw), (kw, same.mean())
    fin = np.isfinite(ref[0]).all(0) & np.isfinite(got[0]).all(0)
    # (iterating past the attainable precision ends some solves on a singular factor: non-finite in both kernels alike)
    assert (fin | ~same).all() or fin.mean() > (0.8 if noise_level else 0.95)
    bound = 1e-4 if noise_level else 1e-6
    e_all = np.abs(ref[0] - got[0]).max(0)
    # a knot on which the two kernels land apart is a failure unless the lane-per-problem kernel does not reproduce
    # ITSELF there: 16 copies of the knot with inputs perturbed by 1e-13 / 1e-11 relative (several roots within reach of a long
    # Newton path -- tests/parity_checks.py::comparable_states does the same with the oracle)
    sel = same & fin
    im = P.make_im(name, lib, device)
    im.set_options(**kw)
    im.set_cooperative(1)
    for i in np.nonzero(sel & ~(e_all < bound))[0]:
        rng = np.random.default_rng(int(i))
        for eps_p in (1e-13, 1e-11):
            Xp = X[:, [i]] * (1 + eps_p * rng.normal(size=(X.shape[0], 16)))
            Up = U[:, [i]] * (1 + eps_p * rng.normal(size=(U.shape[0], 16)))
            Dp = im.step(torch.tensor(Xp, device=device), torch.tensor(Up, device=device))[0].cpu().numpy()
            # (... by as much as the two kernels differ: a scatter of a third of the difference or more explains it)
            if np.ptp(Dp, axis=1).max() > max(10 * bound if eps_p < 1e-12 else 0.0, e_all[i] / 3.0):
                sel[i] = False
    assert (same & fin & ~sel).sum() <= 1
    e = e_all[sel]
    assert np.median(e) < 1e-12 and e.max() < bound, (kw, np.median(e), e.max())
    assert np.array_equal(np.isnan(ref[0]), np.isnan(got[0])) or same.mean() < 1.0
    e5 = np.abs(ref[5] - got[5])[:, sel].max(0)
    assert e5.max() < bound, kw


@pytest.mark.parametrize("kw", EDGE_OPTIONS, ids=lambda d: ",".join("%s=%g" % kv for kv in d.items()))
def test_coop_edge_options_emulated(emu_lib, kw):
    for name in ("hopper", "cartpole_friction"):
        _edge_check(emu_lib, "cpu", name, kw)


@pytest.mark.gpu
@pytest.mark.parametrize("kw", EDGE_OPTIONS, ids=lambda d: ",".join("%s=%g" % kv for kv in d.items()))
def test_coop_edge_options(gpu_lib, kw):
    for name in ("hopper", "cartpole_friction", "acrobot_impact"):
        _edge_check(gpu_lib, "cuda:0", name, kw)
