"""8-lanes-per-problem cooperative solve pass for cones up to dimension 3 (csrc/od_coop3.h; planar push: one contact, four 3-d
friction cones, one 2-d cone in six lanes).  CPU tier: the 8-lane group is emulated lane by lane (Row8Emu, tests/host_emu) --
the same block algebra, routing tables and reductions as on the device.  GPU tier: the bank-masked 64-bit DPP instructions."""
import numpy as np
import pytest
import torch

import parity_checks as P
import workloads as W

NAME = "planar_push"


def _pair(lib, device, B, seed=31, name=NAME, **opts):
    X, U = W.knots(name, B, seed=seed)
    Xd, Ud = torch.tensor(X, device=device), torch.tensor(U, device=device)
    out = []
    for mode in (1, 3):              # lane-per-problem | the 8-lane cooperative form
        im = P.make_im(name, lib, device)
        if opts:
            im.set_options(**opts)
        im.set_cooperative(mode)
        assert lib.cdll.od_uses_cooperative(im._h, B) == (mode == 3)
        out.append([t.cpu().numpy() for t in im.step_grad(Xd, Ud)] + [im.step(Xd, Ud)[0].cpu().numpy()])
    return X, U, out[0], out[1]


def _vs_lane_per_problem(lib, device, B, name=NAME):
    X, U, ref, got = _pair(lib, device, B, name=name)
    same = (ref[3] == got[3]) & (ref[4] == got[4]).all(0)
    assert same.mean() >= 0.999, same.mean()
    ok = ((ref[3] & 3) == 3) & ((got[3] & 3) == 3)
    assert ok.mean() > 0.99
    e = np.abs(ref[0] - got[0]).max(0)[ok & same]
    assert np.median(e) < 1e-14 and np.quantile(e, 0.99) < 1e-10 and e.max() < 1e-7, (np.median(e), e.max())
    assert np.array_equal(got[5], got[0])                                   # od_step == od_step_grad state
    g = W.grad_rel_err(np.concatenate([ref[1], ref[2]], 1), np.concatenate([got[1], got[2]], 1))[ok & same]
    assert np.median(g) < 1e-11 and (g < P.GRAD_TOL).mean() > 0.99


def test_planar_push_has_cooperative_kernels(emu_lib):
    im = P.make_im(NAME, emu_lib, "cpu")
    uses = emu_lib.cdll.od_uses_cooperative
    assert uses(im._h, 12850) == 1 and uses(im._h, 1 << 20) == 1          # automatic at every batch size (measured: 2.4x at 65 536 knots)
    im.set_cooperative(1); assert uses(im._h, 64) == 0
    im.set_cooperative(0); im.set_launch_config(16, 4); assert uses(im._h, 64) == 0     # an explicit mapping wins


@pytest.mark.parametrize("name", [NAME, "hopper"])
def test_coop3_matches_lane_per_problem_emulated(emu_lib, name):
    """(the hopper has both cooperative forms: 16 lanes per problem up to 4096 problems, this one up to 8192)"""
    _vs_lane_per_problem(emu_lib, "cpu", 512, name)


@pytest.mark.parametrize("B", [1, 2, 3, 9, 17, 2049, 4099])
def test_coop3_ragged_batches_emulated(emu_lib, B):
    """2, 4 and 8 problems per wavefront, groups past the end of the batch"""
    X, U = W.knots(NAME, 4099, seed=83)
    im = P.make_im(NAME, emu_lib, "cpu")
    im.set_cooperative(2)
    full = im.step(torch.tensor(X), torch.tensor(U))
    part = im.step(torch.tensor(np.ascontiguousarray(X[:, :B])), torch.tensor(np.ascontiguousarray(U[:, :B])))
    for a, b in zip(full, part):
        assert torch.equal(a[..., :B], b)


def test_coop3_against_oracle_emulated(oracle, emu_lib):
    im, X, U, out = P.check_step_grad(oracle, emu_lib, "cpu", NAME, 256)
    assert emu_lib.cdll.od_uses_cooperative(im._h, 256) == 1


def test_coop3_bundle_emulated(oracle, emu_lib):
    P.check_bundle(oracle, emu_lib, "cpu", NAME, 4, 48)


def _rollout(oracle, lib, device, B, T):
    x1, U = P.planar_push_rollout_inputs(B, T)          # (harder pushes leave knots at max_iter, in the oracle too)
    im = P.make_im(NAME, lib, device)
    im.set_cooperative(2)
    x1d, Ud = torch.tensor(x1, device=device), torch.tensor(U, device=device)
    X, A, Bm, st, it, _ = im.rollout(x1d, Ud)
    X, st, it = X.clone(), st.clone(), it.clone()
    for t in (0, T - 1):
        D, DX, DU, s1, i1 = im.step_grad(X[:, t].contiguous(), Ud[:, t].contiguous())
        assert torch.equal(D, X[:, t + 1]) and torch.equal(s1, st[t]) and torch.equal(i1, it[:, t])
    Xo, Ao, Bo, bad = oracle.rollout(P.make_sim(oracle, NAME), x1, U)
    ok = ((st.cpu().numpy() & 3) == 3).all(0)
    assert ok.mean() > 0.9
    err = np.abs(X.cpu().numpy() - Xo)[:, :, ok].max(0) / np.maximum(1e-2, np.abs(Xo)[:, :, ok].max(0))
    assert err.max() < P.STATE_TOL, err.max()
    im.set_cooperative(1)
    X1, _, _, st1, it1, _ = im.rollout(x1d, Ud)
    assert (it1 == it).double().mean().item() > 0.995 and (X1 - X).abs().max().item() < 1e-7


def test_coop3_rollout_emulated(oracle, emu_lib):
    _rollout(oracle, emu_lib, "cpu", 6, 10)


EDGE_OPTIONS = [dict(max_iter=0), dict(max_iter=2), dict(max_ls=1), dict(kappa_grad_tol=1e-6), dict(r_tol=1e-3), dict(r_tol=1e-13), dict(eps_min=0.0),
                dict(undercut=5.0), dict(gamma_reg=0.0)]


def _edge(lib, device, kw, name=NAME):
    X, U, ref, got = _pair(lib, device, 64, seed=7, name=name, **kw)
    same = (ref[3] == got[3]) & (ref[4] == got[4]).all(0)
    from test_coop import min_agreement
    assert same.mean() >= min_agreement(kw, same.size), (kw, same.mean())
    fin = np.isfinite(ref[0]).all(0) & np.isfinite(got[0]).all(0)
    e = np.abs(ref[0] - got[0]).max(0)[same & fin]
    noise_level = kw.get("r_tol", 1) < 1e-10 or kw.get("eps_min", 1) == 0.0      # (test_coop.py::_edge_check)
    assert np.median(e) < 1e-12 and e.max() < (1e-4 if noise_level else 1e-6), (kw, np.median(e), e.max())


@pytest.mark.parametrize("kw", EDGE_OPTIONS, ids=lambda d: ",".join("%s=%g" % kv for kv in d.items()))
def test_coop3_edge_options_emulated(emu_lib, kw):
    _edge(emu_lib, "cpu", kw)
    _edge(emu_lib, "cpu", kw, "hopper")


# ---- GPU tier -------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name", [NAME, "hopper"])
def test_coop3_matches_lane_per_problem(gpu_lib, name):
    _vs_lane_per_problem(gpu_lib, "cuda:0", 8192, name)


@pytest.mark.gpu
@pytest.mark.parametrize("B", [1, 2, 3, 9, 17, 2049, 4099])
def test_coop3_ragged_batches(gpu_lib, B):
    X, U = W.knots(NAME, B, seed=81)
    im = P.make_im(NAME, gpu_lib, "cuda:0")
    Xd, Ud = torch.tensor(X, device="cuda:0"), torch.tensor(U, device="cuda:0")
    im.set_cooperative(2); a = im.step_grad(Xd, Ud)
    im.set_cooperative(1); b = im.step_grad(Xd, Ud)
    same = (a[3] == b[3]) & (a[4] == b[4]).all(0)
    assert same.float().mean().item() >= 0.999 and (a[0] - b[0])[:, same].abs().max().item() < 1e-8


@pytest.mark.gpu
def test_coop3_against_oracle(oracle, gpu_lib):
    im, X, U, out = P.check_step_grad(oracle, gpu_lib, "cuda:0", NAME, 1024)
    assert gpu_lib.cdll.od_uses_cooperative(im._h, 1024) == 1


@pytest.mark.gpu
def test_coop3_rollout(oracle, gpu_lib):
    _rollout(oracle, gpu_lib, "cuda:0", 64, 30)


@pytest.mark.gpu
@pytest.mark.parametrize("kw", EDGE_OPTIONS, ids=lambda d: ",".join("%s=%g" % kv for kv in d.items()))
def test_coop3_edge_options(gpu_lib, kw):
    _edge(gpu_lib, "cuda:0", kw)
    _edge(gpu_lib, "cuda:0", kw, "hopper")


# ---- every other kernel of the 8-lane form, and the lane-per-problem kernels it replaced by default -------------------------
def _bundle_forms(lib, device, name, B, N, modes):
    """gradient bundle through two kernel forms: same fit to the amplified rounding of the samples (1e-8 / eps)"""
    from optimization_dynamics_amd import gradient_bundle as gbm, models
    X, U = W.knots(name, B, seed=33)
    gb = gbm.GradientBundle(models.BY_NAME[name], N=N, eps=1e-4, seed=5)
    out = []
    for mode in modes:
        im = P.make_im(name, lib, device, info=gb)
        im.set_cooperative(mode)
        dz, st = gbm.gradient_batch(im, gb, torch.tensor(X, device=device), torch.tensor(U, device=device))
        out.append((dz.cpu().numpy(), st.cpu().numpy()))
    ok = (out[0][1] == 1) & (out[1][1] == 1)
    assert ok.mean() > 0.9
    d = np.abs(out[0][0] - out[1][0])[:, :, ok].max() / max(1.0, np.abs(out[0][0][:, :, ok]).max())
    assert d < 1e-3, d


def test_bundle_kernel_forms_emulated(emu_lib):
    _bundle_forms(emu_lib, "cpu", NAME, 3, 64, (1, 3))
    _bundle_forms(emu_lib, "cpu", "hopper", 3, 64, (1, 3))


def test_policy_rollout_eight_lane_form_emulated(emu_lib):
    P.check_coop_policy_rollout(emu_lib, "cpu", B=4, T=8, name=NAME, mode=3)
    P.check_coop_policy_rollout(emu_lib, "cpu", B=4, T=8, name="hopper", mode=3)


@pytest.mark.gpu
def test_bundle_kernel_forms(gpu_lib):
    """k_bundle_coop3<planar push> against k_bundle_ldsf (the LDS factor store, lane per problem); k_bundle_coop3<hopper> (what
    4097..8192 hopper samples run: 50 knots x 101) against k_bundle"""
    _bundle_forms(gpu_lib, "cuda:0", NAME, 50, 256, (1, 0))
    _bundle_forms(gpu_lib, "cuda:0", "hopper", 50, 100, (1, 0))


@pytest.mark.gpu
def test_policy_rollout_eight_lane_form(gpu_lib):
    P.check_coop_policy_rollout(gpu_lib, "cuda:0", B=16, T=20, name=NAME, mode=3)
    P.check_coop_policy_rollout(gpu_lib, "cuda:0", B=16, T=20, name="hopper", mode=3)


def test_parallel_line_search_variant_emulated(emu_lib):
    """the lane-parallel line search of the 8-lane form (a measured, not shipped variant: DESIGN.md section 3.6) picks the trial
    the sequential loop picks: same bits, including on jammed knots that take ten and more trials per iteration"""
    import os, subprocess
    from optimization_dynamics_amd import _lib
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host_emu")
    subprocess.check_call(["make", "-C", d, "-j", "8", "libod_emu_c3pls.so"], stdout=subprocess.DEVNULL)
    var = _lib.Library(os.path.join(d, "libod_emu_c3pls.so"))
    for name, B, opts in ((NAME, 768, {}), ("hopper", 768, {}), (NAME, 96, dict(max_ls=3)), (NAME, 96, dict(max_ls=11)), ("hopper", 96, dict(max_ls=1))):
        X, U = W.knots(name, B, seed=97)
        out = []
        for lib in (emu_lib, var):
            im = P.make_im(name, lib, "cpu")
            if opts:
                im.set_options(**opts)
            im.set_cooperative(3)
            out.append(im.step_grad(torch.tensor(X), torch.tensor(U)))
        for a, b in zip(*out):
            assert torch.equal(a, b), (name, opts)
        assert out[0][4].max().item() >= 1
    # rollouts in which iterations go past the second trial (a build that picks any other trial than the sequential loop's
    # differs on each of these three)
    hx, hU = W.hopper_rollout_inputs(32, 40, seed=0, u_sigma=1.0)
    px, pU = P.planar_push_rollout_inputs(24, 26, seed=5)
    X, U = W.knots(NAME, 512, seed=3)
    for name, fn in (("hopper", lambda im: im.rollout(torch.tensor(hx), torch.tensor(hU))),
                     (NAME, lambda im: im.rollout(torch.tensor(px), torch.tensor(2.0 * pU))),
                     (NAME, lambda im: im.step_grad(torch.tensor(X), torch.tensor(10.0 * U)))):
        out = []
        for lib in (emu_lib, var):
            im = P.make_im(name, lib, "cpu"); im.set_cooperative(3)
            out.append(fn(im))
        for a, b in zip(*out):
            if torch.is_tensor(a):
                assert torch.equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [3, 2])
def test_policy_rollout_with_converged_trajectories_gpu(gpu_lib, mode):
    """the device-resident solver on the hopper under the 8-lane (mode 3) and the 16-lane (mode 2) cooperative rollout kernels, seven
    problems that converge at different iterations: each equals its solve in a batch of one bit for bit, so a group whose DPP-row
    partner has left computes what it computes beside a live partner"""
    import ilqr_checks as C
    its = C.check_converged_neighbours_do_not_disturb(gpu_lib, "cuda:0", mode)
    print("iterations per problem:", its)
