"""Larger parity sweep on the GPU (-m gpu): every mechanical model, several seeds, thousands of knots against the CPU
oracle (OpenMP over the batch) and, for the implicit gradients, against the binary128 arbiter (oracle/arbiter.c) at
the device's own and at the oracle's own gradient iterates.  The 1e-6 / 1e-4 bars are asserted on 100 % of the
converged knots whose solution the oracle itself reproduces under 1e-13 input perturbations (all but ~2 per million); both gradient error columns go to gpurun_out/parity_sweep.json (copied to profiles/ for the round)."""
import json
import os

import numpy as np
import pytest
import torch

import parity_checks as P
import workloads as W

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
MECH = ["acrobot_impact", "acrobot_nominal", "cartpole_friction", "cartpole_frictionless", "hopper", "planar_push"]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# OD_SWEEP_SEEDS=1,2,3,... widens the sweep (a soak run for profiles/; the default three seeds are the test)
SEEDS = tuple(int(t) for t in os.environ.get("OD_SWEEP_SEEDS", "101,202,303").split(","))


def test_parity_sweep(oracle, gpu_lib):
    out = {}
    for name in MECH:
        B = 2048 if name == "planar_push" else 8192
        rows = []
        for seed in SEEDS:
            X, U = W.knots(name, B, seed=seed)
            im = P.make_im(name, gpu_lib, DEV)
            D, DX, DU, st, it = [t.cpu().numpy() for t in im.step_grad(torch.tensor(X), torch.tensor(U))]
            Do, DXo, DUo, bad = oracle.step_grad_batch(P.make_sim(oracle, name), X, U)
            ok = (st & 3) == 3
            # a knot whose cone variables sit exactly on the boundary has a singular Jacobian: the oracle's dense LU
            # returns NaN there (the reference would throw), the device flags it (FACTOR_OK) and drops the unknown
            G_dev, G_ora = np.concatenate([DX, DU], 1), np.concatenate([DXo, DUo], 1)
            nan_ora = ~np.isfinite(G_ora).reshape(-1, B).all(0)
            nan_dev = ~np.isfinite(G_dev).reshape(-1, B).all(0)
            assert not (nan_dev & ok & ~nan_ora).any(), ("non-finite device gradient on a converged knot the oracle differentiates", name, seed, np.nonzero(nan_dev & ok & ~nan_ora)[0][:4].tolist())
            assert (nan_ora & ok).sum() <= 2, ("singular Jacobians in the oracle", name, seed, int((nan_ora & ok).sum()))
            ok = ok & ~nan_ora
            srel_all = np.abs(D - Do).max(0) / np.maximum(1e-2, np.abs(Do).max(0))
            # A state mismatch is a failure unless the ORACLE ITSELF does not reproduce its answer there: the knot is
            # solved again by the oracle from inputs perturbed by 1e-13 relative (16 draws).  Newton / interior-point
            # paths of 20-70 iterations through several contact modes end on different roots under such perturbations
            # (seen on 2 of 1.2 million knots, profiles/r2_parity_soak.json); no implementation can be compared there.
            path_dependent = np.zeros(B, bool)
            for i in np.nonzero(ok & (srel_all >= P.STATE_TOL))[0]:
                rng = np.random.default_rng(int(i))
                # (1e-13 and 1e-11: the two implementations eliminate in different orders, their iterates differ by up to ~1e-12 on
                # ill-conditioned knots -- an oracle that is steady under 1e-13 and scatters over a dozen roots under 1e-11 was seen
                # on one knot of the seed soak, 58 iterations, host build and GPU agreeing with each other to the last bit)
                for eps_p in (1e-13, 1e-11):
                    Xp = X[:, [i]] * (1 + eps_p * rng.normal(size=(X.shape[0], 16)))
                    Up = U[:, [i]] * (1 + eps_p * rng.normal(size=(U.shape[0], 16)))
                    Dp = oracle.step_grad_batch(P.make_sim(oracle, name), Xp, Up)[0]
                    path_dependent[i] |= (np.ptp(Dp, axis=1).max() / max(1e-2, np.abs(Do[:, i]).max())) > 10 * P.STATE_TOL
            ok = ok & ~path_dependent
            srel = srel_all[ok]
            grel = W.grad_rel_err(np.concatenate([DX, DU], 1), np.concatenate([DXo, DUo], 1))[ok]
            nq = X.shape[0] // 2
            e = P.exact_gradient_errors(oracle, im, name, X, U, np.concatenate([DX[nq:], DU[nq:]], 1))
            fin = ok & np.isfinite(e["dev"]) & np.isfinite(e["explained"])
            excess = e["cross"][fin] - 2.0 * e["explained"][fin]
            rows.append(dict(seed=seed, knots=B, converged=int(ok.sum()), path_dependent_knots=int(path_dependent.sum()), oracle_nonconverged_solves=int(bad), oracle_singular=int(nan_ora.sum()),
                             state_rel_max=float(srel.max()), state_rel_median=float(np.median(srel)),
                             grad_rel_median=float(np.median(grel)), grad_rel_p99=float(np.percentile(grel, 99)),
                             grad_rel_p999=float(np.percentile(grel, 99.9)), grad_rel_max=float(grel.max()),
                             frac_grad_within_1e4=float((grel < P.GRAD_TOL).mean()), mean_iterations=float(it[0].mean()),
                             # binary128 arbiter, relative to max |exact gradient| of the knot, over ALL converged knots
                             arbitrated_knots=int(fin.sum()),
                             err_device_vs_exact_at_device_iterate_max=float(e["dev"][fin].max()),
                             err_oracle_vs_exact_at_oracle_iterate_max=float(e["orc"][fin].max()),
                             device_vs_oracle_max=float(e["cross"][fin].max()),
                             exact_at_device_vs_exact_at_oracle_iterate_max=float(e["explained"][fin].max()),
                             device_vs_oracle_beyond_iterates_max=float(excess.max()),
                             iterate_diff_max=float(e["iterate_diff"][fin].max()),
                             cond_median=float(np.nanmedian(e["cond"][fin])), cond_max=float(np.nanmax(e["cond"][fin]))))
        out[name] = rows
    d = os.path.join(ROOT, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    json.dump(out, open(os.path.join(d, "parity_sweep.json"), "w"), indent=1)
    for name, rows in out.items():
        for r in rows:
            assert r["converged"] > 0.99 * r["knots"], (name, r)
            assert r["path_dependent_knots"] <= 2, (name, r)
            assert r["state_rel_max"] < P.STATE_TOL, (name, r)                       # 1e-6 relative on states
            # implicit gradients, 100 % of the converged knots: the device reproduces the exact (binary128) gradient at its
            # own iterate, and differs from the oracle by no more than 1e-4 beyond what the two iterates explain
            assert r["arbitrated_knots"] >= r["converged"] - 2, (name, r)
            assert r["err_device_vs_exact_at_device_iterate_max"] < P.EXACT_TOL, (name, r)
            assert r["err_device_vs_exact_at_device_iterate_max"] <= max(P.GRAD_TOL, r["err_oracle_vs_exact_at_oracle_iterate_max"]), (name, r)
            assert r["device_vs_oracle_beyond_iterates_max"] < P.GRAD_TOL, (name, r)
            assert r["grad_rel_median"] < 1e-9, (name, r)


def test_rocket_parity_sweep(oracle, gpu_lib):
    """the rocket path (src/models/rocket/dynamics.jl:101-268; BASELINE config 5 is the one config asked in single precision) at the
    scale of the mechanical models' sweep: 3 seeds x 8192 knots, double AND single precision, every converged knot -- the dynamics step
    at 1e-6 / 1e-4 against the oracle, the thrust-cone projection's gradient arbitrated per knot in binary128 at the device's own
    iterate, every projected control verified against the algorithm's own stopping rule, the chain product against the oracle's
    dynamics gradient times the arbitrated projection gradient (parity_checks.check_rocket_sweep); the error columns go to
    gpurun_out/rocket_parity_sweep.json (copied to profiles/ for the round)"""
    rows = []
    for dtype in (torch.float64, torch.float32):
        for seed in SEEDS:
            rows.append(P.check_rocket_sweep(oracle, gpu_lib, DEV, 8192, seed, dtype))
    d = os.path.join(ROOT, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    json.dump(rows, open(os.path.join(d, "rocket_parity_sweep.json"), "w"), indent=1)
    for r in rows:
        assert r["dyn_converged"] > 0.999 * r["knots"] and r["proj_arbitrated"] > 0.99 * r["knots"] and r["chain_converged"] > 0.99 * r["knots"], r


def test_rollout_parity_sweep(oracle, gpu_lib):
    """headline-shaped rollouts (hopper, T = 100) against the oracle's rollouts: the recursion amplifies rounding
    differences through contact-mode switches, so the comparison is knot by knot in time"""
    B, T = 512, 100
    x1, U = W.hopper_rollout_inputs(B, T, seed=5, u_sigma=1.0)
    im = P.make_im("hopper", gpu_lib, DEV)
    X, A, Bm, st, it, _ = im.rollout(torch.tensor(x1), torch.tensor(U))
    Xn, stn = X.cpu().numpy(), st.cpu().numpy()
    Xo, Ao, Bo, bad = oracle.rollout(P.make_sim(oracle, "hopper"), x1, U)
    ok = ((stn & 3) == 3).all(0)
    err = np.abs(Xn - Xo)[:, :, ok].max(0) / np.maximum(1e-2, np.abs(Xo)[:, :, ok].max(0))      # (T+1, n_ok)
    rows = {"trajectories": B, "all_knots_converged": int(ok.sum())}
    for t in (1, 10, 25, 50, 100):
        rows["t=%d" % t] = dict(median=float(np.median(err[t])), p99=float(np.percentile(err[t], 99)), max=float(err[t].max()),
                                frac_within_1e6=float((err[t] < P.STATE_TOL).mean()))
    d = os.path.join(ROOT, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    json.dump(rows, open(os.path.join(d, "rollout_parity_sweep.json"), "w"), indent=1)
    assert ok.mean() > 0.9
    assert err[:11].max() < P.STATE_TOL
    assert np.median(err[-1]) < 1e-6 and (err[-1] < P.STATE_TOL).mean() > 0.9


def test_device_solutions_zero_the_hand_written_residuals(oracle, gpu_lib):
    """The second source of the models: device and oracle share one symbolic specification, oracle/models_np.py restates
    src/models/<model>/model.jl by hand.  At the solutions the MI355X returns (2048 knots per model, SEEDS[0]) the hand-written
    residual equals the generated one to 1e-12 and the loop's stopping test holds -- equality rows < r_tol, complementarity rows <
    kappa_eval_tol; gpurun_out/hand_written_residuals.json"""
    out = [P.check_solutions_against_hand_written_residuals(oracle, gpu_lib, DEV, name, B=2048, seed=SEEDS[0]) for name in MECH]
    out.append(P.check_rocket_solutions_against_hand_written_residuals(oracle, gpu_lib, DEV, B=2048, seed=SEEDS[0]))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "hand_written_residuals.json"), "w") as f:
        json.dump(out, f, indent=1)
