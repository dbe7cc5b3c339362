"""Known-answer / self-consistency tests of the CPU oracle (oracle/ip_oracle.c).

The reference ships no tests or golden vectors ("parity unpinned"), so the oracle is
pinned by what can be derived without Julia: roots and finite differences for the smooth models, KKT
conditions for the cone models, the closed-form thrust-cone projection, the commented LS test of
src/ls.jl:62-144, and physical sanity checks."""
import os

import numpy as np
import pytest
from scipy.optimize import fsolve

from oracle import models_np as NP
import workloads as W


def _theta(name, x, u, h, fric=()):
    nq = len(x) // 2
    q1, q2 = x[:nq], x[nq:]
    v1 = (q2 - q1) / h
    return np.concatenate([q2 - h * v1, q2, u, fric, [h]])


@pytest.mark.parametrize("name", ["acrobot_nominal", "cartpole_frictionless"])
def test_smooth_models_root_and_gradient(oracle, name):
    h = 0.05
    sim = oracle.make_sim(name, h)
    X, U = W.knots(name, 20, seed=3)
    for b in range(20):
        x, u = X[:, b], U[:, b]
        st, d, it = oracle.f(sim, x, u)
        assert st == 1
        th = _theta(name, x, u, h)
        root = fsolve(lambda z: NP.RESIDUALS[name](z, th, 0.0), x[2:], xtol=1e-13)
        assert np.abs(d[2:] - root).max() < 1e-7
        assert np.allclose(d[:2], x[2:])
        # implicit gradient vs central finite differences of the solver
        _, dx, _ = oracle.fx(sim, x, u)
        _, du, _ = oracle.fu(sim, x, u)
        e = 1e-6
        for j in range(4):
            xp, xm = x.copy(), x.copy(); xp[j] += e; xm[j] -= e
            fd = (oracle.f(sim, xp, u)[1] - oracle.f(sim, xm, u)[1]) / (2 * e)
            assert np.abs(dx[:, j] - fd).max() < 1e-5 * max(1, np.abs(fd).max())
        up, um = u + e, u - e
        fd = (oracle.f(sim, x, up)[1] - oracle.f(sim, x, um)[1]) / (2 * e)
        assert np.abs(du[:, 0] - fd).max() < 1e-5 * max(1, np.abs(fd).max())


@pytest.mark.parametrize("name", ["acrobot_impact", "cartpole_friction", "hopper", "planar_push"])
def test_cone_models_kkt_conditions(oracle, name):
    h, ke, kg, fric = W.CONFIGS[name]
    # kappa_reg = 0: no regularisation during the iterations, so differentiate_solution! clamps the
    # orthant variables at exactly kappa_tol * gamma_reg and the check below can reproduce it
    kw = dict(kappa_tol=ke, kappa_grad_tol=kg, kappa_reg=0.0)
    if fric:
        kw["friction"] = fric
    sim = oracle.make_sim(name, h, **kw)
    X, U = W.knots(name, 40, seed=5)
    import re
    from optimization_dynamics_amd.codegen import models as CM  # index sets only
    nconv = 0
    its = []
    for b in range(40):
        st, z, dz, it = oracle.step_full(sim, X[:, b], U[:, b], ke, True)
        its.append(it)
        if not st:
            continue
        nconv += 1
        th = _theta(name, X[:, b], U[:, b], h, fric if (fric and name != "planar_push") else ())
        r = NP.RESIDUALS[name](z, th, 0.0)
        d = oracle.dims(name)
        spec = SPECS[name]
        assert np.abs(r[spec["equr"]]).max() < 1e-8          # r_tol (src/dynamics.jl:29)
        assert np.abs(r[spec["bil"]]).max() < ke             # kappa_tol
        for p, q in zip(*spec["ort"]):
            assert z[p] > 0 and z[q] > 0
        for (pi, di) in spec["soc"]:
            assert z[pi[0]] >= np.linalg.norm(z[pi[1:]]) - 1e-12
            assert z[di[0]] >= np.linalg.norm(z[di[1:]]) - 1e-12
        # implicit-function solve, recomputed in numpy at the same (clamped) point
        reg = max(ke * 0.1, 0.0)
        zr = z.copy()
        for p, q in zip(*spec["ort"]):
            zr[p], zr[q] = max(zr[p], reg), max(zr[q], reg)
        rz, rth = oracle.eval_rz(name, zr, th), oracle.eval_rth(name, z, th)
        ref = -np.linalg.solve(rz, rth)
        nq = d["nq"]
        scale = max(1.0, np.abs(ref[:nq]).max())
        # the oracle's reg is max(reg_val, kappa*gamma_reg) with reg_val <= kappa*gamma_reg here
        assert np.abs(dz[:nq] - ref[:nq]).max() < 1e-6 * scale
    assert nconv >= 38
    assert max(its) <= 100 and np.mean(its) < 15


SPECS = {
    "acrobot_impact": dict(equr=[0, 1, 2, 3], bil=[4, 5], ort=([2, 3], [4, 5]), soc=[]),
    "cartpole_friction": dict(equr=list(range(6)), bil=[6, 7, 8, 9], ort=([], []),
                              soc=[([2, 4], [6, 8]), ([3, 5], [7, 9])]),
    "hopper": dict(equr=list(range(12)), bil=list(range(12, 20)), ort=([4, 5, 6, 7], [8, 9, 10, 11]),
                   soc=[([12, 14], [16, 18]), ([13, 15], [17, 19])]),
    "planar_push": dict(equr=list(range(20)), bil=list(range(20, 35)), ort=([5], [6]),
                        soc=[([7 + i, 12 + 2 * i, 13 + 2 * i], [21 + i, 26 + 2 * i, 27 + 2 * i]) for i in range(4)]
                        + [([11, 20], [25, 34])]),
}


def _project_thrust_cone(u, umax):
    """closed-form Euclidean projection onto {|u_1:2| <= u_3, 0 <= u_3 <= umax}"""
    a, t = np.linalg.norm(u[:2]), u[2]
    # project onto the cone first, then handle the cap by 1-D search over t in [0, umax] (convex)
    best, bestd = None, np.inf
    for tt in np.linspace(0, umax, 20001):
        r = min(a, tt)
        p = np.r_[u[:2] * (r / a if a > 0 else 0), tt]
        dd = np.sum((p - u) ** 2)
        if dd < bestd:
            best, bestd = p, dd
    return best


def test_soc_projection_matches_euclidean_projection(oracle):
    # src/models/rocket/dynamics.jl:168-186; kappa_tol = 1e-4 (:79) -> agreement O(kappa_tol)
    rng = np.random.default_rng(2)
    its = []
    for k in range(40):
        u = rng.normal(size=3) * rng.choice([1, 5, 20])
        st, z, dz, it = oracle.soc_projection(12.5, u, True)
        its.append(it)
        assert st == 1
        p = _project_thrust_cone(u, 12.5)
        assert np.abs(z[:3] - p).max() < 5e-3
        assert np.linalg.norm(z[:2]) <= z[2] + 1e-9 and -1e-9 <= z[2] <= 12.5 + 1e-9   # examples/rocket.jl:151
    assert max(its) <= 25


def test_ls_known_answer_from_reference_comment(oracle):
    # src/ls.jl:62-144: f(z) = A x + B u, A = [1 1; 0 1], B = [0; 1], eta = +-0.1 e_i, N = 2 nz
    A = np.array([[1.0, 1.0], [0.0, 1.0]]); Bv = np.array([0.0, 1.0])
    f = lambda z: A @ z[:2] + Bv * z[2]
    nz, eps = 3, 0.1
    eta = np.zeros((nz, 2 * nz))
    for i in range(nz):
        eta[i, i], eta[i, i + nz] = eps, -eps
    z0 = np.random.default_rng(0).random(nz)
    fz = f(z0)
    feta = np.stack([f(z0 + eta[:, i]) for i in range(2 * nz)], axis=1)
    M, iters = oracle.ls_update(fz, feta, eta)
    assert np.allclose(M, np.array([[1.0, 1.0, 0.0], [0.0, 1.0, 1.0]]), atol=1e-10)
    assert iters <= 2     # exactly quadratic cost: one Newton step


def test_hopper_at_rest_stays_at_rest(oracle):
    sim = oracle.make_sim("hopper", 0.05, kappa_tol=1e-4, kappa_grad_tol=1e-3)
    q = np.array([0.0, 0.55, 0.0, 0.5])
    x = np.r_[q, q]
    u = np.array([0.0, 9.81 * 3.0 * 0.5 * 0.05])        # examples/hopper.jl:270
    for _ in range(50):
        st, x, it = oracle.f(sim, x, u)
        assert st == 1
    assert np.abs(x[4:] - q).max() < 1e-6


def test_acrobot_nominal_energy_drift_is_small(oracle):
    # variational (midpoint) integrator without contact: energy error stays bounded; the model has
    # viscous damping -h/2 v (model.jl:103) so the energy must not grow
    sim = oracle.make_sim("acrobot_nominal", 0.01)
    m1 = m2 = 1.0; l1 = 1.0; lc1 = lc2 = 0.5; J1 = J2 = 0.333; g = 9.81

    def energy(q, v):
        M = np.array([[J1 + J2 + m2 * l1 * l1 + 2 * m2 * l1 * lc2 * np.cos(q[1]), J2 + m2 * l1 * lc2 * np.cos(q[1])],
                      [J2 + m2 * l1 * lc2 * np.cos(q[1]), J2]])
        V = -m1 * g * lc1 * np.cos(q[0]) - m2 * g * (l1 * np.cos(q[0]) + lc2 * np.cos(q[0] + q[1]))
        return 0.5 * v @ M @ v + V
    q = np.array([0.5, 0.2])
    x = np.r_[q, q]
    E = []
    for _ in range(300):
        st, x, it = oracle.f(sim, x, np.zeros(1))
        assert st == 1
        E.append(energy(0.5 * (x[:2] + x[2:]), (x[2:] - x[:2]) / 0.01))
    assert E[-1] <= E[0] + 1e-3 and abs(E[-1] - E[0]) < 2.0


def test_golden_fixture_regression(oracle):
    """tests/golden/oracle_v1.npz was produced by this oracle (make_golden.py), NOT by the Julia
    reference: it guards the oracle against regressions and across machines/compilers."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_v1.npz"))
    for name, (h, ke, kg, fric) in W.CONFIGS.items():
        kw = dict(kappa_tol=ke, kappa_grad_tol=kg)
        if fric:
            kw["friction"] = fric
        sim = oracle.make_sim(name, h, **kw)
        D, DX, DU, bad = oracle.step_grad_batch(sim, g[name + "/X"], g[name + "/U"])
        assert bad == int(g[name + "/bad"])
        assert np.abs(D - g[name + "/D"]).max() < 1e-9
        assert W.grad_rel_err(DX, g[name + "/DX"]).max() < 1e-4
        assert np.median(W.grad_rel_err(DX, g[name + "/DX"])) < 1e-9
        assert W.grad_rel_err(DU, g[name + "/DU"]).max() < 1e-4


def test_projection_arbiter_in_binary128(oracle):
    """oracle/arbiter.c::od_arbiter_soc_projection: the projection's interior-point loop in binary128.  The equality rows are
    linear (defect at binary128 rounding), so exact arithmetic accepts every first line-search trial; the exact-acceptance
    path converges, sits at kappa_tol level from the closed-form Euclidean projection, and the double-precision oracle either
    follows it to 1e-7 or ends within a few kappa_tol of it"""
    rng = np.random.default_rng(3)
    off = 0
    for k in range(60):
        u = np.r_[rng.normal(0, 4, 2), rng.uniform(-4, 18)]
        ok, ze, it, trials, lerr = oracle.arbiter_soc_projection(12.5, u, True)
        assert ok == 1 and it <= 25 and lerr < 1e-28 and all(t == 0 for t in trials)
        p = oracle.project_thrust_cone(u, 12.5)
        sc = max(1.0, np.abs(p).max())
        assert np.abs(ze[:3] - p).max() < 6e-3 * sc
        assert np.abs(p - _project_thrust_cone(u, 12.5)).max() < 2e-3 * sc          # closed form == the brute-force search
        zo = oracle.soc_projection(12.5, u, False)[1]
        d = np.abs(zo[:3] - ze[:3]).max() / sc
        off += d >= 1e-7
        assert d < 5e-4
    assert off <= 12                         # measured: 7-8 % of the controls (profiles/r3_projection_paths.json)


def test_ilqr_oracle_against_its_golden_record(oracle):
    """tests/golden/ilqr_v1.json (tests/golden/make_golden_ilqr.py): the numpy AL-iLQR's decision sequences on examples/hopper.jl as
    shipped and on the cartpole task with two multiplier rounds -- this oracle's own output, a regression pin across machines and
    library versions (the accepted step indices are integers: any change of the rules or of the oracle's dynamics shows)"""
    import json
    import os
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    import make_golden_ilqr as G
    want = json.load(open(os.path.join(here, "golden", "ilqr_v1.json")))
    got = G.cases()
    for case in want:
        w, g = want[case], got[case]
        assert g["steps"] == w["steps"] and g["rounds"] == w["rounds"], case
        assert abs(g["violation"] - w["violation"]) <= 1e-9 + 1e-6 * abs(w["violation"])
        assert np.allclose(g["x_T"], w["x_T"], rtol=0, atol=1e-8)
    assert np.allclose(got["hopper_gait_as_shipped"]["theta"], want["hopper_gait_as_shipped"]["theta"], rtol=0, atol=1e-8)
    assert np.allclose(got["cartpole_two_rounds"]["costs"], want["cartpole_two_rounds"]["costs"], rtol=1e-9, atol=0)
