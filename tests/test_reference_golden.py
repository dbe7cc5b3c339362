"""Comparison against TRUE reference vectors (tests/golden/reference/*.bin produced by
oracle/gen_golden.jl where Julia + the pinned packages exist).  Skipped while they are absent --
until then parity is unpinned."""
import os

import numpy as np
import pytest

import workloads as W

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(HERE, "golden", "reference")


@pytest.mark.skipif(not os.path.isdir(REF), reason="no reference vectors (run oracle/gen_golden.jl with Julia)")
@pytest.mark.parametrize("name", list(W.CONFIGS))
def test_oracle_matches_julia_reference(oracle, name):
    g = np.load(os.path.join(HERE, "golden", "oracle_v1.npz"))
    X, U = g[name + "/X"], g[name + "/U"]
    n, B = X.shape
    nu = U.shape[0]
    D = np.fromfile(os.path.join(REF, name + "_D.bin"), dtype="<f8").reshape((n, B), order="F")
    DX = np.fromfile(os.path.join(REF, name + "_DX.bin"), dtype="<f8").reshape((n, n, B), order="F")
    DU = np.fromfile(os.path.join(REF, name + "_DU.bin"), dtype="<f8").reshape((n, nu, B), order="F")
    h, ke, kg, fric = W.CONFIGS[name]
    kw = dict(kappa_tol=ke, kappa_grad_tol=kg)
    if fric:
        kw["friction"] = fric
    Do, DXo, DUo, bad = oracle.step_grad_batch(oracle.make_sim(name, h, **kw), X, U)
    assert np.abs(Do - D).max() <= 1e-6 * max(1.0, np.abs(D).max())          # north_star: 1e-6 on states
    assert W.grad_rel_err(DXo, DX).max() <= 1e-4 and W.grad_rel_err(DUo, DU).max() <= 1e-4
