"""Writes the seeded inputs the Julia reference run needs (oracle/gen_golden.jl) as little-endian Float64 binaries:
the knots of oracle_v1.npz (mechanical models + rocket) and the gradient-bundle cases (knots + the perturbations eta,
so that the reference's unseeded RNG, src/gradient_bundle.jl:49-54, is out of the picture).

    python tests/golden/export_inputs.py [outdir = tests/golden/inputs]"""
import os
import sys

import numpy as np

here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(here))
import workloads as W  # noqa: E402

BUNDLE = {"cartpole_friction": 64, "hopper": 50, "planar_push": 64}     # model -> N samples


def bundle_case(name):
    """knots and eta of the gradient-bundle reference case (shared with tests/test_reference_golden.py)"""
    X, U = W.knots(name, 6, seed=31)
    nzb = X.shape[0] + U.shape[0]
    N = BUNDLE[name]
    rng = np.random.default_rng(5)
    eta = np.zeros((nzb, N))
    for i in range(N):                 # src/gradient_bundle.jl:49-54: one coordinate, eps * randn (every coordinate sampled)
        eta[i if i < nzb else rng.integers(nzb), i] = 1e-4 * rng.normal()
    return X, U, eta


def main(out):
    os.makedirs(out, exist_ok=True)

    def dump(name, a):
        np.asfortranarray(a).astype("<f8").ravel(order="F").tofile(os.path.join(out, name + ".bin"))

    g = np.load(os.path.join(here, "oracle_v1.npz"))
    for k in g.files:
        name, arr = k.split("/")
        if arr in ("X", "U"):
            dump("%s_%s" % (name, arr), g[k])
    for name in BUNDLE:
        X, U, eta = bundle_case(name)
        dump("bundle_%s_X" % name, X); dump("bundle_%s_U" % name, U); dump("bundle_%s_eta" % name, eta)
    print("inputs written to", out)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "inputs"))
