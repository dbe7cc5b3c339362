"""Validates a delivered directory of TRUE reference vectors (written by oracle/gen_golden.jl where Julia + the pinned packages
exist) before the tests consume it:  python tests/golden/check_reference_dir.py [tests/golden/reference]
Checks that every expected file is there with the size its shape implies, that the values are finite where they must be,
that the inputs the vectors were computed from are the committed ones (optional: pass the inputs directory as second
argument), and prints what tests/test_reference_golden.py will be able to compare.  Exit code 0 = ready."""
import os
import sys

import numpy as np

here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(here))
sys.path.insert(0, here)
import export_inputs as E  # noqa: E402
import workloads as W  # noqa: E402

NZ = {"acrobot_impact": 6, "acrobot_nominal": 2, "cartpole_friction": 10, "cartpole_frictionless": 2, "planar_push": 35, "hopper": 20}


def expected():
    """name -> (shape, required, finite)"""
    g = np.load(os.path.join(here, "oracle_v1.npz"))
    out = {}
    for name in W.CONFIGS:
        n, B = g[name + "/X"].shape
        nu = g[name + "/U"].shape[0]
        out[name + "_D"] = ((n, B), True, True)
        out[name + "_DX"] = ((n, n, B), True, True)
        out[name + "_DU"] = ((n, nu, B), True, True)
        out[name + "_IT"] = ((3, B), False, True)
        out[name + "_ST"] = ((3, B), False, True)
        out[name + "_ZG"] = ((NZ[name], B), False, False)
    B = g["rocket/X"].shape[1]
    for k, shp in (("Y", (12, B)), ("DX", (12, 12, B)), ("DU", (12, 3, B)), ("Yp", (12, B)), ("DXp", (12, 12, B)), ("DUp", (12, 3, B)),
                   ("UP", (3, B)), ("DP", (3, 3, B))):
        out["rocket_" + k] = (shp, True, True)
    for name in E.BUNDLE:
        X, U, eta = E.bundle_case(name)
        nq = X.shape[0] // 2
        out["bundle_%s_DZ" % name] = ((nq, X.shape[0] + U.shape[0], X.shape[1]), False, True)
    return out


def main(d, inputs=None):
    problems, notes = [], []
    if not os.path.isdir(d):
        print("no such directory:", d)
        return 2
    for name, (shape, required, finite) in sorted(expected().items()):
        p = os.path.join(d, name + ".bin")
        want = 8 * int(np.prod(shape))
        if not os.path.exists(p):
            (problems if required else notes).append("%s %s.bin (%s, %d bytes)" % ("MISSING" if required else "optional, absent:", name, "x".join(map(str, shape)), want))
            continue
        have = os.path.getsize(p)
        if have != want:
            problems.append("%s.bin has %d bytes, its shape %s needs %d" % (name, have, "x".join(map(str, shape)), want))
            continue
        a = np.fromfile(p, dtype="<f8")
        if finite and not np.isfinite(a).all():
            problems.append("%s.bin holds %d non-finite values" % (name, int((~np.isfinite(a)).sum())))
        if name.endswith("_IT") and (a < 0).all():
            notes.append("%s.bin: the installed RoboDojo does not expose iteration counts (all -1): counts will not be compared" % name)
        if name.endswith("_ZG") and not np.isfinite(a).all():
            notes.append("%s.bin: no gradient iterate (NaN): ill-conditioned gradients are then accepted by condition number only" % name)
    if inputs:
        import tempfile
        with tempfile.TemporaryDirectory() as t:
            E.main(t)
            for f in sorted(os.listdir(t)):
                q = os.path.join(inputs, f)
                if not os.path.exists(q) or open(q, "rb").read() != open(os.path.join(t, f), "rb").read():
                    problems.append("input %s differs from what tests/golden/export_inputs.py writes today: the vectors belong to other inputs" % f)
    for n_ in notes:
        print("note:", n_)
    for p_ in problems:
        print("PROBLEM:", p_)
    print("%s: %d problem(s); %s" % (d, len(problems), "ready -- run python -m pytest tests/test_reference_golden.py (and -m gpu on an MI355X)" if not problems else "not usable yet"))
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "reference"), sys.argv[2] if len(sys.argv) > 2 else None))
