"""Generates tests/golden/oracle_v1.npz: seeded inputs and the CPU oracle's outputs for every model.

IMPORTANT: these vectors were produced by THIS repository's oracle (oracle/ip_oracle.c), not by the
Julia reference (which cannot run here: no julia, RoboDojo.jl un-vendored; "parity unpinned").
They pin the oracle against regressions and across machines; oracle/gen_golden.jl is the script
that someone with Julia + the pinned packages can run to obtain true reference vectors in the
same layout.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import oracle as O  # noqa: E402
import workloads as W  # noqa: E402


def main():
    out = {}
    for name, (h, ke, kg, fric) in W.CONFIGS.items():
        X, U = W.knots(name, 16, seed=7)
        kw = dict(kappa_tol=ke, kappa_grad_tol=kg)
        if fric:
            kw["friction"] = fric
        sim = O.make_sim(name, h, **kw)
        D, DX, DU, bad = O.step_grad_batch(sim, X, U)
        out[name + "/X"], out[name + "/U"] = X, U
        out[name + "/D"], out[name + "/DX"], out[name + "/DU"] = D, DX, DU
        out[name + "/bad"] = np.array(bad)
    Xr, Ur = W.rocket_inputs(16, seed=7)
    Y = np.zeros((12, 16)); DXr = np.zeros((12, 12, 16)); DUr = np.zeros((12, 3, 16))
    Yp = np.zeros((12, 16)); DXp = np.zeros((12, 12, 16)); DUp = np.zeros((12, 3, 16)); UP = np.zeros((3, 16))
    for b in range(16):
        st, y, dz, it = O.rocket(0.05, Xr[:, b], Ur[:, b], True)
        Y[:, b], DXr[:, :, b], DUr[:, :, b] = y, dz[:, :12], dz[:, 12:15]
        ok, y, dx, du = O.rocket_proj(0.05, 12.5, Xr[:, b], Ur[:, b])
        Yp[:, b], DXp[:, :, b], DUp[:, :, b] = y, dx, du
        UP[:, b] = O.soc_projection(12.5, Ur[:, b], False)[1][:3]
    out.update({"rocket/X": Xr, "rocket/U": Ur, "rocket/Y": Y, "rocket/DX": DXr, "rocket/DU": DUr,
                "rocket/Yp": Yp, "rocket/DXp": DXp, "rocket/DUp": DUp, "rocket/UP": UP})
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_v1.npz"), **out)
    print("wrote oracle_v1.npz with", len(out), "arrays")


if __name__ == "__main__":
    main()
