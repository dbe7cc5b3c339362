"""Soak: cooperative against lane-per-problem kernels over many seeds (development record).
usage (GPU box): python tools/coop_soak.py [nseeds]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import parity_checks as P, workloads as W
from optimization_dynamics_amd import _lib
lib = _lib.default_library()
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 30
out = {}
for name in ("hopper", "cartpole_friction", "acrobot_impact"):
    rows = []
    for seed in range(1000, 1000 + ns):
        X, U = W.knots(name, 8192, seed=seed)
        Xd, Ud = torch.tensor(X, device="cuda:0"), torch.tensor(U, device="cuda:0")
        res = []
        for mode in (1, 2):
            im = P.make_im(name, lib, "cuda:0"); im.set_cooperative(mode)
            res.append([t.cpu().numpy() for t in im.step_grad(Xd, Ud)])
        a, b = res
        same = (a[3] == b[3]) & (a[4] == b[4]).all(0)
        ok = ((a[3] & 3) == 3) & ((b[3] & 3) == 3)
        e = np.abs(a[0] - b[0]).max(0)
        g = W.grad_rel_err(np.concatenate([a[1], a[2]], 1), np.concatenate([b[1], b[2]], 1))
        rows.append(dict(seed=seed, same_iters_frac=float(same.mean()), both_converged=float(ok.mean()), state_diff_median=float(np.median(e[ok])),
                         state_diff_p999=float(np.quantile(e[ok], 0.999)), state_diff_max_same_iters=float(e[ok & same].max()),
                         state_diff_max=float(e[ok].max()), grad_rel_p99=float(np.quantile(g[ok], .99)), grad_rel_max_same_iters=float(g[ok & same].max())))
    out[name] = rows
    print(name, "min same-iters", min(r["same_iters_frac"] for r in rows), "max state diff (same iters)", max(r["state_diff_max_same_iters"] for r in rows),
          "max state diff", max(r["state_diff_max"] for r in rows), "max grad p99", max(r["grad_rel_p99"] for r in rows), flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "coop_soak.json"), "w"), indent=1)
