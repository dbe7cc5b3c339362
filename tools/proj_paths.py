"""thrust-cone projection: how often the device (fp64 / fp32) and the oracle follow the exact-arithmetic line-search path
(oracle/arbiter.c::od_arbiter_soc_projection), and how far every end point is from the closed-form Euclidean projection"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from oracle import oracle as O
import optimization_dynamics_amd as od
from optimization_dynamics_amd import models, rocket as rk
O.build()
lib = od.default_library(); dev = "cuda:0"
out = {}
for seed in (7, 8, 9):
    rng = np.random.default_rng(seed)
    B = 2000
    U = np.stack([rng.normal(0, 4, B), rng.normal(0, 4, B), rng.uniform(-4, 18, B)])
    E = np.zeros((3, B)); Oz = np.zeros((3, B)); P = np.zeros((3, B)); okE = np.zeros(B, bool)
    for b in range(B):
        ok, ze, ite, tr, lerr = O.arbiter_soc_projection(12.5, U[:, b], True)
        E[:, b] = ze[:3]; okE[b] = ok
        Oz[:, b] = O.soc_projection(12.5, U[:, b], False)[1][:3]
        P[:, b] = O.project_thrust_cone(U[:, b], 12.5)
    sc = np.maximum(1.0, np.abs(P).max(0))
    row = {"oracle_off_path": float((np.abs(Oz - E).max(0) / sc >= 1e-7).mean()), "oracle_max_dev": float((np.abs(Oz - E).max(0) / sc).max()),
           "exact_vs_closed_form_max": float((np.abs(E - P).max(0) / sc).max()), "exact_converged": float(okE.mean())}
    for dt, tol in ((torch.float64, 1e-7), (torch.float32, 2e-3)):
        info = rk.RocketInfo(models.rocket, 12.5, 0.05, dtype=dt, device=dev, lib=lib)
        UP, DP, st = info.project(torch.tensor(U, dtype=dt), grads=True)
        UP = UP.double().cpu().numpy(); st = st.cpu().numpy()
        conv = (st & 0x30) == 0x30
        e = np.abs(UP - E).max(0) / sc
        nm = "f64" if dt == torch.float64 else "f32"
        row[nm + "_off_path"] = float((e[conv] >= tol).mean()); row[nm + "_max_dev"] = float(e[conv].max())
        row[nm + "_vs_closed_form_max"] = float((np.abs(UP - P).max(0) / sc)[conv].max()); row[nm + "_nonconverged"] = int((~conv).sum())
    out["seed%d" % seed] = row
    print(seed, row, flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "proj_paths.json"), "w"), indent=1)
