"""od_ilqr_backward alone: rocket sizes (n = 12, m = 3, T = 60): the matrix-core kernel (mode 0), the DPP-row kernel (mode 2) and the
LDS kernels (mode 1), GPU box; HIP events around 20 calls"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import ilqr_checks as C
import optimization_dynamics_amd as od
from optimization_dynamics_amd import _lib as _L
lib = _L.Library(os.environ['OD_LIB']) if os.environ.get('OD_LIB') else od.default_library()
for B in [int(v) for v in os.environ.get('OD_BS', '64,1024,4096,16384').split(',')]:
    dyn, obj, x1, U0 = C.rocket_problem(lib, "cuda:0", B, 60, dtype=torch.float32, seed=1)
    x1t, Ut = torch.tensor(x1, device="cuda:0"), torch.tensor(U0, device="cuda:0")
    solver = od.ILQR(dyn, obj, 60)
    X, A, Bm, st, _, _ = dyn.rollout(x1t, Ut)
    lam = torch.zeros(12, B, dtype=torch.float64, device="cuda:0")
    quad = obj.expansion(X, Ut.double(), lam, 1.0)
    row = {}
    for mode, nm in ((1, "lds"), (2, "dpp_row"), (0, "mfma")):
        lib.check(lib.cdll.od_set_cooperative(dyn._h, mode))
        solver.backward(A, Bm, quad, 1e-6); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): out = solver.backward(A, Bm, quad, 1e-6)
        e1.record(); torch.cuda.synchronize(); row[nm] = round(e0.elapsed_time(e1) / 20, 4)
        if nm == "dpp_row": ref = [o.clone() for o in out[:3]]
        if nm == "mfma": row["mfma_vs_row_rel"] = ["%.1e" % ((o - r).abs().max() / r.abs().max()).item() for o, r in zip(out[:3], ref)]
    lib.check(lib.cdll.od_set_cooperative(dyn._h, 0))
    print("rocket backward B=%d T=60 (wrapper included): %s ms" % (B, row))
