"""od_ilqr_backward alone: rocket sizes (n = 12, m = 3, T = 60), the DPP-row kernel against the LDS kernels (GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import ilqr_checks as C
import optimization_dynamics_amd as od
lib = od.default_library()
for B in (64, 1024, 4096, 16384):
    dyn, obj, x1, U0 = C.rocket_problem(lib, "cuda:0", B, 60, dtype=torch.float32, seed=1)
    x1t, Ut = torch.tensor(x1, device="cuda:0"), torch.tensor(U0, device="cuda:0")
    solver = od.ILQR(dyn, obj, 60)
    X, A, Bm, st, _, _ = dyn.rollout(x1t, Ut)
    lam = torch.zeros(12, B, dtype=torch.float64, device="cuda:0")
    quad = obj.expansion(X, Ut.double(), lam, 1.0)
    row = {}
    for mode, nm in ((1, "lds"), (0, "dpp_row")):
        lib.check(lib.cdll.od_set_cooperative(dyn._h, mode))
        solver.backward(A, Bm, quad, 1e-6); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): solver.backward(A, Bm, quad, 1e-6)
        torch.cuda.synchronize(); row[nm] = round((time.perf_counter() - t0) / 10 * 1e3, 4)
    print("rocket backward B=%d T=60 (wrapper included): %s ms" % (B, row))
