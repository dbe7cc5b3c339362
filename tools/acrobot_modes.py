import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import workloads as W, parity_checks as P
import optimization_dynamics_amd as od
lib = od.default_library(); dev = "cuda:0"
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(n):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
for name in ("acrobot_impact", "cartpole_friction"):
    for B in (1024, 4096, 65536):
        X, U = W.knots(name, B, seed=1)
        Xd, Ud = torch.tensor(X, device=dev), torch.tensor(U, device=dev)
        im = P.make_im(name, lib, dev)
        row = {}
        for mode in (0, 1, 2):
            im.set_cooperative(mode)
            row[mode] = round(timeit(lambda: im.step_grad(Xd, Ud)), 4)
        D, DX, DU, st, it = im.step_grad(Xd, Ud)
        print(name, B, "ms by mode (0 auto, 1 lane, 2 coop16):", row, "max it", int(it.max()), flush=True)
