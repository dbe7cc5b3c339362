import os, sys
import torch
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ilqr_checks as C
import optimization_dynamics_amd as od
dev = torch.device("cuda", 0)
from optimization_dynamics_amd import _lib as _L
lib = _L.Library(os.environ["OD_LIB"]) if os.environ.get("OD_LIB") else od.default_library()
B, T, n_it = 4096, 60, 10
for which in ("ex", "hover"):
  for dtype in (torch.float32, torch.float64):
    if which == "ex": dyn, obj, x1, U0 = C.config5_problem(lib, dev, B, dtype=dtype)
    else: dyn, obj, x1, U0 = C.rocket_problem(lib, dev, B, T, dtype=dtype, seed=1)
    x1t, Ut = torch.tensor(x1, device=dev), torch.tensor(U0, device=dev)
    sol = od.ILQR(dyn, obj, T)
    for ppw in (0, 64, 32, 16, 8):
        lib.check(lib.cdll.od_set_launch_config(dyn.info._h, ppw, 0))
        d = sol.device_solver(B, max_iter=n_it, obj_tol=0.0)
        d.init(x1t, Ut); d.iterate(2); d.init(x1t, Ut)
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); d.iterate(n_it); e1.record(); torch.cuda.synchronize(dev)
        X, U, J = d.get()
        print(which, dtype, "ppw", ppw, "ms/iter %.3f" % (e0.elapsed_time(e1) / n_it), "J %.6f" % J.mean().item(), flush=True)
    lib.check(lib.cdll.od_set_launch_config(dyn.info._h, 0, 0))
