"""one BASELINE config alone, a few calls (target of rocprofv3 / PMC runs): python tools/run_config_only.py bundle|acrobot|pp_step|rocket|rocket32|rocket_noproj|soc [reps] [B]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import workloads as W, parity_checks as P
import optimization_dynamics_amd as od
from optimization_dynamics_amd import _lib as _L
lib = _L.Library(os.environ["OD_LIB"]) if os.environ.get("OD_LIB") else od.default_library()   # (OD_LIB: a variant build)
dev = "cuda:0"
what = sys.argv[1]; reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
if what == "bundle":
    im = P.make_im("planar_push", lib, dev)
    gb = od.GradientBundle(od.planarpush, N=256, eps=1e-4, seed=0)
    X, U = W.knots("planar_push", 50, seed=2)
    fn = lambda: od.gradient_batch(im, gb, Xd, Ud)
elif what == "acrobot":
    im = P.make_im("acrobot_impact", lib, dev)
    X, U = W.knots("acrobot_impact", 1024, seed=1)
    fn = lambda: im.step_grad(Xd, Ud)
elif what == "pp_step":
    im = P.make_im("planar_push", lib, dev)
    X, U = W.knots("planar_push", 65536, seed=1)
    fn = lambda: im.step_grad(Xd, Ud)
elif what.startswith("rocket") or what == "soc":
    dt = torch.float32 if what == "rocket32" else torch.float64
    info = od.RocketInfo(od.rocket, 12.5, 0.05, dtype=dt, device=dev, lib=lib)
    X, U = W.rocket_inputs(int(sys.argv[3]) if len(sys.argv) > 3 else 65536, seed=3)
    if what == "soc":
        fn = lambda: info.project(Ud, grads=True)
    else:
        fn = lambda: info.solve(Xd, Ud, project=what != "rocket_noproj", grads=True)
Xd, Ud = torch.tensor(X, device=dev), torch.tensor(U, device=dev)
if what == "rocket32":
    Xd, Ud = Xd.float(), Ud.float()
for _ in range(reps):
    fn()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(20):
    fn()
torch.cuda.synchronize()
print("%s: %.4f ms per call" % (what, (time.perf_counter() - t0) / 20 * 1e3))
