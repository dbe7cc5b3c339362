"""BASELINE config 5 alone: rocket with the thrust-cone projection inside iLQR, fp32 (or fp64), B problems x T = 60 -- target of rocprofv3"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import ilqr_checks as C
import optimization_dynamics_amd as od
from optimization_dynamics_amd import _lib as _L
lib = _L.Library(os.environ["OD_LIB"]) if os.environ.get("OD_LIB") else od.default_library()   # (OD_LIB: a variant build)
dtype = torch.float64 if (len(sys.argv) > 2 and sys.argv[2] == "f64") else torch.float32
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dyn, obj, x1, U0 = C.rocket_problem(lib, "cuda:0", B, 60, dtype=dtype, seed=1)
solver = od.ILQR(dyn, obj, 60)
x1t, Ut = torch.tensor(x1, device="cuda:0"), torch.tensor(U0, device="cuda:0")
solver.solve(x1t, Ut, max_iter=2)
torch.cuda.synchronize(); t0 = time.time()
X, U, J, hist = solver.solve(x1t, Ut, max_iter=10, obj_tol=0.0)
torch.cuda.synchronize(); dt = time.time() - t0
print("rocket %s B=%d: %.2f ms per iLQR iteration (%d iterations)" % (dtype, B, dt / len(hist) * 1e3, len(hist)))
