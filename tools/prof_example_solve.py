"""one reference example solved on the device under rocprofv3 (kernel trace): where an iteration's time goes at batch 1 / 64
   rocprofv3 --kernel-trace --stats -d gpurun_out/prof_ex -- python tools/prof_example_solve.py acrobot_nominal 1"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import ilqr_checks as C
import optimization_dynamics_amd as od
from optimization_dynamics_amd import ilqr as IL
lib = od.default_library(); dev = "cuda:0"
which, B = sys.argv[1], int(sys.argv[2])
al17 = tuple(2.0 ** -i for i in range(17))
if which.startswith("acrobot"):
    im, obj, x1, U0 = C.acrobot_example(lib, dev, B, mode=which.split("_")[1])
    T, opts, alphas = 100, dict(max_iter=50, max_al_iter=20, con_tol=1e-3, obj_tol=1e-5), tuple(2.0 ** -i for i in range(11))
elif which == "hopper_full":
    im, obj, x1, U0, x1v, T, opts = C.hopper_example_full(lib, dev, B); alphas = al17
else:
    im, obj, x1, U0, xT, T, opts = C.cartpole_example(lib, dev, which.split("_")[1], B); alphas = al17
sol = IL.ILQR(im, obj, T, alphas=alphas)
x1t, Ut = torch.tensor(x1, device=dev), torch.tensor(U0, device=dev)
sol.solve(x1t, Ut, **dict(opts, max_iter=2, max_al_iter=1)); sol._dev = None
torch.cuda.synchronize(); t0 = time.perf_counter()
X, U, J, hist = sol.solve(x1t, Ut, **opts)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
info = sol._dev.info()
print("%s B=%d: %.4f s, %d iterations, %.3f ms per iteration" % (which, B, dt, info.iterations, dt / info.iterations * 1e3))
