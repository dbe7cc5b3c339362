"""cartpole-with-friction iLQR alone (B problems x T = 50) -- target of rocprofv3"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import ilqr_checks as C
import optimization_dynamics_amd as od
lib = od.default_library()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
im, obj, x1, U0 = C.cartpole_problem(lib, "cuda:0", B, 50, seed=1)
if len(sys.argv) > 2:
    im.set_launch_config(int(sys.argv[2]), 0)          # problems per wavefront (0: automatic)
solver = od.ILQR(im, obj, 50)
x1t, Ut = torch.tensor(x1, device="cuda:0"), torch.tensor(U0, device="cuda:0")
solver.solve(x1t, Ut, max_iter=2)
torch.cuda.synchronize(); t0 = time.time()
X, U, J, hist = solver.solve(x1t, Ut, max_iter=15, obj_tol=0.0)
torch.cuda.synchronize(); dt = time.time() - t0
print("cartpole B=%d: %.2f ms per iLQR iteration (%d iterations)" % (B, dt / len(hist) * 1e3, len(hist)))
