"""BASELINE config 3 alone (planar push gradient bundle N = 256 x 50 knots), a few calls -- target of rocprofv3 runs"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import workloads as W, parity_checks as P
import optimization_dynamics_amd as od
lib = od.default_library()
im = P.make_im("planar_push", lib, "cuda:0")
gb = od.GradientBundle(od.planarpush, N=256, eps=1e-4, seed=0)
X, U = W.knots("planar_push", 50, seed=2)
Xd, Ud = torch.tensor(X, device="cuda:0"), torch.tensor(U, device="cuda:0")
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    od.gradient_batch(im, gb, Xd, Ud)
torch.cuda.synchronize()
