import os, sys, time
ROOT='/root/repo'; sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+'/tests')
import numpy as np, torch
import ilqr_checks as C
import optimization_dynamics_amd as od
lib = od.default_library()
B,T=4096,60
dyn, obj, x1, U0 = C.rocket_problem(lib, 'cuda:0', B, T, dtype=torch.float32, seed=1)
solver = od.ILQR(dyn, obj, T)
x1t, Ut = torch.tensor(x1, device='cuda:0'), torch.tensor(U0, device='cuda:0')
solver.solve(x1t, Ut, max_iter=4, obj_tol=0.0)
torch.cuda.synchronize()
