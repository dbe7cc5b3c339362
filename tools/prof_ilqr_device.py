"""one configuration of tools/bench_ilqr_device.py for rocprofv3 --kernel-trace --stats: config 5 (rocket, projection, T = 60), 4096
problems, `python tools/prof_ilqr_device.py [float32|float64] [B] [hover|config5]`"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import ilqr_checks as C
import optimization_dynamics_amd as od
dtype = torch.float32 if (len(sys.argv) < 2 or sys.argv[1] == 'float32') else torch.float64
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
T = 60
which = sys.argv[3] if len(sys.argv) > 3 else 'hover'      # hover: the hover-thrust test problem; config5: the inputs of examples/rocket.jl
if which == 'config5':
    dyn, obj, x1, U0 = C.config5_problem(od.default_library(), 'cuda:0', B, dtype=dtype)
else:
    dyn, obj, x1, U0 = C.rocket_problem(od.default_library(), 'cuda:0', B, T, dtype=dtype, seed=1)
x1t, Ut = torch.tensor(x1, device='cuda:0'), torch.tensor(U0, device='cuda:0')
d = od.ILQR(dyn, obj, T).device_solver(B, max_iter=20, obj_tol=0.0)
d.init(x1t, Ut)
d.iterate(20)
torch.cuda.synchronize()
print(d.info().iterations, d.get()[2].mean().item())
