"""Time od_rollout (hopper, B x T) for a given library build and cooperative mode.
usage: python tools/time_rollout.py [lib.so|-] [mode] [B] [T] [max_iter]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import parity_checks as P, workloads as W
from optimization_dynamics_amd import _lib
libp = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] != "-" else None
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
B = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
T = int(sys.argv[4]) if len(sys.argv) > 4 else 100
lib = _lib.Library(libp) if libp else _lib.default_library()
x1, U = W.hopper_rollout_inputs(B, T, seed=0)
im = P.make_im("hopper", lib, "cuda:0")
if len(sys.argv) > 5:
    im.set_options(max_iter=int(sys.argv[5]))
im.set_cooperative(mode)
x1d, Ud = torch.tensor(x1, device="cuda:0"), torch.tensor(U, device="cuda:0")
r = im.rollout(x1d, Ud); torch.cuda.synchronize()
ts = []
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(10):
        r = im.rollout(x1d, Ud)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 100)
it = r[4].cpu().numpy()
print("%s mode %d B=%d T=%d: %.3f ms (min of 3x10), mean iters %.2f" % (os.path.basename(libp or "default"), mode, B, T, min(ts), it.reshape(2, -1)[0].mean() if it.ndim == 1 else it[0].mean()))
