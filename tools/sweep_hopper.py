"""hopper rollouts T = 100: 16-lane cooperative, 8-lane cooperative, lane-per-problem kernels over the batch size"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import workloads as W, parity_checks as P
import optimization_dynamics_amd as od
from optimization_dynamics_amd import _lib
lib = _lib.Library(os.environ["OD_LIB"]) if os.environ.get("OD_LIB") else od.default_library(); dev = "cuda:0"   # (OD_LIB: a variant build)
OUT = os.environ.get("OD_SWEEP_OUT")
T = int(sys.argv[1]) if len(sys.argv) > 1 else 100
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(n):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
out = {}
im = P.make_im("hopper", lib, dev)
for B in (1024, 2048, 4096, 8192, 16384, 32768, 65536):
    x1, U = W.hopper_rollout_inputs(B, T, seed=0, u_sigma=1.0)
    x1d, Ud = torch.tensor(x1, device=dev), torch.tensor(U, device=dev)
    row = {}
    bufs = None
    for mode, nm in ((2, "coop16"), (3, "coop8"), (1, "lane")):
        im.set_cooperative(mode)
        o = [None]
        def f():
            r = im.rollout_compact(x1d, Ud, out=o[0]); o[0] = r[-1]
        row[nm] = timeit(f)
    out[B] = row
    print(B, {k: round(v, 3) for k, v in row.items()}, flush=True)
json.dump(out, open(OUT or os.path.join(ROOT, "gpurun_out", "sweep_hopper.json"), "w"), indent=1)
