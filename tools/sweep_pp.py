"""planar push: cooperative (8 lanes per problem) against lane-per-problem kernels over the batch size -> crossover"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import workloads as W, parity_checks as P
import optimization_dynamics_amd as od
from optimization_dynamics_amd import _lib
lib = _lib.Library(os.environ["OD_LIB"]) if os.environ.get("OD_LIB") else od.default_library(); dev = "cuda:0"   # (OD_LIB: a variant build)
OUT = os.environ.get("OD_SWEEP_OUT")
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(n):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
out = {}
im = P.make_im("planar_push", lib, dev)
for B in (256, 1024, 2048, 4096, 8192, 12288, 16384, 24576, 32768, 49152, 65536):
    X, U = W.knots("planar_push", B, seed=1)
    Xd, Ud = torch.tensor(X, device=dev), torch.tensor(U, device=dev)
    row = {}
    for mode, nm in ((1, "lane"), (2, "coop3")):
        im.set_cooperative(mode)
        row["step_" + nm] = timeit(lambda: im.step(Xd, Ud))
        row["step_grad_" + nm] = timeit(lambda: im.step_grad(Xd, Ud))
    out[B] = row
    print(B, {k: round(v, 4) for k, v in row.items()}, flush=True)
# rollouts: T = 50, small batches (examples/planar_push.jl runs ONE trajectory of T = 26)
T = 50
for B in (1, 16, 256, 2048):
    rng = np.random.default_rng(5)
    q0 = np.array([0.0, 0.0, 0.0, -0.1 - 1e-8, -0.01])[:, None] + np.r_[np.zeros((4, B)), rng.normal(0, 0.02, (1, B))]
    x1 = torch.tensor(np.vstack([q0, q0]), device=dev)
    U = np.zeros((2, T, B)); U[0, : T // 2] = rng.uniform(0.3, 0.6, (T // 2, B)); U[1] = rng.normal(0, 0.1, (T, B))
    Ud = torch.tensor(U, device=dev)
    row = {}
    for mode, nm in ((1, "lane"), (2, "coop3")):
        im.set_cooperative(mode)
        row["rollout_" + nm] = timeit(lambda: im.rollout(x1, Ud), n=5)
    out["rollout_T50_B%d" % B] = row
    print("rollout", B, {k: round(v, 4) for k, v in row.items()}, flush=True)
json.dump(out, open(OUT or os.path.join(ROOT, "gpurun_out", "sweep_pp.json"), "w"), indent=1)
