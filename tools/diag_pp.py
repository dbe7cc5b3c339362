"""Planar push, BASELINE config 3 (gradient bundle N = 256 x 50 knots = 12 850 independent solves): where the time goes.
Run on the GPU box: python tools/diag_pp.py [tag]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import workloads as W, parity_checks as P
import optimization_dynamics_amd as od

lib = od.default_library()
dev = "cuda:0"
out = {}


def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(n):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts)), float(np.min(ts))


im = P.make_im("planar_push", lib, dev)
gb = od.GradientBundle(od.planarpush, N=256, eps=1e-4, seed=0)
X, U = W.knots("planar_push", 50, seed=2)
Xd, Ud = torch.tensor(X, device=dev), torch.tensor(U, device=dev)
out["bundle_ms(median,min)"] = timeit(lambda: od.gradient_batch(im, gb, Xd, Ud))
# the same 12 850 eval solves as independent knots (od_step): iteration statistics
Xt = Xd.repeat(1, 257).contiguous(); Ut = Ud.repeat(1, 257).contiguous()
out["step_12850_ms"] = timeit(lambda: im.step(Xt, Ut))
D, st, it = im.step(Xt, Ut)
it0 = it[0].cpu().numpy()
out["iters_mean"] = float(it0.mean()); out["iters_max"] = int(it0.max())
out["iters_hist"] = np.bincount(it0, minlength=1).tolist()
w16 = it0[: (it0.size // 16) * 16].reshape(-1, 16).max(1)
out["iters_max_per_16_mean"] = float(w16.mean()); out["iters_max_per_16_max"] = int(w16.max())
w8 = it0[: (it0.size // 8) * 8].reshape(-1, 8).max(1)
out["iters_max_per_8_mean"] = float(w8.mean())
w4 = it0[: (it0.size // 4) * 4].reshape(-1, 4).max(1)
out["iters_max_per_4_mean"] = float(w4.mean())
for ppw in (4, 8, 16, 32, 64):
    im.set_launch_config(ppw, 4 if ppw < 64 else 1)
    out["step_12850_ppw%d_ms" % ppw] = timeit(lambda: im.step(Xt, Ut))
im.set_launch_config(0, 0)
for B in (1024, 4096, 16384, 65536):
    Xb, Ub = W.knots("planar_push", B, seed=1)
    Xb, Ub = torch.tensor(Xb, device=dev), torch.tensor(Ub, device=dev)
    out["step_grad_B%d_ms" % B] = timeit(lambda: im.step_grad(Xb, Ub), n=5)
    out["step_B%d_ms" % B] = timeit(lambda: im.step(Xb, Ub), n=5)
print(json.dumps(out, indent=1))
tag = sys.argv[1] if len(sys.argv) > 1 else "r3"
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "diag_pp_%s.json" % tag), "w"), indent=1)
