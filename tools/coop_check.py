"""GPU check + timing: cooperative (16 lanes per problem) state kernels against the lane-per-problem kernels.
usage (on the GPU box): python tools/coop_check.py [B] [T]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import parity_checks as P, workloads as W
from optimization_dynamics_amd import _lib

lib = _lib.default_library()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 100
out = {}
for name in ["hopper", "acrobot_impact"]:
    X, U = W.knots(name, 4096, seed=11)
    im = P.make_im(name, lib, "cuda:0")
    Xd, Ud = torch.tensor(X, device="cuda:0"), torch.tensor(U, device="cuda:0")
    im.set_cooperative(1); ref = [t.cpu().numpy() for t in im.step_grad(Xd, Ud)]
    im.set_cooperative(2); got = [t.cpu().numpy() for t in im.step_grad(Xd, Ud)]
    g = W.grad_rel_err(np.concatenate([ref[1], ref[2]], 1), np.concatenate([got[1], got[2]], 1))
    out[name] = dict(state_max_abs_diff=float(np.abs(ref[0] - got[0]).max()), iters_equal=bool((ref[4] == got[4]).all()),
                     status_equal=bool((ref[3] == got[3]).all()), grad_rel_p99=float(np.quantile(g, .99)), grad_rel_max=float(g.max()))
    print(name, out[name], flush=True)
x1, U = W.hopper_rollout_inputs(B, T, seed=0)
im = P.make_im("hopper", lib, "cuda:0")
x1d, Ud = torch.tensor(x1, device="cuda:0"), torch.tensor(U, device="cuda:0")
res = {}
for mode in (1, 2):
    im.set_cooperative(mode)
    r = im.rollout(x1d, Ud)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        r = im.rollout(x1d, Ud)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 100
    res[mode] = [t.cpu().numpy() for t in r[:5]]
    out["rollout_ms_mode%d" % mode] = ms
    print("mode", mode, "rollout B=%d T=%d: %.3f ms" % (B, T, ms), flush=True)
e = np.abs(res[1][0] - res[2][0]).max(0)
out["rollout"] = dict(state_diff_t1=float(e[1].max()), state_diff_t10=float(e[min(10, T)].max()), state_diff_tend=float(e[-1].max()),
                      state_diff_tend_median=float(np.median(e[-1])), iters_equal_frac=float((res[1][4] == res[2][4]).mean()),
                      status_equal_frac=float((res[1][3] == res[2][3]).mean()))
print(out["rollout"])
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "coop_check.json"), "w"), indent=1)
