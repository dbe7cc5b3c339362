"""Throughput of the other BASELINE.json configs on one MI355X (not the contract bench)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import workloads as W, parity_checks as P
import optimization_dynamics_amd as od
lib = od.default_library()
dev = 'cuda:0'
out = {}
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t0) / n
for name, B in [('acrobot_impact', 1024), ('acrobot_impact', 4096), ('acrobot_impact', 262144), ('hopper', 4096), ('hopper', 262144), ('cartpole_friction', 262144), ('planar_push', 4096), ('planar_push', 65536)]:
    X, U = W.knots(name, B, seed=1)
    im = P.make_im(name, lib, dev); Xd, Ud = torch.tensor(X, device=dev), torch.tensor(U, device=dev)
    dt = timeit(lambda: im.step_grad(Xd, Ud))
    D, DX, DU, st, it = im.step_grad(Xd, Ud)
    out['step_grad %s B=%d' % (name, B)] = dict(ms=round(dt * 1e3, 3), units_per_s=B / dt, mean_iterations=float(it[0].double().mean()), max_iterations=int(it.max()),
                                               cooperative=bool(lib.cdll.od_uses_cooperative(im._h, B)))
    if B <= 4096:   # latency floor of the config: the same call on 16 knots
        dtl = timeit(lambda: im.step_grad(Xd[:, :16].contiguous(), Ud[:, :16].contiguous()))
        out['step_grad %s B=%d' % (name, B)]['latency_floor_ms'] = round(dtl * 1e3, 3)
gb = od.GradientBundle(od.planarpush, N=256, eps=1e-4, seed=0)
im = P.make_im('planar_push', lib, dev)
X, U = W.knots('planar_push', 50, seed=2); Xd, Ud = torch.tensor(X, device=dev), torch.tensor(U, device=dev)
dt = timeit(lambda: od.gradient_batch(im, gb, Xd, Ud))
out['bundle planar_push N=256 x 50 knots'] = dict(ms=round(dt * 1e3, 3), solves_per_s=50 * 257 / dt)
dtl = timeit(lambda: od.gradient_batch(im, gb, Xd[:, :1].contiguous(), Ud[:, :1].contiguous()))
out['bundle planar_push N=256 x 50 knots']['latency_floor_ms'] = round(dtl * 1e3, 3)      # one knot's 257 solves + fit
# rollouts: hopper over the batch size (automatic kernel choice), planar push T = 50
for B in (64, 1024, 2048, 4096, 8192, 16384, 65536):
    x1, U = W.hopper_rollout_inputs(B, 100, seed=0)
    imh = P.make_im('hopper', lib, dev); x1d, Ud2 = torch.tensor(x1, device=dev), torch.tensor(U, device=dev)
    o = [None]
    def f():
        r = imh.rollout_compact(x1d, Ud2, out=o[0]); o[0] = r[-1]
    dt = timeit(f, n=3)
    out['rollout hopper T=100 B=%d' % B] = dict(ms=round(dt * 1e3, 3), units_per_s=B * 100 / dt, cooperative=bool(lib.cdll.od_uses_cooperative(imh._h, B)))
for B in (1, 256, 2048):
    x1, U = P.planar_push_rollout_inputs(B, 50)
    imp = P.make_im('planar_push', lib, dev); x1d, Ud2 = torch.tensor(x1, device=dev), torch.tensor(U, device=dev)
    dt = timeit(lambda: imp.rollout(x1d, Ud2), n=3)
    out['rollout planar_push T=50 B=%d' % B] = dict(ms=round(dt * 1e3, 3), units_per_s=B * 50 / dt)
for dtype in (torch.float64, torch.float32):
    info = od.RocketInfo(od.rocket, 12.5, 0.05, dtype=dtype, device=dev)
    X, U = W.rocket_inputs(65536, seed=3); Xd, Ud = torch.tensor(X, device=dev), torch.tensor(U, device=dev)
    dt = timeit(lambda: info.solve(Xd, Ud, project=True, grads=True))
    out['rocket_proj f+fx+fu %s B=65536' % str(dtype)] = dict(ms=round(dt * 1e3, 3), units_per_s=65536 / dt)
print(json.dumps(out, indent=1))
