"""thrust-cone projection kernels on one MI355X (round 4: closed-form KKT solve, stall exit): od_soc_project on 65 536 random
controls, the closed-loop rollout of config 5's forward pass (4096 x 11 candidates, T = 60) on the test problem (hover thrust: no
stalled projections) and on the inputs of examples/rocket.jl (controls near the apex of the cone: 0.2 % of the projections
stall), with the stall exit on / off; fp32 and fp64.  `python tools/time_projection.py > gpurun_out/...`"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import ilqr_checks as C
import workloads as W
import optimization_dynamics_amd as od
from optimization_dynamics_amd import rocket as rk, models
lib = od.default_library()
dev = "cuda:0"


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


out = {}
for dtype in (torch.float32, torch.float64):
    nm = "f32" if dtype == torch.float32 else "f64"
    for stall in (1, 0):
        key = "%s stall_exit=%d" % (nm, stall)
        row = {}
        X, U = W.rocket_inputs(65536, seed=3)
        info = rk.RocketInfo(models.rocket, 12.5, 0.05, dtype=dtype, device=dev, lib=lib)
        lib.check(lib.cdll.od_set_projection_stall_exit(info._h, stall))
        Ud = torch.tensor(U, device=dev, dtype=dtype)
        UP, DP, st = info.project(Ud, grads=True)
        row["od_soc_project 65536 (grad) ms"] = timeit(lambda: info.project(Ud, grads=True))
        row["od_soc_project 65536 (no grad) ms"] = timeit(lambda: info.project(Ud, grads=False))
        row["od_soc_project nonconverged"] = int(((st & 0x30) != 0x30).sum().item())
        Xd = torch.tensor(X, device=dev, dtype=dtype)
        row["od_rocket f+fx+fu projected 65536 ms"] = timeit(lambda: info.solve(Xd, Ud, project=True, grads=True))
        for pname, prob in (("test problem", lambda: C.rocket_problem(lib, dev, 4096, 60, dtype=dtype, seed=1)), ("config 5 inputs", lambda: C.config5_problem(lib, dev, 4096, dtype=dtype))):
            dyn, obj, x1, U0 = prob()
            lib.check(lib.cdll.od_set_projection_stall_exit(dyn.info._h, stall))
            x1t, Ut = torch.tensor(x1, device=dev), torch.tensor(U0, device=dev)
            solver = od.ILQR(dyn, obj, 60)
            Xn, A, Bm, s0, _, _ = dyn.rollout(x1t, Ut)
            K, k, dV, bst = solver.backward(A, Bm, obj.expansion(Xn, Ut.double(), None, 0.0), 1e-6)
            fn = lambda: rk._rocket_rollout(dyn.info, x1t, Ut, True, policy=(solver.alphas, Xn, K, k))
            Xc, Uc, cst = fn()
            row["closed-loop rollout 45056 x 60, %s, ms" % pname] = timeit(fn, 5)
            row["closed-loop rollout, %s, stalled projections" % pname] = int(((cst & 0x10) == 0).sum().item())
            row["open-loop rollout 4096 x 60, %s, ms" % pname] = timeit(lambda: rk._rocket_rollout(dyn.info, x1t, Ut, True), 5)
            d = solver.device_solver(4096, max_iter=10, obj_tol=0.0)
            d.init(x1t, Ut); d.iterate(2); d.init(x1t, Ut)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); d.iterate(10); e1.record(); torch.cuda.synchronize()
            row["iLQR iteration (od_ilqr_iterate) 4096 problems, %s, ms" % pname] = e0.elapsed_time(e1) / 10
            row["iLQR cost after 10 iterations, %s" % pname] = d.get()[2].mean().item()
        out[key] = row
print(json.dumps(out, indent=1))
