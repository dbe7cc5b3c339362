"""latency of the reference-signature callbacks on host vectors (BASELINE config 1: cartpole with joint friction)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import parity_checks as P
import optimization_dynamics_amd as od
from optimization_dynamics_amd import dynamics as dyn
lib = od.default_library()
for name, n, nu in (("cartpole_friction", 4, 1), ("hopper", 8, 2), ("planar_push", 10, 2)):
    im = P.make_im(name, lib, "cuda:0")
    import workloads as W
    X, U = W.knots(name, 1, seed=3)
    x, u = X[:, 0].copy(), U[:, 0].copy()
    d = np.zeros(n); dx = np.zeros((n, n)); du = np.zeros((n, nu))
    res = {}
    for nm, fn in (("f", lambda: dyn.f(d, im, x, u)), ("fx", lambda: dyn.fx(dx, im, x, u)), ("fu", lambda: dyn.fu(du, im, x, u)), ("f+fx+fu (one solve)", lambda: dyn.ffxfu(d, dx, du, im, x, u))):
        for _ in range(20): fn()
        t0 = time.perf_counter()
        for _ in range(300): fn()
        res[nm] = (time.perf_counter() - t0) / 300 * 1e6
    print(name, {k: round(v, 1) for k, v in res.items()}, "us per call", flush=True)
