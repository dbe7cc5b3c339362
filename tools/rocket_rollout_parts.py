"""where the time of the rocket's closed-loop rollouts goes (config 5's forward pass: 4096 x 11 candidates, T = 60, fp32):
with / without the thrust-cone projection, closed / open loop (GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import ilqr_checks as C
import optimization_dynamics_amd as od
from optimization_dynamics_amd import rocket as rk, _lib as _L
lib = _L.Library(os.environ["OD_LIB"]) if os.environ.get("OD_LIB") else od.default_library()
dtype = torch.float64 if "f64" in sys.argv else torch.float32
B, T, na = 4096, 60, 11
dyn, obj, x1, U0 = C.rocket_problem(lib, "cuda:0", B, T, dtype=dtype, seed=1)
info = dyn.info
x1t, Ut = torch.tensor(x1, device="cuda:0"), torch.tensor(U0, device="cuda:0")
solver = od.ILQR(dyn, obj, T)
X, A, Bm, st, _, _ = dyn.rollout(x1t, Ut)
lam = torch.zeros(12, B, dtype=torch.float64, device="cuda:0")
K, k, dV, bst = solver.backward(A, Bm, obj.expansion(X, Ut.double(), lam, 1.0), 1e-6)
alphas = solver.alphas
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
Ubig = Ut.repeat(1, 1, na)
x1big = x1t.repeat(1, na)
for proj in (True, False):
    print("project=%s closed loop (%d candidates): %.3f ms" % (proj, na * B, timeit(lambda: rk._rocket_rollout(info, x1t, Ut, proj, policy=(alphas, X, K, k)))))
    print("project=%s open loop   (%d rollouts):   %.3f ms" % (proj, na * B, timeit(lambda: rk._rocket_rollout(info, x1big, Ubig, proj))))
    print("project=%s open loop   (%d rollouts):   %.3f ms" % (proj, B, timeit(lambda: rk._rocket_rollout(info, x1t, Ut, proj))))
# lane mapping: fewer candidates per wavefront (more wavefronts in flight per SIMD)
for ppw in (64, 32, 16):
    lib.check(lib.cdll.od_set_launch_config(info._h, ppw, 0))
    print("ppw=%d project=True closed loop (%d candidates): %.3f ms" % (ppw, na * B, timeit(lambda: rk._rocket_rollout(info, x1t, Ut, True, policy=(alphas, X, K, k)))))
lib.check(lib.cdll.od_set_launch_config(info._h, 0, 0))
