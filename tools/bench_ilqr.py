"""iLQR iteration rate on one MI355X (next-row measurement, not the contract bench)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import ilqr_checks as C
import optimization_dynamics_amd as od
lib = od.default_library()
out = {}
for B, T in [(256, 50), (1024, 50), (4096, 50)]:
    im, obj, x1, U0 = C.cartpole_problem(lib, 'cuda:0', B, T, seed=1)
    solver = od.ILQR(im, obj, T)
    x1t, Ut = torch.tensor(x1, device='cuda:0'), torch.tensor(U0, device='cuda:0')
    solver.solve(x1t, Ut, max_iter=2)
    torch.cuda.synchronize(); t0 = time.time()
    X, U, J, hist = solver.solve(x1t, Ut, max_iter=15, obj_tol=0.0)
    torch.cuda.synchronize(); dt = time.time() - t0
    its = len(hist)
    J0 = obj.value(solver.linearize(x1t, Ut)[0], Ut)
    out['cartpole_friction B=%d T=%d' % (B, T)] = dict(iterations=its, ms_per_iteration=dt / its * 1e3,
        trajectory_iterations_per_s=B * its / dt, knot_solves_per_s=B * its * T * (1 + len(solver.alphas)) / dt,
        J0_mean=J0.mean().item(), Jf_mean=J.mean().item())
# BASELINE config 5: rocket, thrust-cone SOCP projection inside the dynamics, T = 61, fp32 and fp64
for dtype in (torch.float32, torch.float64):
    for B in (1024, 4096):
        T = 60
        dyn, obj, x1, U0 = C.rocket_problem(lib, 'cuda:0', B, T, dtype=dtype, seed=1)
        solver = od.ILQR(dyn, obj, T)
        x1t, Ut = torch.tensor(x1, device='cuda:0'), torch.tensor(U0, device='cuda:0')
        solver.solve(x1t, Ut, max_iter=2)
        torch.cuda.synchronize(); t0 = time.time()
        X, U, J, hist = solver.solve(x1t, Ut, max_iter=10, obj_tol=0.0)
        torch.cuda.synchronize(); dt = time.time() - t0
        its = len(hist)
        J0 = obj.value(solver.linearize(x1t, Ut)[0], Ut)
        out['rocket_projection %s B=%d T=%d' % (str(dtype).split('.')[-1], B, T)] = dict(
            iterations=its, ms_per_iteration=dt / its * 1e3, trajectory_iterations_per_s=B * its / dt,
            knot_solves_per_s=B * its * T * (1 + len(solver.alphas)) / dt, J0_mean=J0.mean().item(), Jf_mean=J.mean().item())
print(json.dumps(out, indent=1))
