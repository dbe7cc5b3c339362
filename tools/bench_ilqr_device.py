"""ms per iLQR iteration with the whole iteration on the device (od_ilqr_iterate), directly enqueued and replayed from a HIP graph;
BASELINE config 5 (rocket, thrust-cone projection on the path, T = 60) and cartpole with joint friction.  One MI355X."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import ilqr_checks as C
import optimization_dynamics_amd as od
lib = od.default_library()
out = {}


def timed(fn, n):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def run(name, dyn, obj, x1, U0, T, B, n_it=10):
    x1t, Ut = torch.tensor(x1, device='cuda:0'), torch.tensor(U0, device='cuda:0')
    sol = od.ILQR(dyn, obj, T)
    d = sol.device_solver(B, max_iter=n_it, obj_tol=0.0)
    d.init(x1t, Ut); d.iterate(2)
    d.init(x1t, Ut)
    ms = timed(lambda: d.iterate(n_it), n_it)
    J = d.get()[2]
    info = d.info()
    # graph replay of one iteration
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        d.init(x1t, Ut); d.iterate(1); d.init(x1t, Ut)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            d.iterate(1)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(side)
        for _ in range(n_it):
            g.replay()
        e1.record(side); torch.cuda.synchronize()
        ms_graph = e0.elapsed_time(e1) / n_it
    # the host-composed loop for comparison
    t0 = time.time(); ref = sol.solve_stepwise(x1t, Ut, max_iter=n_it, obj_tol=0.0); torch.cuda.synchronize(); ms_step = (time.time() - t0) / len(ref[3]) * 1e3
    out[name] = dict(ms_per_iteration=ms, ms_per_iteration_graph=ms_graph, ms_per_iteration_stepwise_host_loop=ms_step,
                     iterations=info.iterations, J_mean=J.mean().item(), bad_linearisations=info.bad_linearisations,
                     knot_solves_per_s=B * T * (1 + len(sol.alphas)) / (ms * 1e-3))


for dtype in (torch.float32, torch.float64):
    for B in (1024, 4096):
        T = 60
        dyn, obj, x1, U0 = C.rocket_problem(lib, 'cuda:0', B, T, dtype=dtype, seed=1)
        run('rocket_projection %s B=%d T=%d' % (str(dtype).split('.')[-1], B, T), dyn, obj, x1, U0, T, B)
for B in (256, 4096):
    im, obj, x1, U0 = C.cartpole_problem(lib, 'cuda:0', B, 50, seed=1)
    run('cartpole_friction B=%d T=50' % B, im, obj, x1, U0, 50, B)
print(json.dumps(out, indent=1))
