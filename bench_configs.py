"""The other BASELINE.json configs on one MI355X, measured by bench.py next to the headline (`aux_config_2 / 3 / 5` of its JSON
line): each with the kernel names, HIP-event time, units per second, the algorithmic flops and bytes per unit of SURVEY.md 8(d)
with that model's generated operation counts (csrc/gen/stats.json) and the iteration counts measured on the workload, the
roofline fraction against the vector-ALU roof of the arithmetic type, the latency floor of the configuration (the same call on
one wavefront's worth of work) and the CPU oracle's rate on a bounded sample of the same workload.

  config 2  acrobot with joint limits, 1024 independent knots, fp64                       od_step_grad
  config 3  planar push, gradient bundle N = 256 x 50 knots, fp64                         od_bundle_grad (+ least-squares fit)
  config 5  rocket, thrust-cone SOCP projection inside the iLQR loop, T = 61, fp32        od_ilqr_iterate (4096 problems x 11 step
            sizes), inputs of examples/rocket.jl; the hover-thrust test problem beside it

Inputs come from tests/workloads.py and tests/ilqr_checks.py (the workloads the parity tests run); nothing under oracle/ is touched
except by the `cpu_baseline` legs.
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

PEAK_TFLOPS = {"f64": 78.6, "f32": 157.3}       # MI355X vector ALU peaks (/opt/skills/guides/MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0


def _stats():
    return json.load(open(os.path.join(ROOT, "optimization_dynamics_amd", "csrc", "gen", "stats.json")))


def flops_state(st, iters, c_cone=100.0):
    """the interior-point iterations of SURVEY.md 8(d): I (F_rz + 2/3 nz^3 + 4 nz^2 + 2 F_r + c_cone)"""
    nz = st["nz"]
    return iters * (st["ops_rz"] + (2.0 / 3.0) * nz ** 3 + 4.0 * nz ** 2 + 2.0 * st["ops_r"] + c_cone)


def flops_grad(st, ngc):
    """the implicit gradient: F_rz + F_rth + 2/3 nz^3 + 2 nz^2 ngc"""
    nz = st["nz"]
    return st["ops_rz"] + st["ops_rth"] + (2.0 / 3.0) * nz ** 3 + 2.0 * nz ** 2 * ngc


def _events(fn, n, sync):
    fn(); sync()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); sync()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


def config2(dev, cpu=True):
    import workloads as W, parity_checks as P
    import optimization_dynamics_amd as od
    lib = od.default_library()
    name, B = "acrobot_impact", 1024
    X, U = W.knots(name, B, seed=1)
    im = P.make_im(name, lib, dev)
    Xd, Ud = torch.tensor(X, device=dev), torch.tensor(U, device=dev)
    sync = lambda: torch.cuda.synchronize(dev)
    ms = _events(lambda: im.step_grad(Xd, Ud), 10, sync)
    D, DX, DU, st, it = im.step_grad(Xd, Ud)
    floor = _events(lambda: im.step_grad(Xd[:, :16].contiguous(), Ud[:, :16].contiguous()), 10, sync)
    s = _stats()[name]
    ie = float(it[0].double().mean().item())
    F = flops_state(s, ie) + flops_grad(s, 5)
    nq, nu = 2, 1
    bytes_unit = 8 * ((2 * nq + nu) + nq + nq * (2 * nq + nu))
    coop = bool(lib.cdll.od_uses_cooperative(im._h, B))
    out = dict(workload="acrobot with joint limits (src/models/acrobot, impact), %d independent knots, 25 %% on the joint limit, od_step_grad = f + fx + fu, fp64" % B,
               kernels=("k_step_state_coop<Coop_acrobot_impact> (one knot per 16 lanes)" if coop else "k_step_state<Model_acrobot_impact>") + " + k_grad_knots<Model_acrobot_impact>",
               ms=ms, units_per_s=B / (ms * 1e-3), dtype="f64", mean_iterations=ie, max_iterations=int(it.max().item()),
               nonconverged=int(((st & 3) != 3).sum().item()),
               algorithmic_flops_per_unit=F, algorithmic_bytes_per_unit=bytes_unit,
               roofline=dict(bound="fp64-valu", achieved=F * B / (ms * 1e-3) / 1e12, peak=PEAK_TFLOPS["f64"], unit="TFLOP/s",
                             frac=F * B / (ms * 1e-3) / 1e12 / PEAK_TFLOPS["f64"], hbm_frac=bytes_unit * B / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS),
               latency_floor_ms=floor,
               note="the launch waits for the knots that do not converge (max_iterations = max_iter: an infeasible joint-limit configuration on which the "
                    "line search backtracks ~17 times in each of its 100 iterations, in the oracle alike); everything else is done after latency_floor_ms")
    if cpu:
        from oracle import oracle as O
        sim = P.make_sim(O, name)
        import bench
        cores = bench.effective_cores()
        try:
            import ctypes
            ctypes.CDLL("libgomp.so.1").omp_set_num_threads(cores)
        except Exception:
            pass
        O.step_grad_batch(sim, X, U)
        t0 = time.time(); k = 0                      # (one call over the whole batch, repeated: the batch is small)
        while time.time() - t0 < 3.0:
            O.step_grad_batch(sim, X, U); k += 1
        dt = time.time() - t0
        out["cpu_baseline"] = dict(value=k * B / dt, unit="steps+grads/s", cores=cores, kind="port",
                                   sample="%d passes over the same %d knots, %.1f s, OpenMP over the knots; CPU restatement (f, fx, fu = 3 dense-LU interior-point solves per knot like the reference)" % (k, B, dt))
    return out


def config3(dev, cpu=True):
    import workloads as W, parity_checks as P
    import optimization_dynamics_amd as od
    lib = od.default_library()
    N, K = 256, 50
    gb = od.GradientBundle(od.planarpush, N=N, eps=1e-4, seed=0)
    im = P.make_im("planar_push", lib, dev)
    X, U = W.knots("planar_push", K, seed=2)
    Xd, Ud = torch.tensor(X, device=dev), torch.tensor(U, device=dev)
    sync = lambda: torch.cuda.synchronize(dev)
    ms = _events(lambda: od.gradient_batch(im, gb, Xd, Ud), 10, sync)
    floor = _events(lambda: od.gradient_batch(im, gb, Xd[:, :1].contiguous(), Ud[:, :1].contiguous()), 10, sync)
    # iteration counts of the eval solves: the nominal knots' (the perturbations are 1e-4: same counts)
    D, st, it = im.step(Xd, Ud)
    ie = float(it[0].double().mean().item())
    s = _stats()["planar_push"]
    nq, nu = 5, 2
    nzb = 2 * nq + nu
    solves = K * (N + 1)
    F_solve = flops_state(s, ie)
    F_fit = 2.0 * N * nzb * (nzb + nq) + (2.0 / 3.0) * nzb ** 3 + 2.0 * nzb ** 2 * nq      # normal equations + LU + back-solves, per knot
    F_total = solves * F_solve + K * F_fit
    bytes_total = 8.0 * (K * (2 * nq + nu) + nzb * N + K * nq * nzb + solves * nq * 2)     # inputs, eta, fitted dz, the samples' q3 written and read
    out = dict(workload="planar push contact QP, gradient bundle N = %d x %d knots (%d eval-simulator solves + %d least-squares fits), fp64" % (N, K, solves, K),
               kernels="k_bundle_coop3<Coop3_planar_push> (one solve per 8 lanes) + k_ls_fit_fused<12, 5>",
               ms=ms, units_per_s=solves / (ms * 1e-3), unit="solves/s", dtype="f64", mean_iterations=ie,
               algorithmic_flops_per_unit=F_solve, algorithmic_flops_per_fit=F_fit, algorithmic_bytes_per_step=bytes_total,
               roofline=dict(bound="fp64-valu", achieved=F_total / (ms * 1e-3) / 1e12, peak=PEAK_TFLOPS["f64"], unit="TFLOP/s",
                             frac=F_total / (ms * 1e-3) / 1e12 / PEAK_TFLOPS["f64"], hbm_frac=bytes_total / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS),
               latency_floor_ms=floor)
    if cpu:
        from oracle import oracle as O
        sim = P.make_sim(O, "planar_push")
        eta = np.asfortranarray(gb.eta)
        t0 = time.time(); k = 0
        while time.time() - t0 < 4.0 and k < K:
            O.gradient_bundle(sim, eta, X[:5, k], X[5:, k], U[:, k]); k += 1
        dt = time.time() - t0
        out["cpu_baseline"] = dict(value=k * (N + 1) / dt, unit="solves/s", cores=1, kind="port",
                                   sample="gradient! on the first %d of the %d knots (%d solves + %d Newton least-squares fits), %.1f s, one thread; CPU restatement" % (k, K, k * (N + 1), k, dt))
        # ... and on all host cores (single thread AND all cores): the knots are independent, one gradient! per thread (the C
        # oracle runs outside the interpreter lock)
        import os
        from concurrent.futures import ThreadPoolExecutor
        cores = os.cpu_count() or 1
        sims = [P.make_sim(O, "planar_push") for _ in range(cores)]
        def one(i):
            O.gradient_bundle(sims[i % cores], eta, X[:5, i % K], X[5:, i % K], U[:, i % K])
        t0 = time.time(); done = 0
        with ThreadPoolExecutor(cores) as ex:
            while time.time() - t0 < 6.0:
                list(ex.map(one, range(done, done + cores))); done += cores
        dta = time.time() - t0
        out["cpu_baseline"]["all_cores"] = dict(value=done * (N + 1) / dta, unit="solves/s", cores=cores, kind="port",
                                                 sample="gradient! on %d knots (cycling through the %d), one knot per thread, %.1f s" % (done, K, dta))
    return out


def _config5_one(dev, which, dtype, cpu, B=4096):
    import ilqr_checks as C
    import optimization_dynamics_amd as od
    from optimization_dynamics_amd import interior_point as IP
    lib = od.default_library()
    T = 60
    if which == "examples/rocket.jl inputs":
        dyn, obj, x1, U0 = C.config5_problem(lib, dev, B, dtype=dtype)
    else:
        dyn, obj, x1, U0 = C.rocket_problem(lib, dev, B, T, dtype=dtype, seed=1)
    x1t, Ut = torch.tensor(x1, device=dev), torch.tensor(U0, device=dev)
    sol = od.ILQR(dyn, obj, T)
    na = sol.alphas.numel()
    n_it = 10
    d = sol.device_solver(B, max_iter=n_it, obj_tol=0.0)
    d.init(x1t, Ut); d.iterate(2); d.init(x1t, Ut)
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); d.iterate(n_it); e1.record(); torch.cuda.synchronize(dev)
    ms = e0.elapsed_time(e1) / n_it
    X, U, J = d.get()
    info = d.info()
    # iteration counts of the two solves of a unit on the final nominal trajectory's knots (the generated raw solvers)
    Xk, Uk = X[:, :-1].reshape(12, -1)[:, ::7].contiguous(), U.reshape(3, -1)[:, ::7].contiguous()
    nk = Xk.shape[1]
    ipp = IP.InteriorPoint("rocket_projection", device=dev, lib=lib)
    z0 = torch.tensor([0.1, 0.1, 1.1, 0.1, 0.1, 0.1, 0.0, 0.1, 0.1, 1.1], device=dev)[:, None].repeat(1, nk)
    zp, _, stp, itp = ipp.solve(z0, torch.cat([Uk, torch.full((1, nk), 12.5, dtype=torch.float64, device=dev)]), diff_sol=False)
    ipd = IP.InteriorPoint("rocket_dynamics", device=dev, lib=lib)
    th = torch.cat([Xk, zp[:3], torch.full((1, nk), 0.05, dtype=torch.float64, device=dev)])
    _, _, std, itd = ipd.solve(Xk.clone(), th, diff_sol=False)
    conv = ((stp & 1) == 1)
    i_p, i_d = float(itp[0][conv].double().mean().item()), float(itd[0].double().mean().item())
    s = _stats()
    F_state = flops_state(s["rocket_projection"], i_p) + flops_state(s["rocket_dynamics"], i_d, c_cone=0.0)
    F_grad = flops_grad(s["rocket_projection"], 3) + flops_grad(s["rocket_dynamics"], 15) + 2.0 * 12 * 3 * 3
    n_state, n_lin = B * na * T, B * T
    n, m = 12, 3
    F_riccati = T * B * (2.0 * (2 * n ** 3 + 2 * n * n * m + n * m * m) + 2.0 * n * n * m + m ** 3 / 3.0)
    F_total = n_state * F_state + n_lin * (F_state + F_grad) + F_riccati
    es = 4 if dtype == torch.float32 else 8
    # candidates' states and controls written and read once (cost), the accepted one copied; linearisation written and read
    bytes_total = es * (2.0 * n_state * (n + m) + 2.0 * n_lin * (n * n + n * m + n + m) + n_lin * (m * n + m) * 2)
    nm = "f32" if dtype == torch.float32 else "f64"
    out = dict(ms_per_iteration=ms, problems=B, step_sizes=na, horizon=T + 1, dtype=nm,
               units_per_s=(n_state + n_lin) / (ms * 1e-3), unit="projected rocket steps/s (state solves of all candidates + the linearisation's f+fx+fu)",
               mean_iterations_projection=i_p, mean_iterations_dynamics=i_d,
               stalled_projections_in_linearisation=int(info.bad_linearisations),
               algorithmic_flops_per_state_unit=F_state, algorithmic_flops_per_gradient=F_grad, algorithmic_flops_per_iteration=F_total,
               algorithmic_bytes_per_iteration=bytes_total,
               roofline=dict(bound="%s-valu" % ("fp32" if nm == "f32" else "fp64"), achieved=F_total / (ms * 1e-3) / 1e12, peak=PEAK_TFLOPS[nm], unit="TFLOP/s",
                             frac=F_total / (ms * 1e-3) / 1e12 / PEAK_TFLOPS[nm], hbm_frac=bytes_total / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS))
    return out, J, (X, U)


def _riccati_pass(dev, B=4096, T=60, reps=20):
    """od_ilqr_backward alone at config 5's size (n = 12, m = 3, 4096 trajectories x 60 knots, per-knot Hessians as the C entry point
    takes them): k_ilqr_backward_mfma, the one GEMM-shaped kernel of the path -- HIP events around `reps` launches; HBM and MFMA rooflines"""
    import ilqr_checks as C
    import optimization_dynamics_amd as od
    dyn, obj, x1, U0 = C.rocket_problem(od.default_library(), dev, B, T, dtype=torch.float64, seed=1)
    x1t, Ut = torch.tensor(x1, device=dev), torch.tensor(U0, device=dev)
    sol = od.ILQR(dyn, obj, T)
    X, A, Bm, st = sol.linearize(x1t, Ut)
    quad = obj.expansion(X, Ut, torch.zeros(12, B, dtype=torch.float64, device=dev), 1.0)
    sol.backward(A, Bm, quad, 1e-6); torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        K, k, dV, bst = sol.backward(A, Bm, quad, 1e-6)
    e1.record(); torch.cuda.synchronize(dev)
    ms = e0.elapsed_time(e1) / reps
    n, m = 12, 3
    per_knot_in, per_knot_out = 8 * (n * n + n * m + n + m + n * n + m * m + m * n), 8 * (m * n + m)
    bytes_launch = float(T * B * (per_knot_in + per_knot_out))
    F_alg = T * B * (2.0 * (2 * n ** 3 + 2 * n * n * m + n * m * m) + 2.0 * n * n * m + m ** 3 / 3.0)
    F_mfma = T * B * 9 * 2.0 * 16 * 16 * 4
    return dict(workload="od_ilqr_backward, n = 12, m = 3, %d trajectories x %d knots, fp64, per-knot Hessians" % (B, T),
                kernel="k_ilqr_backward_mfma<12, 3, double, true, 16> (v_mfma_f64_16x16x4_f64: 9 per knot and trajectory; one wavefront per trajectory, 16 per workgroup, knots staged through LDS)",
                ms=ms, factorised=int((bst == 1).sum().item()), algorithmic_bytes_per_launch=bytes_launch, algorithmic_bytes_per_knot=per_knot_in + per_knot_out,
                roofline=dict(bound="hbm", achieved=bytes_launch / (ms * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit="GB/s", frac=bytes_launch / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS),
                mfma=dict(executed_tflops=F_mfma / (ms * 1e-3) / 1e12, algorithmic_tflops=F_alg / (ms * 1e-3) / 1e12, peak=PEAK_TFLOPS["f64"],
                          executed_frac=F_mfma / (ms * 1e-3) / 1e12 / PEAK_TFLOPS["f64"],
                          note="16 x 16 x 4 tiles on 12 x 12 / 12 x 3 blocks: 1.9 x the algorithmic flops; tools/ubench/mfma_f64.hip measures 27 ns per MFMA per SIMD (77 TFLOP/s)"))


def config5(dev, cpu=True):
    out = dict(workload="rocket soft landing, thrust-cone SOCP projection inside the dynamics (f_rocket_proj and its implicit gradients), iLQR iteration on the "
                        "device (od_ilqr_iterate: expansion, Riccati pass with regularisation retry, closed-loop rollouts of 11 step sizes, cost, Armijo "
                        "selection, copy, linearisation, bookkeeping -- no host synchronisation), 4096 problems, T = 61",
               kernels="k_rocket_rollout<T> (closed loop, 45 056 candidates) + k_ilqr_backward_mfma<12, 3, T, false, 16> (Riccati pass on v_mfma_f64_16x16x4_f64, one wavefront per trajectory) + k_rocket<T> (245 760 knots x f+fx+fu) + k_quad_cost<T, 12, 3> + k_il_*")
    for which in ("examples/rocket.jl inputs", "hover-thrust test problem"):
        for dtype in (torch.float32, torch.float64):
            key = "%s, %s" % (which, "fp32" if dtype == torch.float32 else "fp64")
            r, J, _ = _config5_one(dev, which, dtype, cpu)
            r["cost_mean_after_10_iterations"] = float(J.mean().item())
            out[key] = r
    # the same iteration at the batch that fills the chip's 1024 SIMDs with one wavefront of candidates each (5957 problems x 11 step sizes =
    # 65 527 candidates = 1024 wavefronts; at 4096 problems 704 wavefronts leave 30 % of the SIMDs idle by construction): where the forward
    # pass's roofline fraction stands when occupancy is not the limit
    for dtype in (torch.float32, torch.float64):
        try:
            r, J, _ = _config5_one(dev, "hover-thrust test problem", dtype, cpu, B=5957)
            out["chip-filling batch (5957 problems), hover-thrust test problem, %s" % ("fp32" if dtype == torch.float32 else "fp64")] = r
        except Exception as e:
            out["chip-filling batch, %s" % dtype] = {"error": repr(e)}
    try:
        out["riccati_pass"] = _riccati_pass(dev)
    except Exception as e:
        out["riccati_pass"] = {"error": repr(e)}
    if cpu:
        from oracle import oracle as O
        import ilqr_checks as C
        import optimization_dynamics_amd as od
        dyn, obj, x1, U0 = C.config5_problem(od.default_library(), dev, 8, dtype=torch.float64)
        rng = np.random.default_rng(0)
        t0 = time.time(); k = 0
        x = x1[:, 0].copy()
        while time.time() - t0 < 3.0:
            ok, y, dx, du = O.rocket_proj(0.05, 12.5, x, U0[:, k % 60, 0] + 1e-3 * rng.normal(size=3)); k += 1
        dt = time.time() - t0
        out["cpu_baseline"] = dict(value=k / dt, unit="projected rocket steps (f+fx+fu)/s", cores=1, kind="port",
                                   sample="%d calls of f_rocket_proj + fx + fu (projection solve, dynamics solve, both implicit gradients) at the example's initial state, %.1f s, one thread; CPU restatement" % (k, dt))
        # ... and on all host cores: the oracle's batched entry points (OpenMP over the knots) -- projection with its gradient, then the
        # dynamics step with its gradient at the projected control, the work of f / fx / fu_rocket_proj on independent knots
        import os
        cores = os.cpu_count() or 1
        nb = 4096
        Xb = np.repeat(x1[:, :1], nb, axis=1) + 1e-3 * rng.normal(size=(12, nb))
        Ub = U0[:, rng.integers(0, 60, nb), 0] + 1e-3 * rng.normal(size=(3, nb))
        t0 = time.time(); done = 0
        while time.time() - t0 < 6.0:
            Zp, _, _, _ = O.soc_projection_batch(12.5, Ub, True)
            O.rocket_batch(0.05, Xb, Zp[:3], True); done += nb
        dta = time.time() - t0
        out["cpu_baseline"]["all_cores"] = dict(value=done / dta, unit="projected rocket steps (f+fx+fu)/s", cores=cores, kind="port",
                                                 sample="%d knots in batches of %d (soc_projection + gradient, then the dynamics step + gradient), OpenMP over the knots, %.1f s" % (done, nb, dta))
    return out


def all_configs(dev, cpu=True):
    res = {}
    for key, fn in (("aux_config_2", config2), ("aux_config_3", config3), ("aux_config_5", config5)):
        try:
            res[key] = fn(dev, cpu)
        except Exception as e:          # an auxiliary block must never take the headline line down
            res[key] = {"error": repr(e)}
    return res


if __name__ == "__main__":
    print(json.dumps(all_configs(torch.device("cuda", 0)), indent=1))
