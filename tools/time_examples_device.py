"""The reference's examples solved end to end on the device (od_ilqr_solve), wall time per solve on one MI355X, 1 / 64 / 1024 problems
(perturbed initial controls): acrobot swing-up, cartpole (frictionless), planar push rotate / translate, the hopper's gait problem (initial configuration fixed), rocket landing with the
thrust-cone projection and its constraints (fp64 and fp32).  `python tools/time_examples_device.py > profiles/r5_examples_device.json`"""
import os, sys, time, json, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import ilqr_checks as C
import optimization_dynamics_amd as od
from optimization_dynamics_amd import ilqr as IL
lib = od.default_library(); dev = "cuda:0"
out = {}


def acrobot(B, mode="impact"):
    h, T = 0.05, 100
    if mode == "impact":
        im = od.ImplicitDynamics(od.acrobot_impact, h, r_tol=1e-8, kappa_eval_tol=1e-4, kappa_grad_tol=1e-3, device=dev, lib=lib)
    else:
        im = od.ImplicitDynamics(od.acrobot_nominal, h, r_tol=1e-8, kappa_eval_tol=1.0, kappa_grad_tol=1.0, device=dev, lib=lib)
    I2 = np.eye(2); Q = 0.1 / h ** 2 * np.block([[I2, -I2], [-I2, I2]])
    xT = np.array([math.pi, 0.0, math.pi, 0.0])
    obj = IL.QuadraticObjective(Q, np.eye(1), Q, x_ref=np.zeros(4), goal_idx=[0, 1, 2, 3], goal=xT, device=dev)
    U0 = np.stack([1e-3 * np.random.default_rng(1 + b).normal(size=(1, T)) for b in range(B)], axis=-1)
    return im, obj, np.zeros((4, B)), U0, T, dict(max_iter=50, max_al_iter=20, con_tol=1e-3, obj_tol=1e-5), tuple(2.0 ** -i for i in range(11))


def run(name, make, B, bundle=None):
    im, obj, x1, U0, T, opts, alphas = make(B)
    sol = IL.ILQR(im, obj, T, alphas=alphas, bundle=bundle)
    x1t, Ut = torch.tensor(x1, device=dev), torch.tensor(U0, device=dev)
    sol.solve(x1t, Ut, **dict(opts, max_iter=2, max_al_iter=1))        # buffers, lazy loads
    sol._dev = None
    torch.cuda.synchronize(); t0 = time.perf_counter()
    X, U, J, hist = sol.solve(x1t, Ut, **opts)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    info = sol._dev.info(); fl, viol, rho = sol._dev.status()
    out["%s, %d problem(s)" % (name, B)] = dict(seconds=dt, lockstep_iterations=info.iterations, multiplier_rounds=info.al_iterations,
                                                 ms_per_iteration=dt / max(info.iterations, 1) * 1e3,
                                                 fraction_at_con_tol=float((viol < opts["con_tol"]).double().mean().item()), max_violation=float(viol.max().item()))


for B in (1, 64, 1024):
    run("acrobot swing-up (examples/acrobot.jl)", acrobot, B)
    run("acrobot swing-up AS SHIPPED (`:nominal`: no joint limits; examples/acrobot.jl:11-12)", lambda b: acrobot(b, "nominal"), B)
    run("cartpole frictionless (examples/cartpole.jl)", lambda b: (lambda r: (r[0], r[1], r[2], r[3], r[5], r[6], tuple(2.0 ** -i for i in range(17))))(C.cartpole_example(lib, dev, "frictionless", b)), B)
    for mode in ("rotate", "translate"):
        run("planar push %s (examples/planar_push.jl)" % mode, lambda b, mode=mode: (lambda r: (r[0], r[1], r[2], r[3], r[5], r[6], tuple(2.0 ** -i for i in range(17))))(C.planar_push_example(lib, dev, mode, b)), B)
    from optimization_dynamics_amd import gradient_bundle as gbm
    for mode in ("rotate", "translate"):
        run("planar push %s with GB = true: GradientBundle N = 50 as the linearisation (examples/planar_push.jl:15, od_ilqr_set_gradient_bundle)" % mode,
            lambda b, mode=mode: (lambda r: (r[0], r[1], r[2], r[3], r[5], r[6], tuple(2.0 ** -i for i in range(17))))(C.planar_push_example(lib, dev, mode, b)), B,
            bundle=gbm.GradientBundle(od.planarpush, N=50, eps=1.0e-4, seed=3))
    run("cartpole with joint friction 0.35 (examples/cartpole.jl `:friction`)", lambda b: (lambda r: (r[0], r[1], r[2], r[3], r[5], r[6], tuple(2.0 ** -i for i in range(17))))(C.cartpole_example(lib, dev, "friction", b)), B)
    run("rocket landing AS SHIPPED (`:nominal`: thrust limits as stage constraints, no projection; examples/rocket.jl)",
        lambda b: (lambda r: (r[0], r[1], r[2], r[3], 60, r[5], r[6]))(C.rocket_example_nominal_problem(lib, dev, b)), B)
    run("hopper gait, initial configuration fixed (examples/hopper.jl)", lambda b: (lambda r: (r[0], r[1], r[2], r[3], r[5], r[6], tuple(2.0 ** -i for i in range(17))))(C.hopper_example(lib, dev, b)), B)
    run("hopper gait AS SHIPPED: initial configurations optimised, nonlinear foot constraint (examples/hopper.jl, od_ilqr_set_parameter_stage)",
        lambda b: (lambda r: (r[0], r[1], r[2], r[3], r[5], r[6], tuple(2.0 ** -i for i in range(17))))(C.hopper_example_full(lib, dev, b)), B)
    for dt_ in (torch.float64, torch.float32):
        run("rocket landing, projection + constraints, %s (examples/rocket.jl)" % ("fp64" if dt_ == torch.float64 else "fp32"),
            lambda b, dt_=dt_: (lambda r: (r[0], r[1], r[2], r[3], 60, r[5], r[6]))(C.rocket_example_problem(lib, dev, b, dtype=dt_)), B)
print(json.dumps(out, indent=1))
