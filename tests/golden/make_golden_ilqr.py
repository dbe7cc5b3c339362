"""Regression vectors of the numpy AL-iLQR oracle (oracle/ilqr_np.py) -- THIS REPOSITORY'S OWN oracle output, not reference output (the
reference's IterativeLQR.jl cannot run here; oracle/gen_golden.jl produces its per-iteration record where Julia exists and
tests/test_reference_golden.py::test_ilqr_iterations_match_julia_reference compares).  `python tests/golden/make_golden_ilqr.py` rewrites
tests/golden/ilqr_v1.json: the decision sequence (accepted step index per iteration, multiplier round), final merit, violation and end
point of (a) examples/hopper.jl as shipped on the reference's own stage dimensions (solve_stages) and (b) the cartpole-with-friction task
with two augmented-Lagrangian rounds (solve), both driven by the C oracle's dynamics."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def cases():
    from oracle import ilqr_np as N
    from oracle import oracle as O
    out = {}
    sim = O.make_sim("hopper", 0.05, kappa_tol=1e-4, kappa_grad_tol=1e-3, friction=[0.5, 0.5])
    st, objT, conT, nti, x1, U0 = N.hopper_gait_stages(sim)
    r = N.solve_stages(st, objT, conT, nti, x1, U0, alphas=tuple(2.0 ** -i for i in range(17)), max_iter=10, max_al_iter=15, con_tol=1e-3, obj_tol=1e-3)
    out["hopper_gait_as_shipped"] = dict(steps=[l["step"] for l in r["log"]], rounds=[l["al"] for l in r["log"]], merit=r["J"], objective=r["objective"],
                                         violation=r["violation"], theta=[float(v) for v in r["U"][0][2:]], x_T=[float(v) for v in r["X"][-1][:8]])
    sim = O.make_sim("cartpole_friction", 0.05, kappa_tol=1e-4, kappa_grad_tol=1e-4, friction=[0.35, 0.35])
    step, lin = N.mechanical_dynamics(sim)
    goal = np.array([0.3, 0.0, 0.3, 0.0])
    p = N.Problem(step, lin, np.diag([0.0, 0.1, 0.0, 0.1]), np.diag([0.01]), 100.0 * np.eye(4), goal, goal_idx=[0, 1, 2, 3], goal=goal)
    rng = np.random.default_rng(1)
    x1 = rng.normal(0, 0.01, 4); x1[2:] = x1[:2]
    U0 = 0.4 + 1e-2 * rng.normal(size=(15, 1))
    r = N.solve(p, x1, U0, max_iter=8, max_al_iter=2, obj_tol=1e-7, con_tol=1e-4)
    out["cartpole_two_rounds"] = dict(x1=[float(v) for v in x1], U0=[float(v) for v in U0[:, 0]], steps=[l["step"] for l in r["log"]],
                                      rounds=[l["al"] for l in r["log"]], costs=[l["J"] for l in r["log"]], x_T=[float(v) for v in r["X"][-1]],
                                      violation=r["violation"], rho=r["rho"])
    return out


if __name__ == "__main__":
    d = cases()
    with open(os.path.join(HERE, "ilqr_v1.json"), "w") as f:          # (one line per entry)
        f.write("{\n" + ",\n".join(' "%s": {\n%s\n }' % (k, ",\n".join('  "%s": %s' % (kk, json.dumps(vv)) for kk, vv in v.items())) for k, v in d.items()) + "\n}\n")
    print("written", os.path.join(HERE, "ilqr_v1.json"))
