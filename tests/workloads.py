"""Seeded synthetic inputs per SURVEY.md 8(d) (shared by CPU-tier and GPU-tier tests)."""
import os

import numpy as np

CONFIGS = {
    # name: (h, kappa_eval, kappa_grad, friction)
    "acrobot_impact": (0.05, 1e-4, 1e-3, None),          # examples/acrobot.jl:15-23
    "acrobot_nominal": (0.05, 1.0, 1.0, None),           # examples/acrobot.jl:25-27
    "cartpole_friction": (0.05, 1e-4, 1e-4, [0.35, 0.35]),   # examples/cartpole.jl:15-21
    "cartpole_frictionless": (0.05, 1.0, 1.0, None),
    "planar_push": (0.1, 1e-4, 1e-2, None),              # examples/planar_push.jl:18-22
    "hopper": (0.05, 1e-4, 1e-3, [0.5, 0.5]),            # examples/hopper.jl:13,42
}


# OD_SEED_OFFSET=k shifts every seed of this module: `OD_SEED_OFFSET=3 pytest tests -m gpu` reruns the whole suite on
# other random inputs (a soak; the committed expectations that depend on the inputs -- iteration statistics, golden
# fixtures -- skip themselves)
SEED_OFFSET = int(os.environ.get("OD_SEED_OFFSET", "0"))


def knots(name, B, seed=1):
    """(X (2nq,B), U (nu,B)) knot-point batches exercising contact / no-contact branches."""
    rng = np.random.default_rng(seed + SEED_OFFSET)
    if name.startswith("acrobot"):
        q1 = np.stack([rng.uniform(-np.pi, np.pi, B), rng.uniform(-np.pi / 2 + 0.05, np.pi / 2 - 0.05, B)])
        k = B // 4                                         # 25 % pushed onto the joint limit
        q1[1, :k] = np.sign(rng.normal(size=k)) * (np.pi / 2 - 1e-3)
        q2 = q1 + 0.05 * rng.normal(size=(2, B))
        q2[1] = np.clip(q2[1], -np.pi / 2 + 1e-4, np.pi / 2 - 1e-4)
        U = rng.normal(size=(1, B))
    elif name.startswith("cartpole"):
        q1 = rng.normal(size=(2, B))
        q2 = q1 + 0.05 * rng.normal(size=(2, B))
        U = 3 * rng.normal(size=(1, B))
    elif name == "hopper":
        q = np.array([0, 0.55, 0, 0.5])[:, None]
        q1 = q + rng.normal(0, 0.02, (4, B))
        q2 = q1 + 0.5 * rng.normal(0, 0.02, (4, B))
        U = np.array([0, 9.81 * 3 * 0.5 * 0.05])[:, None] + rng.normal(size=(2, B))
    elif name == "planar_push":
        q = np.array([0, 0, 0, -0.1 - 1e-8, -0.01])[:, None]
        q1 = q + np.zeros((5, B))
        q1[3] -= np.abs(rng.normal(0, 0.01, B))
        q1[4] += rng.normal(0, 0.02, B)
        q2 = q1.copy()
        q2[3] += np.abs(rng.normal(0, 0.003, B))
        U = np.stack([rng.uniform(0, 1.5, B), rng.normal(0, 0.2, B)])
    else:
        raise KeyError(name)
    return np.vstack([q1, q2]), U


def hopper_rollout_inputs(B, T, seed=0, h=0.05, u_sigma=1.0):
    rng = np.random.default_rng(seed + SEED_OFFSET)
    q = np.array([0.0, 0.55, 0.0, 0.5])[:, None] + rng.normal(0.0, 0.02, (4, B))
    x1 = np.vstack([q, q])
    U = np.array([0.0, 9.81 * 3.0 * 0.5 * h])[:, None, None] + rng.normal(0.0, u_sigma, (2, T, B))
    return x1, U


def rocket_inputs(B, seed=1):
    rng = np.random.default_rng(seed + SEED_OFFSET)
    X = np.zeros((12, B))
    X[2] = 10.0 + rng.normal(0, 1, B)
    X[0:2] = rng.normal(0, 1, (2, B))
    X[3:6] = rng.normal(0, 0.1, (3, B))
    X[6:9] = rng.normal(0, 1, (3, B))
    X[9:12] = rng.normal(0, 0.2, (3, B))
    U = np.stack([rng.normal(0, 2, B), rng.normal(0, 2, B), rng.uniform(-2, 16, B)])
    return X, U


def grad_rel_err(G, Go):
    """per-sample max |G - Go| / max |Go| over the gradient entries; G: (..., B)"""
    import numpy as np
    B = G.shape[-1]
    a = np.abs(G - Go).reshape(-1, B).max(0)
    return a / np.maximum(np.abs(Go).reshape(-1, B).max(0), 1e-12)
