"""The model generator as a tool (counterpart of deps/build.jl:27-48 + src/models/*/codegen.jl):
`python -m optimization_dynamics_amd.codegen --add spec.py` must produce everything a build needs -- device header,
oracle header, cooperative glue, translation unit, make variable, id registries -- with no hand edits.  The test adds a
ninth and a tenth model (tests/specs/pendulum_limit.py: one contact; tests/specs/sliding_block.py: a contact and a
friction cone) to a SCRATCH COPY of the sources, builds the library there (CPU tier: the host-emulation build; GPU tier:
hipcc for gfx950) and checks the new models against the oracle built from the same copy."""
import importlib.util
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SPEC = os.path.join(ROOT, "tests", "specs", "pendulum_limit.py")
SPEC2 = os.path.join(ROOT, "tests", "specs", "sliding_block.py")


def scratch_copy(dst):
    ign = shutil.ignore_patterns("build", "*.so", "*.o", "__pycache__", "_ref")
    for d in ("optimization_dynamics_amd", "oracle", "include"):
        shutil.copytree(os.path.join(ROOT, d), os.path.join(dst, d), ignore=ign)
    os.makedirs(os.path.join(dst, "tests"))
    shutil.copytree(os.path.join(ROOT, "tests", "host_emu"), os.path.join(dst, "tests", "host_emu"), ignore=ign)
    return dst


def add_model(root):
    env = dict(os.environ, PYTHONPATH=ROOT)
    for sp_, nm in ((SPEC, "pendulum_limit"), (SPEC2, "sliding_block")):
        out = subprocess.run([sys.executable, "-m", "optimization_dynamics_amd.codegen", "--root", root, "--add", sp_],
                             cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-3000:]
        assert "registered: " + nm in out.stdout
    g = os.path.join(root, "optimization_dynamics_amd", "csrc", "gen")
    for f in ("pendulum_limit.h", "coop_pendulum_limit.h", "sliding_block.h", "coop_sliding_block.h", "model_list.h", "models.mk"):
        assert os.path.exists(os.path.join(g, f)), f
    assert "X(pendulum_limit, 8)" in open(os.path.join(g, "model_list.h")).read()
    assert "X(sliding_block, 9)" in open(os.path.join(g, "model_list.h")).read()
    assert "pendulum_limit" in open(os.path.join(g, "models.mk")).read()
    assert os.path.exists(os.path.join(root, "optimization_dynamics_amd", "csrc", "od_model_pendulum_limit.hip"))
    assert "&pendulum_limit_model" in open(os.path.join(root, "oracle", "gen", "models_gen.h")).read()


def load_oracle(root):
    subprocess.check_call(["make", "-C", os.path.join(root, "oracle")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    spec = importlib.util.spec_from_file_location("scratch_oracle", os.path.join(root, "oracle", "oracle.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def check_new_model(lib, O, device):
    from optimization_dynamics_amd import dynamics as dyn, models
    assert lib.model_ids["pendulum_limit"] == 8 and lib.cdll.od_model_id(b"pendulum_limit") == 8 and lib.cdll.od_num_models() == 10
    assert lib.model_ids["hopper"] == 7                              # the built-in ids do not move
    m = models.from_library(lib, "pendulum_limit")
    assert (m.nq, m.nu, m.nc) == (1, 1, 1)
    rng = np.random.default_rng(3)
    B = 512
    q1 = rng.uniform(-1.0, 0.75, B)
    q1[: B // 4] = 0.8 - 1e-3 * rng.uniform(0, 1, B // 4)           # a quarter on the joint limit
    q2 = np.minimum(q1 + 0.05 * rng.normal(0, 1, B), 0.8 - 1e-4)
    X = np.vstack([q1, q2]); U = rng.normal(0, 1, (1, B))
    im = dyn.ImplicitDynamics(m, 0.05, r_tol=1e-8, kappa_eval_tol=1e-4, kappa_grad_tol=1e-3, device=device, lib=lib)
    sim = O.make_sim("pendulum_limit", 0.05, kappa_tol=1e-4, kappa_grad_tol=1e-3)
    Do, DXo, DUo, bad = O.step_grad_batch(sim, X, U)
    for mode in (1, 2):                                             # lane-per-problem and cooperative kernels
        im.set_cooperative(mode)
        assert bool(lib.cdll.od_uses_cooperative(im._h, B)) == (mode == 2)
        D, DX, DU, st, it = [t.cpu().numpy() for t in im.step_grad(torch.tensor(X), torch.tensor(U))]
        ok = (st & 3) == 3
        assert ok.mean() > 0.99
        assert (D[1, ok] <= 0.8 + 1e-6).all()                       # the limit holds
        assert (np.abs(D - Do).max(0) / np.maximum(1e-2, np.abs(Do).max(0)))[ok].max() < 1e-6
        G, Go = np.concatenate([DX, DU], 1).reshape(-1, B), np.concatenate([DXo, DUo], 1).reshape(-1, B)
        rel = np.abs(G - Go).max(0) / np.maximum(np.abs(Go).max(0), 1e-12)
        assert rel[ok].max() < 1e-4
    assert (D[1, : B // 4] > 0.79).mean() > 0.2                     # the contact branch was exercised
    check_block_model(lib, O, device)


def check_block_model(lib, O, device):
    """the model with a friction cone: both kernel families against the oracle, and Coulomb's law on the solution"""
    from optimization_dynamics_amd import dynamics as dyn, models
    assert lib.model_ids["sliding_block"] == 9
    m = models.from_library(lib, "sliding_block")
    assert (m.nq, m.nu, m.nc) == (2, 2, 1)
    rng = np.random.default_rng(5)
    B, h, mu = 512, 0.05, 0.5
    x1 = rng.normal(0, 1, B); y1 = np.abs(rng.normal(0, 0.3, B)); y1[: B // 2] = rng.uniform(0, 2e-3, B // 2)   # half on the floor
    vx, vy = rng.normal(0, 1.0, B), rng.normal(0, 0.3, B)
    X = np.vstack([x1, y1, x1 + h * vx, np.maximum(y1 + h * vy, 0.0)]); U = rng.normal(0, 2.0, (2, B))
    im = dyn.ImplicitDynamics(m, h, r_tol=1e-8, kappa_eval_tol=1e-4, kappa_grad_tol=1e-3, device=device, lib=lib)
    assert list(m.friction) == [mu]                                 # the spec's default coefficient, through od_default_friction
    sim = O.make_sim("sliding_block", h, kappa_tol=1e-4, kappa_grad_tol=1e-3)
    Do, DXo, DUo, bad = O.step_grad_batch(sim, X, U)
    assert bad == 0
    for mode in (1, 2):
        im.set_cooperative(mode)
        assert bool(lib.cdll.od_uses_cooperative(im._h, B)) == (mode == 2)
        D, DX, DU, st, it = [t.cpu().numpy() for t in im.step_grad(torch.tensor(X), torch.tensor(U))]
        ok = (st & 3) == 3
        assert ok.mean() > 0.99
        assert (D[3, ok] > -1e-6).all()                             # no penetration
        assert (np.abs(D - Do).max(0) / np.maximum(1e-2, np.abs(Do).max(0)))[ok].max() < 1e-6
        G, Go = np.concatenate([DX, DU], 1).reshape(-1, B), np.concatenate([DXo, DUo], 1).reshape(-1, B)
        rel = np.abs(G - Go).max(0) / np.maximum(np.abs(Go).max(0), 1e-12)
        assert rel[ok].max() < 1e-4
        gam, b, _, _, st2 = im.contact_forces(torch.tensor(X), torch.tensor(U), grads=False)
        gam, b = gam.cpu().numpy()[0], b.cpu().numpy()[0]
        assert (np.abs(b) <= mu * gam + 1e-3).all()                 # inside the friction cone (to the central-path tolerance)
        vT = (D[2] - X[2]) / h
        sliding = (gam > 0.05) & (np.abs(vT) > 0.05)
        assert sliding.sum() > 20
        assert (np.sign(b[sliding]) == -np.sign(vT[sliding])).all()                       # friction opposes sliding
        assert np.abs(np.abs(b[sliding]) - mu * gam[sliding]).max() < 5e-3                 # ... at the cone's boundary


def test_add_a_ninth_model_emulated(tmp_path):
    root = scratch_copy(str(tmp_path))
    add_model(root)
    subprocess.check_call(["make", "-C", os.path.join(root, "tests", "host_emu"), "-j", "8"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    from optimization_dynamics_amd import _lib
    lib = _lib.Library(os.path.join(root, "tests", "host_emu", "libod_emu.so"))
    check_new_model(lib, load_oracle(root), "cpu")


@pytest.mark.gpu
def test_add_a_ninth_model_on_the_gpu(tmp_path):
    root = scratch_copy(str(tmp_path))
    add_model(root)
    subprocess.check_call(["make", "-C", os.path.join(root, "optimization_dynamics_amd", "csrc"), "-j", "16"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    from optimization_dynamics_amd import _lib
    lib = _lib.Library(os.path.join(root, "optimization_dynamics_amd", "libod_mi355x.so"))
    check_new_model(lib, load_oracle(root), "cuda:0")
