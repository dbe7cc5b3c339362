"""Thrust-cone projection on the HOST BUILD of the product sources (tests/host_emu: libod_emu.so, or libod_emu_devmath.so = the device's
reciprocal / rsqrt sequences and contracted multiply-adds): how often the product's solve, the literal oracle and the oracle with the
exact-boundary completions follow the exact-arithmetic path (oracle/arbiter.c::od_arbiter_soc_projection), how a single-precision handle
compares with a double-precision one with and without od_set_mixed_precision, and the stalled population of each.  No GPU needed.
    python tools/proj_paths_host.py [out.json]      (EMU=<path of the host build>, SEEDS=41,101,...)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from oracle import oracle as O
O.build()
import parity_checks as P
from optimization_dynamics_amd import _lib, models, rocket as rk
path = os.environ.get("EMU", os.path.join(ROOT, "tests", "host_emu", "libod_emu.so"))
lib = _lib.Library(path)
B = int(os.environ.get("B", "8192"))
out = {"library": os.path.basename(path), "knots_per_seed": B, "inputs": "tests/parity_checks.py::rocket_sweep_inputs (apex-heavy)", "rows": []}
for seed in [int(t) for t in os.environ.get("SEEDS", "41,101,202,303").split(",")]:
    X, U = P.rocket_sweep_inputs(B, seed, torch.float32)          # (floats, so that both precisions see the same numbers)
    E, oke, ite = O.arbiter_soc_projection_batch(12.5, U, True)
    Zo, _, sto, ito = O.soc_projection_batch(12.5, U, False)
    Zx, _, stx, itx = O.soc_projection_batch(12.5, U, False, exact_boundary=True)
    Pc = O.project_thrust_cone_batch(U, 12.5)
    sc = np.maximum(1.0, np.abs(Pc).max(0))
    i64 = rk.RocketInfo(models.rocket, 12.5, 0.05, dtype=torch.float64, device="cpu", lib=lib)
    i32 = rk.RocketInfo(models.rocket, 12.5, 0.05, dtype=torch.float32, device="cpu", lib=lib)
    Z64, _, st64, it64 = [t.double().numpy() if t is not None and t.dtype.is_floating_point else (None if t is None else t.numpy()) for t in i64.project_full(torch.tensor(U), grads=False)]
    Zm, _, stm, itm = [t.double().numpy() if t is not None and t.dtype.is_floating_point else (None if t is None else t.numpy()) for t in i32.project_full(torch.tensor(U), grads=False)]
    lib.check(lib.cdll.od_set_mixed_precision(i32._h, 0))
    Zf, _, stf, itf = [t.double().numpy() if t is not None and t.dtype.is_floating_point else (None if t is None else t.numpy()) for t in i32.project_full(torch.tensor(U), grads=False)]
    c64, cm, cf = (st64 & 0x10) != 0, (stm & 0x10) != 0, (stf & 0x10) != 0
    use = c64 & (sto == 1) & (oke == 1)
    dev = lambda Z: np.abs(Z[:3] - E[:3]).max(0) / sc
    row = dict(seed=seed, converged=dict(device_f64=int(c64.sum()), device_f32_mixed=int(cm.sum()), device_f32_float=int(cf.sum()), oracle=int((sto == 1).sum()),
                                         oracle_exact_boundary=int((stx == 1).sum()), arbiter=int((oke == 1).sum())),
               stalled_30_iterations_or_more=dict(device_f64=int((it64 >= 30).sum()), oracle=int((ito >= 30).sum()), oracle_exact_boundary=int((itx >= 30).sum()), arbiter=int((ite >= 30).sum())),
               on_exact_path_1e7=dict(device_f64=float((dev(Z64)[use] < 1e-7).mean()), device_f32_mixed_1e6=float((dev(Zm)[use & cm] < 1e-6).mean()),
                                      device_f32_float_1e6=float((dev(Zf)[use & cf] < 1e-6).mean()), device_f32_float_2e3=float((dev(Zf)[use & cf] < 2e-3).mean()),
                                      oracle_literal=float((dev(Zo)[use] < 1e-7).mean()), oracle_exact_boundary=float((dev(Zx)[use & (stx == 1)] < 1e-7).mean())),
               off_path_deviation_max=dict(device_f64=float(dev(Z64)[use].max()), oracle_literal=float(dev(Zo)[use].max()), oracle_exact_boundary=float(dev(Zx)[use & (stx == 1)].max())),
               f32_handle_vs_f64_handle_max=dict(mixed=float((np.abs(Zm[:3] - Z64[:3]).max(0) / sc)[c64 & cm].max()), float_only=float((np.abs(Zf[:3] - Z64[:3]).max(0) / sc)[c64 & cf].max())),
               device_f64_vs_oracle_exact_boundary=dict(within_1e6=float(((np.abs(Z64[:3] - Zx[:3]).max(0) / sc)[use & (stx == 1)] < 1e-6).mean()),
                                                       iterations_equal=float((it64[use & (stx == 1)] == itx[use & (stx == 1)]).mean())))
    out["rows"].append(row)
    print(json.dumps(row), flush=True)
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
