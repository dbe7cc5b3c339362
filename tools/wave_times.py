"""Per-wavefront duration of the cooperative rollout kernel (development record, profiles/r2_wave_times.txt).
Needs a library built from sources patched with tools/wave_times.patch (each wavefront overwrites the iteration
counts of its first knot with its wall_clock64 ticks, 100 MHz):
    patch -p0 < tools/wave_times.patch ; make -C optimization_dynamics_amd/csrc OUT=../../variants/libod_wt.so ; patch -R -p0 < tools/wave_times.patch
usage (GPU box): python tools/wave_times.py [variants/libod_wt.so]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import parity_checks as P, workloads as W
from optimization_dynamics_amd import _lib
lib = _lib.Library(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "variants", "libod_wt.so"))
B, T = 4096, 100
x1, U = W.hopper_rollout_inputs(B, T, seed=0)
im = P.make_im("hopper", lib, "cuda:0")
x1d, Ud = torch.tensor(x1, device="cuda:0"), torch.tensor(U, device="cuda:0")
for rep in range(3):
    r = im.rollout(x1d, Ud); torch.cuda.synchronize()
it = r[4].cpu().numpy()
w = it[0, 0, :].astype(np.int64).reshape(-1, 4)[:, 0]
s = it[1, 0, :].astype(np.int64).reshape(-1, 4)[:, 0]
s = (s - s.min()) % (1 << 31)
print("hopper 4096 x 100, 1024 wavefronts of 4 rollouts, wall_clock64 (100 MHz)")
print("wavefront duration [ms]: mean %.3f  min %.3f  p50 %.3f  p90 %.3f  p99 %.3f  max %.3f" % tuple(x / 1e5 for x in (w.mean(), w.min(), np.median(w), np.quantile(w, .9), np.quantile(w, .99), w.max())))
print("wavefronts slower than 1.05 x the median: %d" % int((w > 1.05 * np.median(w)).sum()))
print("start spread: %.1f us; the last wavefront ends at %.3f ms, the mean one at %.3f ms" % (s.max() / 100.0, (s + w).max() / 1e5, (s + w).mean() / 1e5))
