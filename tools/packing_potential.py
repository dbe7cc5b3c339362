"""What could a different assignment of the 4096 rollouts of the headline workload to wavefronts (four per wavefront, lockstep: a
wavefront's loop trips = sum over knots of the max over its four rollouts) gain, if the per-knot iteration counts were known in
advance (e.g. from the previous call on similar inputs)?  Host build; iteration counts of the real workload.
usage: python tools/packing_potential.py        (result of round 4: DESIGN.md 3.3)"""
import os, subprocess, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
import parity_checks as P
from optimization_dynamics_amd import _lib as _L
subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "host_emu"), "-j8"], stdout=subprocess.DEVNULL)
lib = _L.Library(os.path.join(ROOT, "tests", "host_emu", "libod_emu.so"))
B, T = 4096, 100
x, U = bench.workload_slice(0, B, B, T)
it = P.make_im("hopper", lib, "cpu").rollout_compact(torch.tensor(x), torch.tensor(U))[3].numpy()
n = np.maximum(it[0], it[1]).astype(np.int16)             # loop trips of a knot
tot = n.sum(0).astype(np.int64)
w = n.reshape(T, B // 4, 4).max(2).sum(0)
print("as given:                 wavefront trips mean %.1f max %d   (trajectories alone: mean %.1f max %d; %d with a knot at max_iter)"
      % (w.mean(), w.max(), tot.mean(), tot.max(), int((n >= 100).any(0).sum())))
p = np.argsort(tot)
w = n[:, p].reshape(T, B // 4, 4).max(2).sum(0)
print("sorted by total trips:    mean %.1f max %d" % (w.mean(), w.max()))
t0 = time.time()
un = np.ones(B, bool); waves = []
for s in np.argsort(-tot):                                # heaviest first; partners = the three that raise sum_t max least
    if not un[s]: continue
    un[s] = False; cur = n[:, s].copy()
    for _ in range(3):
        cand = np.nonzero(un)[0]
        j = cand[np.argmin(np.maximum(n[:, cand], cur[:, None]).sum(0))]
        un[j] = False; cur = np.maximum(cur, n[:, j])
    waves.append(cur.sum())
w = np.array(waves)
print("greedy profile matching:  mean %.1f max %d   (%.1f s of numpy for the matching)" % (w.mean(), w.max(), time.time() - t0))
