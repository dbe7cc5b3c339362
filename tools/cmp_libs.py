"""two builds of the library on the same knots: largest differences of states / gradients / iteration counts (GPU box)
usage: python tools/cmp_libs.py libA.so libB.so model [B] [mode]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import workloads as W, parity_checks as P
from optimization_dynamics_amd import _lib
name = sys.argv[3]; B = int(sys.argv[4]) if len(sys.argv) > 4 else 4099; mode = int(sys.argv[5]) if len(sys.argv) > 5 else 1
X, U = W.knots(name, B, seed=17)
out = []
for path in sys.argv[1:3]:
    lib = _lib.Library(path) if path != "-" else _lib.default_library()
    im = P.make_im(name, lib, "cuda:0"); im.set_cooperative(mode)
    out.append([t.cpu().numpy() for t in im.step_grad(torch.tensor(X, device="cuda:0"), torch.tensor(U, device="cuda:0"))])
a, b = out
G = lambda o: np.concatenate([o[1].reshape(-1, B), o[2].reshape(-1, B)], 0)
ga, gb = G(a), G(b)
rel = np.abs(ga - gb).max(0) / np.maximum(np.abs(ga).max(0), 1e-300)
print(name, "mode", mode, "state max diff", np.abs(a[0] - b[0]).max(), "iters equal", np.array_equal(a[4], b[4]), "grad: knots that differ", int((rel > 0).sum()), "of", B,
      "max rel diff", rel.max(), "p99", np.quantile(rel, 0.99))
