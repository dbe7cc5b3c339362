"""sha256 of the device results of fixed workloads -- a bit-for-bit regression record across kernel changes that are meant to
keep every knot's arithmetic (GPU box):  python tools/headline_hash.py [out.json]"""
import hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bench, workloads as W, parity_checks as P
import optimization_dynamics_amd as od
from optimization_dynamics_amd import _lib
lib = _lib.Library(sys.argv[2]) if len(sys.argv) > 2 else od.default_library()      # (a variant build: tools/build_variants.sh)
dev = "cuda:0"
def h(*ts):
    m = hashlib.sha256()
    for t in ts:
        m.update(np.ascontiguousarray(t.cpu().numpy()).tobytes())
    return m.hexdigest()[:24]
out = {}
im = P.make_im("hopper", lib, dev)
for B, T in ((4096, 100), (1024, 100), (2048, 40), (8192, 30)):
    x1, U = bench.make_inputs(B, T, seed=0)
    X, G, st, it, _ = im.rollout_compact(torch.tensor(x1, device=dev), torch.tensor(U, device=dev))
    out["hopper_rollout_%dx%d" % (B, T)] = dict(X=h(X), G=h(G), st_it=h(st, it), it_sum=int(it.sum().item()), nonconv=int(((st & 3) != 3).sum().item()))
for name, mode in (("acrobot_impact", 2), ("acrobot_impact", 1), ("cartpole_friction", 2), ("hopper", 2), ("hopper", 3), ("planar_push", 3), ("planar_push", 1)):
    X, U = W.knots(name, 4099, seed=17)
    im2 = P.make_im(name, lib, dev); im2.set_cooperative(mode)
    D, DX, DU, st, it = im2.step_grad(torch.tensor(X, device=dev), torch.tensor(U, device=dev))
    out["%s_step_mode%d" % (name, mode)] = dict(D=h(D), G=h(DX, DU), st_it=h(st, it), it_sum=int(it.sum().item()), max_it=int(it.max().item()))
print(json.dumps(out, indent=1))
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
