"""Where the forward pass of config 5 (inputs of examples/rocket.jl) spends its time: interior-point iteration counts of the
thrust-cone projection on every candidate knot of an iLQR iteration, per step size, and what lockstep (64 candidates per wavefront:
sum over knots of the max over the lanes) makes of them.   usage: python tools/diag_config5_rollout.py [iterations before] [out.txt]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ilqr_checks as C
import optimization_dynamics_amd as od
from optimization_dynamics_amd import interior_point as IP

dev = torch.device("cuda", 0)
from optimization_dynamics_amd import _lib as _L
lib = _L.Library(os.environ["OD_LIB"]) if os.environ.get("OD_LIB") else od.default_library()   # variants/libod_itstat.so: counts of the rollout kernel itself
B, T = 4096, 60
dt = torch.float32 if "f32" in sys.argv else torch.float64
n0 = int(sys.argv[1]) if len(sys.argv) > 1 else 3
out = sys.stdout
dyn, obj, x1, U0 = C.config5_problem(lib, dev, B, dtype=dt)
x1t, Ut = torch.tensor(x1, device=dev), torch.tensor(U0, device=dev)
sol = od.ILQR(dyn, obj, T)
d = sol.device_solver(B, max_iter=50, obj_tol=0.0)
d.init(x1t, Ut); d.iterate(n0)
X, U, J = d.get()
X, A, Bm, st = sol.linearize(x1t, U)
quad = obj.expansion(X, U, None, 0.0)
K, k, dV, bst = sol.backward(A, Bm, quad, 1e-6)
Xc, Uc, cst = sol.forward(x1t, X, U, K, k)
na = sol.alphas.numel()
P = na * B
cst = cst.reshape(T, na, B)
if (cst >> 8).any():            # the measurement build: the kernel's own counts (with its stall exit)
    it = ((cst >> 8) & 0xFF).double()
    conv = ((cst >> 4) & 1) == 1
    itd = ((cst >> 16) & 0xFF).double()
    print("counts of k_rocket_rollout itself; dynamics solve: mean %.2f max %d iterations" % (itd.mean().item(), int(itd.max().item())), file=out)
else:                           # the generic solver on the candidates' controls (no stall exit)
    ipp = IP.InteriorPoint("rocket_projection", device=dev, lib=lib)
    Uk = Uc.double().reshape(3, T * P).contiguous()
    nk = Uk.shape[1]
    z0 = torch.tensor([0.1, 0.1, 1.1, 0.1, 0.1, 0.1, 0.0, 0.1, 0.1, 1.1], device=dev)[:, None].repeat(1, nk)
    zp, _, stp, itp = ipp.solve(z0, torch.cat([Uk, torch.full((1, nk), 12.5, dtype=torch.float64, device=dev)]), diff_sol=False)
    it = itp[0].reshape(T, na, B).double()
    conv = ((stp.reshape(-1) & 1) == 1).reshape(T, na, B)
un = Uc.double().reshape(3, T * P).norm(dim=0).reshape(T, na, B)
print("after %d iterations; cost mean %.1f" % (n0, J.mean().item()), file=out)
print("alpha    mean it   max it   non-conv   |u| mean    |u| max   lockstep sum_t max_64   sum_t mean", file=out)
for a in range(na):
    ia = it[:, a]                                   # (T, B)
    w = ia.reshape(T, B // 64, 64).max(dim=2).values.sum(dim=0)          # per wavefront
    print("%.4f  %8.2f %8d %10d %10.2f %10.1f      mean %7.1f max %7.1f   %8.1f" % (sol.alphas[a].item(), ia.mean().item(), int(ia.max().item()), int((~conv[:, a]).sum().item()),
          un[:, a].mean().item(), un[:, a].max().item(), w.mean().item(), w.max().item(), ia.mean(dim=1).sum().item()), file=out)
h = torch.bincount(it.reshape(-1).long(), minlength=101).cpu().numpy()
print("histogram of iteration counts (count : knots):", {i: int(c) for i, c in enumerate(h) if c}, file=out)

# ---- what other assignments of candidates to wavefronts would need (the launch ends with its slowest wavefront when every
# wavefront has a SIMD of its own: <= 1024 wavefronts) ----
def lockstep(groups):            # groups: (T, nw, lanes) trips, zero-padded
    w = groups.max(dim=2).values.sum(dim=0)
    return w.shape[0], w.mean().item(), w.max().item()
def padded(x, lanes):            # x: (T, N) -> (T, ceil(N / lanes), lanes)
    N = x.shape[1]; nw = (N + lanes - 1) // lanes
    y = torch.zeros(x.shape[0], nw * lanes, dtype=x.dtype, device=x.device); y[:, :N] = x
    return y.reshape(x.shape[0], nw, lanes)
tot = it + (itd if (cst >> 8).any() else 0) * 0.5            # a dynamics iteration is roughly half a projection iteration
print("assignment of the %d candidates to wavefronts: wavefronts, mean / max of sum_t max_lanes (projection trips; with the dynamics trips at half weight)" % P, file=out)
for name, lanes in (("64 consecutive problems of one step size (shipped)", 64), ("48 of one step size", 48), ("44 of one step size", 44), ("32 of one step size", 32)):
    g = padded(it.reshape(T, P), lanes); g2 = padded(tot.reshape(T, P), lanes)
    a1, a2 = lockstep(g), lockstep(g2)
    print("  %-60s %5d  %7.1f %7.1f   (%7.1f %7.1f)" % (name, a1[0], a1[1], a1[2], a2[1], a2[2]), file=out)
for G in (5, 4, 3, 2, 1):
    def by_problem(x):           # (T, na, B) -> (T, B / G, G * na): all step sizes of G problems share a wavefront
        y = x.permute(0, 2, 1).reshape(T, B * na)
        return padded(y, G * na)
    a1, a2 = lockstep(by_problem(it)), lockstep(by_problem(tot))
    print("  %-60s %5d  %7.1f %7.1f   (%7.1f %7.1f)" % ("all %d step sizes of %d problems" % (na, G), a1[0], a1[1], a1[2], a2[1], a2[2]), file=out)
