"""Which stall-exit rule for the thrust-cone projection (csrc/od_rocket_proj_direct.h: STALL_ALPHA, STALL_ALPHA_F32, STALL_ITERS)?
Host build only (tests/host_emu, its per-iteration trace): the candidate controls of one iLQR iteration of config 5 (inputs of
examples/rocket.jl, 64 problems x 11 step sizes x 60 knots = 42 240 projections) are projected with the exit off and the accepted
step length of every iteration is traced; every (threshold, consecutive iterations) rule is then replayed on the traces:
solves abandoned, how many of those would have converged had they run on ("lucky"), and the lockstep cost (sum over knots of the
max over the 64 candidates of a wavefront).   usage: python tools/stall_rule_eval.py [f32|f64] [iLQR iterations before = 3]
(result of the round-4 run: profiles/r4_config5_rollout_lockstep.txt)"""
import ctypes, os, subprocess, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

def candidates(lib, dt, n0, B=64, T=60):
    import ilqr_checks as C
    import optimization_dynamics_amd as od
    dyn, obj, x1, U0 = C.config5_problem(lib, "cpu", B, dtype=dt)
    x1t, Ut = torch.tensor(x1), torch.tensor(U0)
    sol = od.ILQR(dyn, obj, T)
    d = sol.device_solver(B, max_iter=50, obj_tol=0.0)
    d.init(x1t, Ut); d.iterate(n0)
    X, U, J = d.get()
    X, A, Bm, st = sol.linearize(x1t, U)
    K, k, dV, bst = sol.backward(A, Bm, obj.expansion(X, U, None, 0.0), 1e-6)
    Xc, Uc, cst = sol.forward(x1t, X, U, K, k)
    return Uc.reshape(3, -1).contiguous(), sol.alphas.numel()

if len(sys.argv) > 1 and sys.argv[1] == "--trace":          # child: prints the trace of the projections of the saved controls
    from optimization_dynamics_amd import _lib as _L, rocket as rk, models
    lib = _L.Library(os.path.join(ROOT, "tests", "host_emu", "libod_emu.so"))
    dt = torch.float32 if sys.argv[3] == "f32" else torch.float64
    info = rk.RocketInfo(models.rocket, 12.5, 0.05, dtype=dt, device="cpu", lib=lib)
    lib.check(lib.cdll.od_set_projection_stall_exit(info._h, 0))
    Uk = torch.tensor(np.load(sys.argv[2])).to(dt)
    ctypes.c_int.in_dll(lib.cdll, "od_trace_flag").value = 1
    info.project(Uk, grads=False)
    sys.exit(0)

prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
n0 = int(sys.argv[2]) if len(sys.argv) > 2 else 3
from optimization_dynamics_amd import _lib as _L
subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "host_emu"), "-j8"], stdout=subprocess.DEVNULL)
lib = _L.Library(os.path.join(ROOT, "tests", "host_emu", "libod_emu.so"))
Uk, na = candidates(lib, torch.float32 if prec == "f32" else torch.float64, n0)
np.save("/tmp/od_cand_u.npy", Uk.numpy())
env = dict(os.environ, OMP_NUM_THREADS="1")                    # (one thread: the trace lines come in problem order)
txt = subprocess.run([sys.executable, __file__, "--trace", "/tmp/od_cand_u.npy", prec], env=env, capture_output=True, text=True).stdout
S, cur = [], None
for l in txt.split("\n"):
    if not l.startswith("dev it"): continue
    w = l.split()
    if int(w[2]) == 1: cur = []; S.append(cur)
    cur.append(float(w[4]))
T, B = 60, 64
assert len(S) == T * na * B, len(S)
base = np.array([len(s) for s in S]).reshape(T, na, B)
print("%s, after %d iLQR iterations; exit off: lockstep %.1f, without lockstep %.1f; %d solves at max_iter, %d between 20 and 99 iterations"
      % (prec, n0, base.max(2).sum(0).mean(), base.mean(2).sum(0).mean(), int((base >= 100).sum()), int(((base >= 20) & (base < 100)).sum())))
for thr in (1e-9, 1e-6, 1e-5, 1e-4, 1e-3, 1e-2):
    for nst in (2, 4):
        trips, caught, lucky = [], 0, 0
        for s in S:
            c, ex = 0, None
            for i, a in enumerate(s):
                c = c + 1 if a < thr else 0
                if c >= nst: ex = i + 1; break
            if ex is not None and ex < len(s):
                caught += 1; lucky += len(s) < 100; trips.append(ex)
            else:
                trips.append(len(s))
        tr = np.array(trips).reshape(T, na, B)
        print("   step length < %g in %d consecutive iterations: abandons %d, %d of them would have converged; lockstep %.1f" % (thr, nst, caught, lucky, tr.max(2).sum(0).mean()))
