"""ctypes binding of the CPU oracle (oracle/ip_oracle.c) -- TEST INFRASTRUCTURE ONLY.

Imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

MODEL_IDS = {
    "acrobot_impact": 0, "acrobot_nominal": 1, "cartpole_friction": 2, "cartpole_frictionless": 3,
    "planar_push": 4, "rocket_dynamics": 5, "rocket_projection": 6, "hopper": 7,
}


class Opts(C.Structure):
    _fields_ = [("r_tol", C.c_double), ("kappa_tol", C.c_double), ("kappa_grad_tol", C.c_double),
                ("max_iter", C.c_int), ("max_ls", C.c_int),
                ("eps_min", C.c_double), ("kappa_reg", C.c_double), ("gamma_reg", C.c_double),
                ("undercut", C.c_double)]


class Sim(C.Structure):
    _fields_ = [("model_id", C.c_int), ("opts", Opts), ("h", C.c_double),
                ("fric", C.c_double * 4), ("u_max", C.c_double)]


def build(force=False):
    so = os.path.join(_HERE, "libod_oracle.so")
    if force or not os.path.exists(so):
        subprocess.check_call(["make", "-C", _HERE, "libod_oracle.so"] + (["-B"] if force else []),
                              stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(os.environ.get("OD_ORACLE_LIB") or build())       # (OD_ORACLE_LIB: a sanitizer build, tools/asan_tier.sh)
        _LIB.od_oracle_model_name.restype = C.c_char_p
        for i in range(_LIB.od_oracle_num_models_()):          # the generated registry (incl. models added with --add)
            MODEL_IDS[_LIB.od_oracle_model_name(i).decode()] = i
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def dims(model):
    mid = MODEL_IDS[model] if isinstance(model, str) else model
    v = [C.c_int() for _ in range(5)]
    lib().od_oracle_model_dims(mid, *[C.byref(x) for x in v])
    return dict(zip(["nq", "nu", "nz", "nth", "nfric"], [x.value for x in v]))


def make_sim(model, h, **over):
    s = Sim()
    lib().od_oracle_default_sim(MODEL_IDS[model], C.c_double(h), C.byref(s))
    for k, v in over.items():
        if k == "friction":
            for i, f in enumerate(v):
                s.fric[i] = f
        elif k == "u_max":
            s.u_max = v
        else:
            setattr(s.opts, k, v)
    return s


def eval_r(model, z, th, kappa=0.0):
    d = dims(model)
    z = np.ascontiguousarray(z, dtype=np.float64)
    th = np.ascontiguousarray(th, dtype=np.float64)
    r = np.zeros(d["nz"])
    lib().od_oracle_eval_r(MODEL_IDS[model], _p(z), _p(th), C.c_double(kappa), _p(r))
    return r


def eval_rz(model, z, th):
    d = dims(model)
    z = np.ascontiguousarray(z, dtype=np.float64)
    th = np.ascontiguousarray(th, dtype=np.float64)
    out = np.zeros(d["nz"] * d["nz"])
    lib().od_oracle_eval_rz(MODEL_IDS[model], _p(z), _p(th), _p(out))
    return out.reshape(d["nz"], d["nz"], order="F")


def eval_rth(model, z, th):
    d = dims(model)
    z = np.ascontiguousarray(z, dtype=np.float64)
    th = np.ascontiguousarray(th, dtype=np.float64)
    out = np.zeros(d["nz"] * d["nth"])
    lib().od_oracle_eval_rth(MODEL_IDS[model], _p(z), _p(th), _p(out))
    return out.reshape(d["nz"], d["nth"], order="F")


def ip_solve(model, z0, th, kappa_tol=None, diff_sol=False, opts=None):
    """raw interior_point_solve!: returns (status, z, dz or None, iters)"""
    d = dims(model)
    s = make_sim(model, 0.0)
    o = opts if opts is not None else s.opts
    z = np.array(z0, dtype=np.float64)
    th = np.ascontiguousarray(th, dtype=np.float64)
    dz = np.zeros(d["nz"] * d["nth"]) if diff_sol else None
    it = C.c_int()
    kt = o.kappa_tol if kappa_tol is None else kappa_tol
    st = lib().od_oracle_ip_solve(MODEL_IDS[model], C.byref(o), C.c_double(kt), int(diff_sol), _p(z), _p(th), _p(dz), C.byref(it))
    return st, z, (dz.reshape(d["nz"], d["nth"], order="F") if diff_sol else None), it.value


def f(sim, x, u):
    d = dims(sim.model_id)
    x = np.ascontiguousarray(x, dtype=np.float64)
    u = np.ascontiguousarray(u, dtype=np.float64)
    out = np.zeros(2 * d["nq"])
    it = C.c_int()
    st = lib().od_oracle_f(C.byref(sim), _p(x), _p(u), _p(out), C.byref(it))
    return st, out, it.value


def fx(sim, x, u):
    d = dims(sim.model_id)
    n = 2 * d["nq"]
    x = np.ascontiguousarray(x, dtype=np.float64)
    u = np.ascontiguousarray(u, dtype=np.float64)
    out = np.zeros(n * n)
    it = C.c_int()
    st = lib().od_oracle_fx(C.byref(sim), _p(x), _p(u), _p(out), C.byref(it))
    return st, out.reshape(n, n, order="F"), it.value


def fu(sim, x, u):
    d = dims(sim.model_id)
    n = 2 * d["nq"]
    x = np.ascontiguousarray(x, dtype=np.float64)
    u = np.ascontiguousarray(u, dtype=np.float64)
    out = np.zeros(n * d["nu"])
    it = C.c_int()
    st = lib().od_oracle_fu(C.byref(sim), _p(x), _p(u), _p(out), C.byref(it))
    return st, out.reshape(n, d["nu"], order="F"), it.value


def step_full(sim, x, u, kappa_tol, diff_sol=True):
    d = dims(sim.model_id)
    x = np.ascontiguousarray(x, dtype=np.float64)
    u = np.ascontiguousarray(u, dtype=np.float64)
    z = np.zeros(d["nz"])
    dz = np.zeros(d["nz"] * d["nth"])
    it = C.c_int()
    st = lib().od_oracle_step_full(C.byref(sim), C.c_double(kappa_tol), int(diff_sol), _p(x), _p(u), _p(z), _p(dz), C.byref(it))
    return st, z, dz.reshape(d["nz"], d["nth"], order="F"), it.value


def ls_update(fz, feta, eta, theta0=None):
    """src/ls.jl update!: feta (ny x N), eta (nzb x N); returns (theta as ny x nzb, newton iterations)"""
    feta = np.asfortranarray(feta, dtype=np.float64)
    eta = np.asfortranarray(eta, dtype=np.float64)
    fz = np.ascontiguousarray(fz, dtype=np.float64)
    ny, N = feta.shape
    nzb = eta.shape[0]
    th = np.zeros(ny * nzb) if theta0 is None else np.array(theta0, dtype=np.float64).reshape(-1, order="F")
    it = lib().od_oracle_ls_update(N, ny, nzb, _p(fz), _p(feta.reshape(-1, order="F")), _p(eta.reshape(-1, order="F")), _p(th))
    return th.reshape(ny, nzb, order="F"), it


def gradient_bundle(sim, eta, q1, q2, u1, theta0=None):
    d = dims(sim.model_id)
    nq, nu = d["nq"], d["nu"]
    nzb = 2 * nq + nu
    eta = np.asfortranarray(eta, dtype=np.float64)
    N = eta.shape[1]
    th = np.zeros(nq * nzb) if theta0 is None else np.array(theta0, dtype=np.float64).reshape(-1, order="F")
    out = np.zeros(nq * nzb)
    q1 = np.ascontiguousarray(q1, dtype=np.float64)
    q2 = np.ascontiguousarray(q2, dtype=np.float64)
    u1 = np.ascontiguousarray(u1, dtype=np.float64)
    ok = lib().od_oracle_gradient_bundle(C.byref(sim), N, _p(eta.reshape(-1, order="F")), _p(q1), _p(q2), _p(u1), _p(th), _p(out))
    return ok, out.reshape(nq, nzb, order="F")


def rocket(h, x, u, diff_sol=True):
    x = np.ascontiguousarray(x, dtype=np.float64)
    u = np.ascontiguousarray(u, dtype=np.float64)
    y = np.zeros(12)
    dz = np.zeros(12 * 16)
    it = C.c_int()
    st = lib().od_oracle_rocket(C.c_double(h), _p(x), _p(u), int(diff_sol), _p(y), _p(dz), C.byref(it))
    return st, y, dz.reshape(12, 16, order="F"), it.value


def soc_projection(u_max, u, diff_sol=True):
    u = np.ascontiguousarray(u, dtype=np.float64)
    z = np.zeros(10)
    dz = np.zeros(40)
    it = C.c_int()
    st = lib().od_oracle_soc_projection(C.c_double(u_max), _p(u), int(diff_sol), _p(z), _p(dz), C.byref(it))
    return st, z, dz.reshape(10, 4, order="F"), it.value


def rocket_proj(h, u_max, x, u):
    x = np.ascontiguousarray(x, dtype=np.float64)
    u = np.ascontiguousarray(u, dtype=np.float64)
    y = np.zeros(12)
    dx = np.zeros(144)
    du = np.zeros(36)
    ok = lib().od_oracle_rocket_proj(C.c_double(h), C.c_double(u_max), _p(x), _p(u), _p(y), _p(dx), _p(du))
    return ok, y, dx.reshape(12, 12, order="F"), du.reshape(12, 3, order="F")


def step_grad_batch(sim, X, U):
    """X: (2nq, B), U: (nu, B) -> d (2nq,B), dx (2nq,2nq,B), du (2nq,nu,B), nbad"""
    d = dims(sim.model_id)
    n, nu = 2 * d["nq"], d["nu"]
    X = np.asfortranarray(X, dtype=np.float64)
    U = np.asfortranarray(U, dtype=np.float64)
    B = X.shape[1]
    D = np.zeros(n * B)
    DX = np.zeros(n * n * B)
    DU = np.zeros(n * nu * B)
    bad = lib().od_oracle_step_grad_batch(C.byref(sim), B, _p(X.reshape(-1, order="F")), _p(U.reshape(-1, order="F")), _p(D), _p(DX), _p(DU), 0)
    return D.reshape(n, B, order="F"), DX.reshape(n, n, B, order="F"), DU.reshape(n, nu, B, order="F"), bad


def grad_iterates(sim, X, U):
    """the grad simulator's iterates: Zg (nz+1, B) (last row: the clamp differentiate_solution! used) and
    G (nq, 2nq+nu, B) = dq3/d(q1, q2, u1) as this oracle computes it (dense partial-pivot LU in double)"""
    d = dims(sim.model_id)
    nq, nu, nz = d["nq"], d["nu"], d["nz"]
    X = np.asfortranarray(X, dtype=np.float64)
    U = np.asfortranarray(U, dtype=np.float64)
    B = X.shape[1]
    Zg = np.zeros((nz + 1) * B)
    G = np.zeros(nq * (2 * nq + nu) * B)
    bad = lib().od_oracle_grad_iterates(C.byref(sim), B, _p(X.reshape(-1, order="F")), _p(U.reshape(-1, order="F")), _p(Zg), _p(G))
    return Zg.reshape(nz + 1, B, order="F"), G.reshape(nq, 2 * nq + nu, B, order="F"), bad


def arbiter_dq3(sim, X, U, Zg):
    """EXTENDED-PRECISION ARBITER (arbiter.c): dq3/d(q1, q2, u1) = rows q of -rz(z; reg)^{-1} rtheta(z) at the given
    iterates Zg (nz+1, B), solved in IEEE binary128 -> G (nq, 2nq+nu, B), cond (B,) = ||rz||_inf ||rz^-1||_inf"""
    d = dims(sim.model_id)
    nq, nu, nz = d["nq"], d["nu"], d["nz"]
    X = np.asfortranarray(X, dtype=np.float64)
    U = np.asfortranarray(U, dtype=np.float64)
    Zg = np.asfortranarray(Zg, dtype=np.float64)
    B = X.shape[1]
    G = np.zeros(nq * (2 * nq + nu) * B)
    cond = np.zeros(B)
    lib().od_arbiter_dq3_batch(C.byref(sim), B, _p(X.reshape(-1, order="F")), _p(U.reshape(-1, order="F")),
                               _p(Zg.reshape(-1, order="F")), _p(G), _p(cond))
    return G.reshape(nq, 2 * nq + nu, B, order="F"), cond


def arbiter_dz(sim, x, u, zg):
    """EXTENDED-PRECISION ARBITER, one knot, the whole solution: dz/d theta (nz, ntheta) = -rz(z; reg)^{-1} rtheta(z) at
    the iterate zg (nz+1: z and, last, the clamp), theta built from (x, u) as the step does; + cond"""
    d = dims(sim.model_id)
    nq, nu, nz, nth, nf = d["nq"], d["nu"], d["nz"], d["nth"], d["nfric"]
    x = np.asarray(x, dtype=np.float64); u = np.asarray(u, dtype=np.float64); zg = np.ascontiguousarray(zg, dtype=np.float64)
    th = np.zeros(nth)
    v1 = (x[nq:] - x[:nq]) / sim.h
    th[:nq] = x[nq:] - sim.h * v1
    th[nq:2 * nq] = x[nq:]
    th[2 * nq:2 * nq + nu] = u
    th[2 * nq + nu:2 * nq + nu + nf] = [sim.fric[i] for i in range(nf)]
    th[2 * nq + nu + nf] = sim.h
    dz = np.zeros(nz * nth)
    cond = C.c_double()
    lib().od_arbiter_gradient(sim.model_id, _p(zg[:nz].copy()), _p(th), C.c_double(float(zg[nz])), _p(dz), C.byref(cond))
    return dz.reshape(nz, nth, order="F"), cond.value


def rollout(sim, x1, U, grads=True, bufs=None):
    """x1: (2nq,B); U: (nu,T,B) -> X (2nq,T+1,B), A (2nq,2nq,T,B), Bm (2nq,nu,T,B), nbad.
    `bufs` (a dict, filled on first use) lets a caller time the solves without the page faults of fresh output arrays."""
    d = dims(sim.model_id)
    n, nu = 2 * d["nq"], d["nu"]
    x1 = np.asfortranarray(x1, dtype=np.float64)
    U = np.asfortranarray(U, dtype=np.float64)
    T, B = U.shape[1], U.shape[2]
    if bufs is not None and bufs.get("shape") == (n, nu, T, B, grads):
        X, A, Bm = bufs["X"], bufs["A"], bufs["Bm"]
    else:
        X = np.zeros(n * (T + 1) * B)
        A = np.zeros(n * n * T * B) if grads else None
        Bm = np.zeros(n * nu * T * B) if grads else None
        if bufs is not None:
            bufs.update(shape=(n, nu, T, B, grads), X=X, A=A, Bm=Bm)
    bad = lib().od_oracle_rollout(C.byref(sim), B, T, _p(x1.reshape(-1, order="F")), _p(U.reshape(-1, order="F")), _p(X), _p(A), _p(Bm))
    X = X.reshape(n, T + 1, B, order="F")
    if grads:
        return X, A.reshape(n, n, T, B, order="F"), Bm.reshape(n, nu, T, B, order="F"), bad
    return X, None, None, bad


def arbiter_soc_projection(u_max, u, exact_acceptance=True):
    """the thrust-cone projection (src/models/rocket/dynamics.jl:168-186) by the same interior-point loop in binary128
    (oracle/arbiter.c); exact_acceptance: the line search accepts its first trial, as it does in exact arithmetic (the equality
    rows are linear).  -> ok, z (10), iterations, accepted trial per iteration, linearity defect seen"""
    u = np.ascontiguousarray(u, dtype=np.float64)
    z = np.zeros(10)
    it = C.c_int()
    trials = (C.c_int * 128)()
    lerr = C.c_double()
    ok = lib().od_arbiter_soc_projection(C.c_double(u_max), _p(u), int(bool(exact_acceptance)), _p(z), C.byref(it), trials, C.byref(lerr))
    return ok, z, it.value, list(trials[:it.value]), lerr.value


def project_thrust_cone(u, u_max):
    """closed-form Euclidean projection onto {|u_1:2| <= u_3, 0 <= u_3 <= u_max}: minimise (min(a, t) - a)^2 + (t - u_3)^2 over the
    height t (a = |u_1:2|): on t <= a the minimiser is (a + u_3)/2, on t >= a it is u_3, each clipped to its interval"""
    u = np.asarray(u, dtype=np.float64)
    a, t = float(np.hypot(u[0], u[1])), float(u[2])
    cands = []
    t1 = min(max(0.5 * (a + t), 0.0), min(a, u_max))
    cands.append(((t1 - a) ** 2 + (t1 - t) ** 2, t1, t1))
    if u_max >= a:
        t2 = min(max(t, a), u_max)
        cands.append(((t2 - t) ** 2, a, t2))
    _, r, tt = min(cands)
    s = r / a if a > 0 else 0.0
    return np.array([u[0] * s, u[1] * s, tt])


# ---- batched (OpenMP) rocket functions: the parity sweeps (tests/test_gpu_parity_sweep.py, tests/ilqr_checks.py) ----------------
def _pi(a):
    return a.ctypes.data_as(C.POINTER(C.c_int)) if a is not None else None


def rocket_batch(h, X, U, diff_sol=True):
    """f_rocket (+ fx / fu_rocket) on B knots: X (12, B), U (3, B) -> Y (12, B), DZ (12, 16, B) or None, status (B,), iters (B,)"""
    X = np.asfortranarray(X, dtype=np.float64); U = np.asfortranarray(U, dtype=np.float64)
    B = X.shape[1]
    Y = np.zeros(12 * B); DZ = np.zeros(192 * B) if diff_sol else None
    st = np.zeros(B, dtype=np.int32); it = np.zeros(B, dtype=np.int32)
    lib().od_oracle_rocket_batch(C.c_double(h), B, _p(X.reshape(-1, order="F")), _p(U.reshape(-1, order="F")), int(diff_sol), _p(Y), _p(DZ), _pi(st), _pi(it))
    return Y.reshape(12, B, order="F"), (DZ.reshape(12, 16, B, order="F") if diff_sol else None), st, it


def soc_projection_batch(u_max, U, diff_sol=True, exact_boundary=False):
    """soc_projection(_gradient) on B controls: U (3, B) -> Z (10, B), DZ (10, 4, B) or None, status (B,), iters (B,).
    exact_boundary: the loop with its three rounding-decided places completed as exact arithmetic has them (ip_oracle.c::
    od_oracle_set_exact_boundary); the default is the literal double-precision transcription"""
    if exact_boundary:
        lib().od_oracle_set_exact_boundary(1)
        try:
            return soc_projection_batch(u_max, U, diff_sol)
        finally:
            lib().od_oracle_set_exact_boundary(0)
    U = np.asfortranarray(U, dtype=np.float64)
    B = U.shape[1]
    Z = np.zeros(10 * B); DZ = np.zeros(40 * B) if diff_sol else None
    st = np.zeros(B, dtype=np.int32); it = np.zeros(B, dtype=np.int32)
    lib().od_oracle_soc_projection_batch(C.c_double(u_max), B, _p(U.reshape(-1, order="F")), int(diff_sol), _p(Z), _p(DZ), _pi(st), _pi(it))
    return Z.reshape(10, B, order="F"), (DZ.reshape(10, 4, B, order="F") if diff_sol else None), st, it


def rocket_rollout(h, u_max, x1, Ubar, project=True, policy=None):
    """iLQR.rollout over f_rocket(_proj) (examples/rocket.jl:29-41,118): x1 (12, B), Ubar (3, T, B); policy = (alpha, xbar (12, T+1, B),
    K (3, 12, T, B), kff (3, T, B)) for the closed loop u = ubar + alpha kff + K (x - xbar).
    -> X (12, T+1, B), U applied (before projection) (3, T, B), status (T, B) (bit 0 dynamics, bit 4 projection converged)"""
    x1 = np.asfortranarray(x1, dtype=np.float64); Ubar = np.asfortranarray(Ubar, dtype=np.float64)
    T, B = Ubar.shape[1], Ubar.shape[2]
    X = np.zeros(12 * (T + 1) * B); Ua = np.zeros(3 * T * B); st = np.zeros(T * B, dtype=np.int32)
    if policy is None:
        alpha, xb, K, kf = 0.0, None, None, None
    else:
        alpha = float(policy[0])
        xb = np.asfortranarray(policy[1], dtype=np.float64).reshape(-1, order="F")
        K = np.asfortranarray(policy[2], dtype=np.float64).reshape(-1, order="F")          # (3, 12, T, B): element j + 3 i of knot (t, b)
        kf = np.asfortranarray(policy[3], dtype=np.float64).reshape(-1, order="F")
    lib().od_oracle_rocket_rollout(C.c_double(h), C.c_double(u_max), int(bool(project)), B, T, _p(x1.reshape(-1, order="F")), _p(Ubar.reshape(-1, order="F")),
                                   C.c_double(alpha), _p(xb), _p(K), _p(kf), _p(X), _p(Ua), _pi(st))
    return X.reshape(12, T + 1, B, order="F"), Ua.reshape(3, T, B, order="F"), st.reshape(T, B, order="F")


def arbiter_gradient_batch(model, Z, TH, reg=None):
    """EXTENDED-PRECISION ARBITER, batched: dz/dtheta = -rz(z; reg)^{-1} rtheta(z) in binary128 at the given iterates.
    Z (nz, B), TH (nth, B), reg (B,) or None -> DZ (nz, nth, B), cond (B,)"""
    d = dims(model)
    nz, nth = d["nz"], d["nth"]
    Z = np.asfortranarray(Z, dtype=np.float64); TH = np.asfortranarray(TH, dtype=np.float64)
    B = Z.shape[1]
    DZ = np.zeros(nz * nth * B); cond = np.zeros(B)
    r = None if reg is None else np.ascontiguousarray(reg, dtype=np.float64)
    lib().od_arbiter_gradient_batch(MODEL_IDS[model] if isinstance(model, str) else model, B, _p(Z.reshape(-1, order="F")), _p(TH.reshape(-1, order="F")), _p(r), _p(DZ), _p(cond))
    return DZ.reshape(nz, nth, B, order="F"), cond


def arbiter_soc_projection_batch(u_max, U, exact_acceptance=True):
    """the thrust-cone projection in binary128 on B controls (arbiter_soc_projection): U (3, B) -> Z (10, B), ok (B,), iters (B,)"""
    U = np.asfortranarray(U, dtype=np.float64)
    B = U.shape[1]
    Z = np.zeros(10 * B); ok = np.zeros(B, dtype=np.int32); it = np.zeros(B, dtype=np.int32)
    lib().od_arbiter_soc_projection_batch(C.c_double(u_max), B, _p(U.reshape(-1, order="F")), int(bool(exact_acceptance)), _p(Z), _pi(ok), _pi(it))
    return Z.reshape(10, B, order="F"), ok, it


def project_thrust_cone_batch(U, u_max):
    return np.stack([project_thrust_cone(U[:, b], u_max) for b in range(U.shape[1])], axis=1)


def violations_batch(model, Z, TH):
    """the stopping test of the interior-point loop at given points: -> r_vio (B,), k_vio (B,) (max |equality rows|, max |bilinear rows| at kappa = 0)"""
    d = dims(model)
    Z = np.asfortranarray(Z, dtype=np.float64); TH = np.asfortranarray(TH, dtype=np.float64)
    B = Z.shape[1]
    rv = np.zeros(B); kv = np.zeros(B)
    lib().od_oracle_violations_batch(MODEL_IDS[model] if isinstance(model, str) else model, B, _p(Z.reshape(-1, order="F")), _p(TH.reshape(-1, order="F")), _p(rv), _p(kv))
    return rv, kv
