"""Host-side mirror of the reference's `ImplicitDynamics` and its iLQR callbacks
(src/dynamics.jl): same names, argument meaning and in-place semantics, plus batched variants that
hand whole batches of knots / rollouts to the HIP kernels through the C ABI.

    im_dyn = ImplicitDynamics(acrobot_impact, h, r_tol=1e-8, kappa_eval_tol=1e-4, kappa_grad_tol=1e-3)
    f(d, im_dyn, x, u, w);  fx(dx, im_dyn, x, u, w);  fu(du, im_dyn, x, u, w)       # reference signatures
    D, DX, DU, status, iters = im_dyn.step_grad(X, U)                               # batched, device tensors
    X, A, B, status, iters = im_dyn.rollout(x1, U)

The residual / Jacobian functions the reference passes to the constructor (`eval(r_func)` ...) are
compiled into the library per model, so the constructor takes the model only.
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib
from .models import Model


def _ptr(t):
    return 0 if t is None else t.data_ptr()


class ImplicitDynamics:
    """src/dynamics.jl:1-14,51-79.  One native handle plays both `eval_sim` (kappa_eval_tol, no
    differentiation) and `grad_sim` (kappa_grad_tol, differentiation)."""

    def __init__(self, model: Model, h: float, *, T=1, r_tol=1.0e-8, kappa_eval_tol=1.0e-6,
                 kappa_grad_tol=1.0e-6, no_impact=False, no_friction=False, n=None, m=None, d=None,
                 nc=None, nb=None, info=None, device="cuda", lib=None, options=None):
        self.model = model
        self.h = float(h)
        self.n = 2 * model.nq if n is None else n          # src/dynamics.jl:54
        self.m = model.nu if m is None else m
        self.d = model.nw if d is None else d
        nc = model.nc if nc is None else nc
        nb = model.nc if nb is None else nb
        self.nc = 0 if no_impact else nc                   # :58-59 (bookkeeping sizes only)
        self.nb = 0 if no_friction else nb
        self.info = info
        self.idx_q1 = list(range(model.nq))                # :72-74 (0-based)
        self.idx_q2 = list(range(model.nq, 2 * model.nq))
        self.idx_u1 = list(range(model.nu))
        self.device = torch.device(device)
        self.lib = lib if lib is not None else _lib.default_library()
        self.layout = _lib.LAYOUT_BATCH_MINOR
        o = self.lib.default_options(model.name)           # preset of get_simulator :25-33
        o.r_tol, o.kappa_eval_tol, o.kappa_grad_tol = r_tol, kappa_eval_tol, kappa_grad_tol
        o.undercut, o.gamma_reg, o.max_ls, o.eps_min = math.inf, 0.1, 25, 0.25
        if options:
            for k, v in options.items():
                setattr(o, k, v)
        self.options = o
        hd = C.c_void_p()
        with _lib.on_device(self.device):      # (the handle lives on the device that is current in od_create; every later call runs there)
            self.lib.check(self.lib.cdll.od_create(self.lib.model_id(model.name), _lib.OD_F64, C.byref(o), self.h, C.byref(hd)))
        self._h = hd
        self._fric_sent = None
        self._sync_friction()

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self.lib.cdll.od_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # -- plumbing ------------------------------------------------------------------------------
    def _sync_friction(self):
        fr = np.ascontiguousarray(self.model.friction, dtype=np.float64)
        if fr.size and (self._fric_sent is None or not np.array_equal(fr, self._fric_sent)):
            self.lib.check(self.lib.cdll.od_set_friction(self._h, fr.ctypes.data_as(C.POINTER(C.c_double)), fr.size))
            self._fric_sent = fr.copy()

    def set_stream(self, stream_ptr):
        self.lib.check(self.lib.cdll.od_set_stream(self._h, stream_ptr))

    def _use_current_stream(self):
        if self.device.type == "cuda":
            self.set_stream(torch.cuda.current_stream(self.device).cuda_stream)

    def set_timestep(self, h):
        """sim.h of both simulators"""
        self.h = float(h)
        self.lib.check(self.lib.cdll.od_set_timestep(self._h, self.h))

    def set_options(self, **kw):
        """change solver options of the live handle (r_tol, kappa_eval_tol, kappa_grad_tol, max_iter, max_ls, eps_min,
        kappa_reg, gamma_reg, undercut); returns the options now in force"""
        for k, v in kw.items():
            setattr(self.options, k, v)
        self.lib.check(self.lib.cdll.od_set_options(self._h, C.byref(self.options)))
        return self.get_options()

    def get_options(self):
        o = _lib.Options()
        self.lib.check(self.lib.cdll.od_get_options(self._h, C.byref(o)))
        return o

    def set_launch_config(self, ppw=0, waves_per_block=0):
        """launch tuning (0 / -1 = automatic), see od_set_launch_config"""
        self.lib.check(self.lib.cdll.od_set_launch_config(self._h, int(ppw), int(waves_per_block)))

    def set_cooperative(self, mode=0):
        """cooperative solve pass (16 lanes per problem): 0 automatic, 1 never, 2 always where the model has it"""
        self.lib.check(self.lib.cdll.od_set_cooperative(self._h, int(mode)))

    def grad_iterates(self, K):
        """(nz+1, K): the iterates (and, last row, orthant clamps) at which the last step_grad / rollout call on this
        handle took its K implicit gradients -- diagnostics: a checker can recompute dz = -rz^{-1} rtheta at exactly those points"""
        self._use_current_stream()
        nz = self.lib.model_dims(self.model.name)["nz"]
        out = self._new(nz + 1, K)
        self.lib.check(self.lib.cdll.od_get_grad_iterates(self._h, K, _ptr(out)))
        return out

    def synchronize(self):
        self.lib.check(self.lib.cdll.od_synchronize(self._h))

    def _new(self, *shape, dtype=torch.float64):
        return torch.empty(*shape, dtype=dtype, device=self.device)

    def _prep(self, t):
        return t.to(device=self.device, dtype=torch.float64).contiguous()

    # -- batched entry points (BATCH_MINOR tensors: element axis first, batch axis last) --------
    def step(self, X, U):
        """f for B knots.  X: (2nq, B), U: (nu, B) -> D (2nq, B), status (B,), iters (2, B)."""
        self._sync_friction(); self._use_current_stream()
        X, U = self._prep(X), self._prep(U)
        B = X.shape[-1]
        D = self._new(2 * self.model.nq, B)
        st = self._new(B, dtype=torch.int32); it = self._new(2, B, dtype=torch.int32)
        self.lib.check(self.lib.cdll.od_step(self._h, B, _ptr(X), _ptr(U), _ptr(D), _ptr(st), _ptr(it)))
        return D, st, it

    def step_grad(self, X, U):
        """f + fx + fu for B knots -> D (2nq,B), DX (2nq,2nq,B), DU (2nq,nu,B), status, iters.
        DX[:, :, b] / DU[:, :, b] are the reference's dx / du matrices (row, col, batch)."""
        self._sync_friction(); self._use_current_stream()
        X, U = self._prep(X), self._prep(U)
        B = X.shape[-1]
        n, nu = 2 * self.model.nq, self.model.nu
        D = self._new(n, B)
        DXf = self._new(n * n, B); DUf = self._new(n * nu, B)
        st = self._new(B, dtype=torch.int32); it = self._new(2, B, dtype=torch.int32)
        self.lib.check(self.lib.cdll.od_step_grad(self._h, B, _ptr(X), _ptr(U), _ptr(D), _ptr(DXf), _ptr(DUf), _ptr(st), _ptr(it)))
        # column-major (i + n*j) flattening -> (col, row, B) -> (row, col, B)
        DX = DXf.view(n, n, B).transpose(0, 1)
        DU = DUf.view(nu, n, B).transpose(0, 1)
        return D, DX, DU, st, it

    def step_grad_compact(self, X, U):
        """-> q3 (nq,B), dq3 (nq, 2nq+nu, B) = [dq3/dq1 dq3/dq2 dq3/du1], status, iters"""
        self._sync_friction(); self._use_current_stream()
        X, U = self._prep(X), self._prep(U)
        B = X.shape[-1]
        nq, nzb = self.model.nq, 2 * self.model.nq + self.model.nu
        Q3 = self._new(nq, B); G = self._new(nq * nzb, B)
        st = self._new(B, dtype=torch.int32); it = self._new(2, B, dtype=torch.int32)
        self.lib.check(self.lib.cdll.od_step_grad_compact(self._h, B, _ptr(X), _ptr(U), _ptr(Q3), _ptr(G), _ptr(st), _ptr(it)))
        return Q3, G.view(nzb, nq, B).transpose(0, 1), st, it

    def step_full(self, X, U, grads=True):
        """the whole solution: Z (nz, B) at kappa_eval, DZ (nz, 2nq+nu, B) = dz/d(q1, q2, u1) at kappa_grad, status, iters;
        rows via `self.indices` = {"q": [...], "gamma": [...], "b": [...]} (RoboDojo's sim.traj.gamma / b, sim.grad.*)"""
        self._sync_friction(); self._use_current_stream()
        X, U = self._prep(X), self._prep(U)
        B = X.shape[-1]
        nz = self.lib.model_dims(self.model.name)["nz"]
        ngc = 2 * self.model.nq + self.model.nu
        Z = self._new(nz, B)
        DZf = self._new(nz * ngc, B) if grads else None
        st = self._new(B, dtype=torch.int32); it = self._new(2, B, dtype=torch.int32)
        self.lib.check(self.lib.cdll.od_step_full(self._h, B, _ptr(X), _ptr(U), _ptr(Z), _ptr(DZf), _ptr(st), _ptr(it)))
        DZ = DZf.view(ngc, nz, B).transpose(0, 1) if grads else None
        return Z, DZ, st, it

    @property
    def indices(self):
        return self.lib.model_indices(self.model.name)

    def contact_forces(self, X, U, grads=True):
        """gamma (nc, B), b (nb, B) and, with grads, d gamma / d(q1, q2, u1) (nc, 2nq+nu, B), d b / d(...) (nb, ..., B)"""
        Z, DZ, st, it = self.step_full(X, U, grads)
        ix = self.indices
        g, b = ix["gamma"], ix["b"]
        return Z[g], Z[b], (DZ[g] if grads else None), (DZ[b] if grads else None), st

    def rollout(self, x1, U, grads=True, out=None):
        """iLQR.rollout + derivative sweep.  x1: (2nq, B); U: (nu, T, B)
        -> X (2nq, T+1, B), A (2nq, 2nq, T, B), Bm (2nq, nu, T, B), status (T, B), iters (2, T, B).
        `out` may carry preallocated flat buffers from a previous call (dict returned as 6th value)."""
        self._sync_friction(); self._use_current_stream()
        x1, U = self._prep(x1), self._prep(U)
        nu_, T, B = U.shape
        n, nu = 2 * self.model.nq, self.model.nu
        if out is None:
            out = dict(X=self._new(n, T + 1, B),
                       A=self._new(n * n, T, B) if grads else None,
                       Bm=self._new(n * nu, T, B) if grads else None,
                       st=self._new(T, B, dtype=torch.int32), it=self._new(2, T, B, dtype=torch.int32))
        self.lib.check(self.lib.cdll.od_rollout(self._h, B, T, _ptr(x1), _ptr(U), _ptr(out["X"]), _ptr(out["A"]),
                                                _ptr(out["Bm"]), _ptr(out["st"]), _ptr(out["it"])))
        A = out["A"].view(n, n, T, B).transpose(0, 1) if grads else None
        Bm = out["Bm"].view(nu, n, T, B).transpose(0, 1) if grads else None
        return out["X"], A, Bm, out["st"], out["it"], out

    def rollout_compact(self, x1, U, out=None):
        """rollout with the linearisation in compact form: -> X (2nq, T+1, B), G (nq, 2nq+nu, T, B) = dq3/d(q1, q2, u1)
        per knot (the non-constant block of fx / fu), status (T, B), iters (2, T, B), out"""
        self._sync_friction(); self._use_current_stream()
        x1, U = self._prep(x1), self._prep(U)
        nu_, T, B = U.shape
        nq, n, nu = self.model.nq, 2 * self.model.nq, self.model.nu
        if out is None:
            out = dict(X=self._new(n, T + 1, B), G=self._new(nq * (n + nu), T, B),
                       st=self._new(T, B, dtype=torch.int32), it=self._new(2, T, B, dtype=torch.int32))
        self.lib.check(self.lib.cdll.od_rollout_compact(self._h, B, T, _ptr(x1), _ptr(U), _ptr(out["X"]), _ptr(out["G"]),
                                                        _ptr(out["st"]), _ptr(out["it"])))
        return out["X"], out["G"].view(n + nu, nq, T, B).transpose(0, 1), out["st"], out["it"], out

    # -- scalar host path (reference signatures) ------------------------------------------------
    def _host(self, fn, x, u, out):
        self._sync_friction()
        x = np.ascontiguousarray(x, dtype=np.float64)
        u = np.ascontiguousarray(u, dtype=np.float64)
        if self.device.type == "cuda":
            self.set_stream(None)
        buf = np.zeros(out.size, dtype=np.float64)
        self.lib.check(fn(self._h, x.ctypes.data, u.ctypes.data, buf.ctypes.data))
        return buf


def f(d, model: ImplicitDynamics, x, u, w=None):
    """src/dynamics.jl:81-94: d <- [q2; q3]; returns d."""
    buf = model._host(model.lib.cdll.od_f_host, x, u, d)
    d[...] = buf.reshape(d.shape)
    return d


def fx(dx, model: ImplicitDynamics, x, u, w=None):
    """src/dynamics.jl:96-114: writes the identity block and dq3/dq1, dq3/dq2 into dx (2nq x 2nq).
    Like the reference, only those blocks are touched (the caller pre-zeroes dx)."""
    nq = model.model.nq
    n = 2 * nq
    buf = model._host(model.lib.cdll.od_fx_host, x, u, np.empty(n * n)).reshape(n, n, order="F")
    for i in range(nq):
        dx[model.idx_q1[i], model.idx_q2[i]] = 1.0
    dx[np.ix_(model.idx_q2, model.idx_q1)] = buf[nq:, :nq]
    dx[np.ix_(model.idx_q2, model.idx_q2)] = buf[nq:, nq:]
    return dx


def fu(du, model: ImplicitDynamics, x, u, w=None):
    """src/dynamics.jl:116-128: du[idx_q2, :] <- dq3/du1."""
    nq, nu = model.model.nq, model.model.nu
    n = 2 * nq
    buf = model._host(model.lib.cdll.od_fu_host, x, u, np.empty(n * nu)).reshape(n, nu, order="F")
    du[model.idx_q2, :] = buf[nq:, :]
    return du


def ffxfu(d, dx, du, model: ImplicitDynamics, x, u, w=None):
    """f, fx and fu of one knot from ONE solve (od_ffxfu_host; the three reference callbacks solve three times):
    writes d, the three blocks of dx and the lower block of du like f / fx / fu above"""
    model._sync_friction()
    nq, nu = model.model.nq, model.model.nu
    n = 2 * nq
    x = np.ascontiguousarray(x, dtype=np.float64); u = np.ascontiguousarray(u, dtype=np.float64)
    if model.device.type == "cuda":
        model.set_stream(None)
    bd = np.zeros(n); bx = np.zeros(n * n); bu = np.zeros(n * nu)
    model.lib.check(model.lib.cdll.od_ffxfu_host(model._h, x.ctypes.data, u.ctypes.data, bd.ctypes.data, bx.ctypes.data, bu.ctypes.data))
    d[...] = bd.reshape(d.shape)
    bx = bx.reshape(n, n, order="F"); bu = bu.reshape(n, nu, order="F")
    for i in range(nq):
        dx[model.idx_q1[i], model.idx_q2[i]] = 1.0
    dx[np.ix_(model.idx_q2, model.idx_q1)] = bx[nq:, :nq]
    dx[np.ix_(model.idx_q2, model.idx_q2)] = bx[nq:, nq:]
    du[model.idx_q2, :] = bu[nq:, :]
    return d, dx, du


def state_to_configuration(x):
    """src/dynamics.jl:131-145: [x_1 .. x_H] (each [q_t; q_{t+1}]) -> [q_1, q_2, ..., q_{H+1}]."""
    q = []
    nq = len(x[0]) // 2
    for t, xt in enumerate(x):
        if t == 0:
            q.append(np.array(xt[:nq]))
        q.append(np.array(xt[nq:2 * nq]))
    return q
