"""Host-side mirror of src/models/rocket/dynamics.jl: `RocketInfo` and the callbacks
f_rocket / fx_rocket / fu_rocket and their thrust-cone-projected variants.

One HIP lane per knot runs the SOCP projection (10-variable interior-point solve), the implicit-
midpoint dynamics solve (12-variable Newton), both implicit gradients and the 12x3 * 3x3 chain
product (mul!, dynamics.jl:264-267).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .dynamics import _ptr
from .models import rocket as rocket_model


class RocketInfo:
    """RocketInfo(rocket, u_max, h, r, rz, rtheta, r_proj, rz_proj, rtheta_proj)
    (dynamics.jl:13-99); residual functions are compiled into the library."""

    def __init__(self, rocket=rocket_model, u_max=12.5, h=0.05, *, dtype=torch.float64, device="cuda",
                 lib=None, options=None):
        self.model = rocket
        self.u_max = float(u_max)
        self.h = float(h)
        self.dtype = dtype
        self.device = torch.device(device)
        self.lib = lib if lib is not None else _lib.default_library()
        o = self.lib.default_options("rocket_dynamics")      # dynamics.jl:21-27
        if dtype == torch.float32:
            # r_tol = 1e-8 is below fp32 resolution (eps ~ 6e-8 * |y|): rescaled, see DESIGN.md
            o.r_tol = 1.0e-4
        if options:
            for k, v in options.items():
                setattr(o, k, v)
        self.options = o
        hd = C.c_void_p()
        od_dtype = _lib.OD_F64 if dtype == torch.float64 else _lib.OD_F32
        with _lib.on_device(self.device):
            self.lib.check(self.lib.cdll.od_create(self.lib.model_id("rocket_dynamics"), od_dtype, C.byref(o), self.h, C.byref(hd)))
        self._h = hd
        self.projection_stall_exit = False          # od_create's default (the reference runs every iteration)
        self.lib.check(self.lib.cdll.od_set_u_max(self._h, self.u_max))

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self.lib.cdll.od_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def _use_current_stream(self):
        if self.device.type == "cuda":
            self.lib.check(self.lib.cdll.od_set_stream(self._h, torch.cuda.current_stream(self.device).cuda_stream))

    def set_projection_stall_exit(self, on):
        """od_set_projection_stall_exit: off (the default) runs every iteration of a stalled thrust-cone projection like the reference"""
        self.lib.check(self.lib.cdll.od_set_projection_stall_exit(self._h, 1 if on else 0))
        self.projection_stall_exit = bool(on)

    def project_full(self, U, grads=True):
        """od_soc_project_full: U (3, B) -> z (10, B) whole solution of the projection's solve, duproj (3, 3, B) or None, status, iterations"""
        self._use_current_stream()
        U = U.to(device=self.device, dtype=self.dtype).contiguous()
        B = U.shape[1]
        Z = torch.empty(10, B, dtype=self.dtype, device=self.device)
        DP = torch.zeros(9, B, dtype=self.dtype, device=self.device) if grads else None
        st = torch.zeros(B, dtype=torch.int32, device=self.device)
        it = torch.zeros(B, dtype=torch.int32, device=self.device)
        self.lib.check(self.lib.cdll.od_soc_project_full(self._h, B, _ptr(U), _ptr(Z), _ptr(DP) if grads else None, _ptr(st), _ptr(it)))
        if grads:
            DP = DP.view(3, 3, B).transpose(0, 1)
        return Z, DP, st, it

    def solve(self, X, U, project=False, grads=True):
        """batched: X (12, B), U (3, B) -> Y (12,B), DX (12,12,B), DU (12,3,B), Uproj (3,B), status (B,)"""
        self._use_current_stream()
        X = X.to(device=self.device, dtype=self.dtype).contiguous()
        U = U.to(device=self.device, dtype=self.dtype).contiguous()
        B = X.shape[-1]
        mk = lambda *s: torch.empty(*s, dtype=self.dtype, device=self.device)
        Y = mk(12, B)
        DX = mk(144, B) if grads else None
        DU = mk(36, B) if grads else None
        UP = mk(3, B) if project else None
        st = torch.empty(B, dtype=torch.int32, device=self.device)
        self.lib.check(self.lib.cdll.od_rocket(self._h, B, 1 if project else 0, _ptr(X), _ptr(U), _ptr(Y), _ptr(DX), _ptr(DU), _ptr(UP), _ptr(st)))
        if grads:
            DX = DX.view(12, 12, B).transpose(0, 1)
            DU = DU.view(3, 12, B).transpose(0, 1)
        return Y, DX, DU, UP, st

    def project(self, U, grads=True):
        """batched soc_projection(_gradient): U (3, B) -> (uproj (3, B), duproj (3, 3, B) or None, status (B,))"""
        self._use_current_stream()
        U = U.to(device=self.device, dtype=self.dtype).contiguous()
        B = U.shape[1]
        UP = torch.empty(3, B, dtype=self.dtype, device=self.device)
        DP = torch.zeros(9, B, dtype=self.dtype, device=self.device) if grads else None
        st = torch.zeros(B, dtype=torch.int32, device=self.device)
        self.lib.check(self.lib.cdll.od_soc_project(self._h, B, _ptr(U), _ptr(UP), _ptr(DP) if grads else None, _ptr(st)))
        if grads:
            DP = DP.view(3, 3, B).transpose(0, 1)       # col-major 3 x 3 per problem: element (i, c) at row i + 3c
        return UP, DP, st


def _rocket_rollout(info, x1, U, project, policy=None):
    """time recursion on the device.  x1 (12, B), U = ubar (3, T, B).
    policy = (alphas (na,), xbar (12, T+1, B), K (36, T, B), k (3, T, B)) for the closed-loop forward pass.
    -> X (12, T+1, P), Uapplied (3, T, P), status (T, P) with P = B or na*B"""
    info._use_current_stream()
    dt, dev = info.dtype, info.device
    x1 = x1.to(device=dev, dtype=dt).contiguous()
    U = U.to(device=dev, dtype=dt).contiguous()
    T, B = U.shape[1], U.shape[2]
    if policy is None:
        na, P = 0, B
    else:
        alphas, xbar, K, k = [t.to(device=dev, dtype=dt).contiguous() for t in policy]
        na, P = alphas.numel(), alphas.numel() * B
    X = torch.empty(12, T + 1, P, dtype=dt, device=dev)
    Ua = torch.empty(3, T, P, dtype=dt, device=dev)
    st = torch.empty(T, P, dtype=torch.int32, device=dev)
    if policy is None:
        rc = info.lib.cdll.od_rocket_rollout(info._h, B, T, 0, 0, 1 if project else 0, _ptr(x1), 0, _ptr(U), 0, 0, _ptr(X), _ptr(Ua), _ptr(st))
    else:
        rc = info.lib.cdll.od_rocket_rollout(info._h, B, T, na, _ptr(alphas), 1 if project else 0, _ptr(x1), _ptr(xbar), _ptr(U),
                                             _ptr(K), _ptr(k), _ptr(X), _ptr(Ua), _ptr(st))
    info.lib.check(rc)
    return X, Ua, st


class RocketDynamics:
    """iLQR view of a RocketInfo (f_rocket / f_rocket_proj as the dynamics of examples/rocket.jl:29-41):
    the interface optimization_dynamics_amd.ilqr.ILQR expects (n, m, rollout, rollout_policy)."""

    def __init__(self, info, project=True):
        self.info, self.project = info, project
        self.n, self.m = 12, 3
        self.device, self.lib = info.device, info.lib
        self._use_current_stream = info._use_current_stream

    @property
    def _h(self):
        return self.info._h          # (None once the RocketInfo is closed)

    def _prep(self, t):
        return t.to(device=self.device, dtype=torch.float64).contiguous()

    def rollout(self, x1, U, grads=True):
        X, Ua, st = _rocket_rollout(self.info, x1, U, self.project)
        A = Bm = None
        if grads:
            T, B = U.shape[1], U.shape[2]
            Y, DX, DU, UP, s2 = self.info.solve(X[:, :-1].reshape(12, T * B), U.reshape(3, T * B), project=self.project, grads=True)
            A = DX.reshape(12, 12, T, B).double()
            Bm = DU.reshape(12, 3, T, B).double()
        return X.double(), A, Bm, st, None, None

    def linearize_knots(self, Xk, Uk, with_status=False):
        """fx / fu (projection chain rule included) on independent knots: Xk (12, K), Uk (3, K) -> (12, 12, K), (12, 3, K)"""
        _, DX, DU, _, st = self.info.solve(Xk, Uk, project=self.project, grads=True)
        return (DX.double(), DU.double(), st) if with_status else (DX.double(), DU.double())

    def rollout_policy(self, x1, X, U, K, k, alphas):
        Xc, Uc, st = _rocket_rollout(self.info, x1, U, self.project, policy=(alphas, X, K, k))
        return Xc, Uc, st            # (the handle's element type: od_quad_cost reads it as it is, ILQR converts what it keeps)


def _scalar(info, x, u, project, grads):
    X = torch.tensor(np.asarray(x, dtype=np.float64)).reshape(12, 1)
    U = torch.tensor(np.asarray(u, dtype=np.float64)).reshape(3, 1)
    Y, DX, DU, UP, st = info.solve(X, U, project=project, grads=grads)
    return (Y[:, 0].double().cpu().numpy(),
            None if DX is None else DX[:, :, 0].double().cpu().numpy(),
            None if DU is None else DU[:, :, 0].double().cpu().numpy())


def f_rocket(d, info: RocketInfo, x, u, w=None):
    """dynamics.jl:101-114"""
    d[...] = _scalar(info, x, u, False, False)[0]
    return d


def fx_rocket(dx, info: RocketInfo, x, u, w=None):
    """dynamics.jl:134-148"""
    dx[...] = _scalar(info, x, u, False, True)[1]
    return dx


def fu_rocket(du, info: RocketInfo, x, u, w=None):
    """dynamics.jl:150-164"""
    du[...] = _scalar(info, x, u, False, True)[2]
    return du


def f_rocket_proj(d, info: RocketInfo, x, u, w=None):
    """dynamics.jl:215-228"""
    d[...] = _scalar(info, x, u, True, False)[0]
    return d


def fx_rocket_proj(dx, info: RocketInfo, x, u, w=None):
    """dynamics.jl:239-252"""
    dx[...] = _scalar(info, x, u, True, True)[1]
    return dx


def fu_rocket_proj(du, info: RocketInfo, x, u, w=None):
    """dynamics.jl:254-268"""
    du[...] = _scalar(info, x, u, True, True)[2]
    return du


def soc_projection(x, info: RocketInfo):
    """dynamics.jl:168-186: Euclidean projection of x onto {|u_1:2| <= u_3, 0 <= u_3 <= u_max}
    (interior-point solution at kappa_tol = 1e-4)."""
    U = torch.tensor(np.asarray(x, dtype=np.float64)).reshape(3, 1)
    return info.project(U, grads=False)[0][:, 0].double().cpu().numpy()


def soc_projection_gradient(x, info: RocketInfo):
    """dynamics.jl:191-211: d soc_projection(x) / d x (3 x 3), the implicit gradient of the same solve."""
    U = torch.tensor(np.asarray(x, dtype=np.float64)).reshape(3, 1)
    return info.project(U, grads=True)[1][:, :, 0].double().cpu().numpy()
