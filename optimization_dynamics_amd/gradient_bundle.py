"""Host-side mirror of the reference's zero-order "gradient bundle"
(src/gradient_bundle.jl): N single-coordinate Gaussian perturbations of (q1, q2, u1), N+1
contact-implicit steps with the EVAL simulator and a least-squares fit of d q3 / d(q1, q2, u1).

On the device the N+1 steps of every knot run as independent lanes of one kernel launch
(od_bundle_grad), followed by the fit (normal equations; the cost of src/gradient_bundle.jl:35-39
is exactly quadratic so the reference's Newton iteration of src/ls.jl:44-60 ends on the same
minimiser).
"""
import numpy as np
import torch

from .dynamics import ImplicitDynamics, _ptr


class MInfo:
    """src/gradient_bundle.jl:1-6 (index bookkeeping, 0-based)."""

    def __init__(self, nq, nu):
        self.idx_q1 = list(range(nq))
        self.idx_q2 = list(range(nq, 2 * nq))
        self.idx_u1 = list(range(2 * nq, 2 * nq + nu))


class GradientBundle:
    """src/gradient_bundle.jl:15-85.  `eta[:, i]` has exactly one nonzero, at a uniformly random
    coordinate, with value eps * randn (:49-54).  Unlike the reference (which sizes its scratch with
    the module-global nq/nu, :79-81) all sizes come from the model.  `seed` makes the draw
    reproducible (the reference uses the unseeded global RNG)."""

    def __init__(self, model, N=100, eps=1.0e-4, seed=None, eta=None):
        self.model = model
        self.N = N
        self.eps = eps
        self.ny = model.nq
        self.nz = 2 * model.nq + model.nu
        self.info = MInfo(model.nq, model.nu)
        if eta is None:
            rng = np.random.default_rng(seed)
            eta = np.zeros((self.nz, N))
            for i in range(N):
                w = eps * rng.standard_normal()
                eta[rng.integers(self.nz), i] = w
        self.eta = np.asfortranarray(eta, dtype=np.float64)
        assert self.eta.shape == (self.nz, N)
        self.dz = np.zeros((self.ny, self.nz))
        self._eta_dev = None
        self._ws = None

    def _device_eta(self, device):
        if self._eta_dev is None or self._eta_dev.device != device:
            # (nzb x N) column-major == N rows of nzb in C order
            self._eta_dev = torch.tensor(np.ascontiguousarray(self.eta.T), dtype=torch.float64, device=device)
        return self._eta_dev


def gradient_batch(im: ImplicitDynamics, gb: GradientBundle, X, U):
    """gradient! (src/gradient_bundle.jl:87-104) for B knots at once.
    X: (2nq, B), U: (nu, B) -> dz (nq, 2nq+nu, B), status (B,) [1 = Gram matrix non-singular and fit finite]."""
    im._sync_friction(); im._use_current_stream()
    X, U = im._prep(X), im._prep(U)
    B = X.shape[-1]
    nq, nzb = gb.ny, gb.nz
    eta = gb._device_eta(im.device)
    need = im.lib.cdll.od_bundle_workspace_bytes(im._h, B, gb.N)
    if gb._ws is None or gb._ws.numel() < need or gb._ws.device != im.device:
        gb._ws = torch.empty(need, dtype=torch.uint8, device=im.device)
    out = im._new(nq * nzb, B)
    st = im._new(B, dtype=torch.int32)
    im.lib.check(im.lib.cdll.od_bundle_grad(im._h, B, gb.N, _ptr(X), _ptr(U), _ptr(eta), _ptr(out), _ptr(gb._ws), need, _ptr(st)))
    return out.view(nzb, nq, B).transpose(0, 1), st


def gradient_(im: ImplicitDynamics, gb: GradientBundle, q1, q2, u1):
    """gradient!(sim, gb, q1, q2, u1) -> gb.dz (ny x nz), src/gradient_bundle.jl:87-104."""
    x = np.concatenate([np.asarray(q1, dtype=np.float64), np.asarray(q2, dtype=np.float64)])
    dz, _ = gradient_batch(im, gb, torch.tensor(x).reshape(-1, 1), torch.tensor(np.asarray(u1, dtype=np.float64)).reshape(-1, 1))
    gb.dz[...] = dz[:, :, 0].cpu().numpy()
    return gb.dz


def fx_gb(dx, model: ImplicitDynamics, x, u, w=None):
    """src/gradient_bundle.jl:109-126 (model.info must be a GradientBundle)."""
    nq = model.model.nq
    x = np.asarray(x, dtype=np.float64)
    for i in range(nq):
        dx[model.idx_q1[i], model.idx_q2[i]] = 1.0
    dz = gradient_(model, model.info, x[:nq], x[nq:2 * nq], u)
    dx[np.ix_(model.idx_q2, model.idx_q1)] = dz[:, :nq]
    dx[np.ix_(model.idx_q2, model.idx_q2)] = dz[:, nq:2 * nq]
    return dx


def fu_gb(du, model: ImplicitDynamics, x, u, w=None):
    """src/gradient_bundle.jl:136-147."""
    nq = model.model.nq
    x = np.asarray(x, dtype=np.float64)
    dz = gradient_(model, model.info, x[:nq], x[nq:2 * nq], u)
    du[model.idx_q2, :] = dz[:, 2 * nq:]
    return du


def f_gb(d, model: ImplicitDynamics, x, u, w=None):
    """Exported but never defined in the reference (src/OptimizationDynamics.jl:34); the zero-order
    bundle leaves the step itself unchanged, so this is `f`."""
    from .dynamics import f
    return f(d, model, x, u, w)
