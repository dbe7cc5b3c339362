"""Host-side mirror of src/ls.jl: least squares sum_i |f_eta_i - f_z - M eta_i|^2 over M.

The reference drives a generic Newton iteration over Symbolics-generated cost / gradient / Hessian
(eval_cost!/eval_grad!/eval_hess!/update!, src/ls.jl:20-60).  The device solves the same problem
through its normal equations (od_ls_fit); `update_` keeps the reference's entry-point name.
"""
import numpy as np
import torch

from . import _lib
from .dynamics import _ptr


class LeastSquares:
    """src/ls.jl:1-18 (data only: N samples, f_z, f_eta[i], eta[i], theta = vec(M))."""

    def __init__(self, fz, feta, eta, owner):
        """fz: (ny,), feta: (ny, N), eta: (nz, N); owner: any object exposing .lib/._h/.device
        (an ImplicitDynamics handle provides the stream and device)."""
        self.fz = np.asarray(fz, dtype=np.float64)
        self.feta = np.asarray(feta, dtype=np.float64)
        self.eta = np.asarray(eta, dtype=np.float64)
        self.N = self.eta.shape[1]
        self.ny, self.nz = self.feta.shape[0], self.eta.shape[0]
        self.theta = np.zeros(self.ny * self.nz)
        self.owner = owner


def update_(ls: LeastSquares, tol=1.0e-8, verbose=False):
    """update!(ls) (src/ls.jl:44-60): on return ls.theta = vec(M) (column-major, ny x nz)."""
    o = ls.owner
    dev = o.device
    samples = np.concatenate([ls.fz.reshape(-1, 1), ls.feta], axis=1)      # (ny, N+1), sample 0 = f_z
    f_dev = torch.tensor(samples, dtype=torch.float64, device=dev).contiguous()       # BATCH_MINOR, B = 1
    eta_dev = torch.tensor(np.ascontiguousarray(ls.eta.T), dtype=torch.float64, device=dev)
    M = torch.empty(ls.ny * ls.nz, 1, dtype=torch.float64, device=dev)
    st = torch.empty(1, dtype=torch.int32, device=dev)
    o._use_current_stream()
    o.lib.check(o.lib.cdll.od_ls_fit(o._h, 1, ls.N, ls.ny, ls.nz, _ptr(eta_dev), _ptr(f_dev), _ptr(M), _ptr(st)))
    ls.theta[...] = M[:, 0].cpu().numpy()
    return ls.theta
