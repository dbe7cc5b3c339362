"""`interior_point(z0, theta0; idx, r!, rz!, rtheta!, opts)` + `interior_point_solve!(ip)` as the reference uses them
directly for the rocket (src/models/rocket/dynamics.jl:34-43,68-86,109-262): the raw solve of r(z; theta) = 0 from a
caller-supplied starting point, with the implicit gradient dz/dtheta of the solution block.  The residual functions are
the library's compiled models (`od_ip_solve`)."""
import ctypes as C

import torch

from . import _lib
from .dynamics import _ptr


class InteriorPoint:
    def __init__(self, model_name, *, dtype=torch.float64, device="cuda", lib=None, options=None):
        self.name = model_name
        self.dtype = dtype
        self.device = torch.device(device)
        self.lib = lib if lib is not None else _lib.default_library()
        d = self.lib.model_dims(model_name)
        self.nz, self.ntheta = d["nz"], d["ntheta"]
        self.nzq, self.ngc = self.lib.raw_grad_dims(model_name)      # rows (solution block) x leading theta columns of dz
        o = self.lib.default_options(model_name)
        if options:
            for k, v in options.items():
                setattr(o, k, v)
        self.options = o
        hd = C.c_void_p()
        od_dtype = _lib.OD_F64 if dtype == torch.float64 else _lib.OD_F32
        with _lib.on_device(self.device):
            self.lib.check(self.lib.cdll.od_create(self.lib.model_id(model_name), od_dtype, C.byref(o), 0.0, C.byref(hd)))
        self._h = hd

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self.lib.cdll.od_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def solve(self, z0, theta, diff_sol=True):
        """z0: (nz, B), theta: (ntheta, B) -> z (nz, B), dz (nzq, ngc, B) or None, status (B,), iters (2, B).
        With diff_sol the solve runs to kappa_grad_tol first (gradient) and on to kappa_eval_tol (state); without,
        to kappa_eval_tol only."""
        if self.device.type == "cuda":
            self.lib.check(self.lib.cdll.od_set_stream(self._h, torch.cuda.current_stream(self.device).cuda_stream))
        z0 = z0.to(device=self.device, dtype=self.dtype).contiguous()
        theta = theta.to(device=self.device, dtype=self.dtype).contiguous()
        B = z0.shape[-1]
        z = torch.empty(self.nz, B, dtype=self.dtype, device=self.device)
        dz = torch.empty(self.nzq * self.ngc, B, dtype=self.dtype, device=self.device) if diff_sol else None
        st = torch.empty(B, dtype=torch.int32, device=self.device)
        it = torch.empty(2, B, dtype=torch.int32, device=self.device)
        self.lib.check(self.lib.cdll.od_ip_solve(self._h, B, _ptr(z0), _ptr(theta), _ptr(z), _ptr(dz), _ptr(st), _ptr(it)))
        if diff_sol:
            dz = dz.view(self.ngc, self.nzq, B).transpose(0, 1)
        return z, dz, st, it
