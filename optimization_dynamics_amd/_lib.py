"""ctypes binding of libod_mi355x.so (include/od_mi355x.h).

There is NO fallback: if the HIP library has not been built (``python -c 'import __graft_entry__ as g;
g.build()'`` or ``make -C optimization_dynamics_amd/csrc``) loading fails loudly, and ``od_create``
fails with OD_ERR_NO_DEVICE when no MI355X is visible.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_PATH = os.path.join(_HERE, "libod_mi355x.so")

MODEL_IDS = {
    "acrobot_impact": 0, "acrobot_nominal": 1, "cartpole_friction": 2, "cartpole_frictionless": 3,
    "planar_push": 4, "rocket_dynamics": 5, "rocket_projection": 6, "hopper": 7,
}
OD_F64, OD_F32 = 0, 1
LAYOUT_BATCH_MINOR, LAYOUT_BATCH_MAJOR = 0, 1
STATUS_EVAL_OK, STATUS_GRAD_OK, STATUS_FACTOR_OK = 1, 2, 4


class Options(C.Structure):
    """od_options == InteriorPointOptions preset (src/dynamics.jl:25-33)."""
    _fields_ = [("r_tol", C.c_double), ("kappa_eval_tol", C.c_double), ("kappa_grad_tol", C.c_double),
                ("max_iter", C.c_int), ("max_ls", C.c_int),
                ("eps_min", C.c_double), ("kappa_reg", C.c_double), ("gamma_reg", C.c_double),
                ("undercut", C.c_double)]


class IlqrOptions(C.Structure):
    """od_ilqr_options (cf. iLQR.Options, examples/acrobot.jl:98-107)"""
    _fields_ = [("reg", C.c_double), ("c1", C.c_double), ("obj_tol", C.c_double), ("con_tol", C.c_double),
                ("rho_init", C.c_double), ("rho_scale", C.c_double), ("max_iter", C.c_int), ("max_al_iter", C.c_int),
                ("project", C.c_int), ("history", C.c_int), ("proj_stall_exit", C.c_int), ("rho_max", C.c_double)]


class IlqrParameterStage(C.Structure):
    """od_ilqr_parameter_stage"""
    _fields_ = [("constraint", C.c_int), ("n_p", C.c_int), ("p", C.POINTER(C.c_double)), ("w_theta", C.POINTER(C.c_double)),
                ("cost_const", C.c_double), ("nt", C.c_int), ("nt_ineq", C.c_int), ("Ct_x", C.POINTER(C.c_double)),
                ("Ct_theta", C.POINTER(C.c_double)), ("dt", C.POINTER(C.c_double))]


class IlqrInfo(C.Structure):
    """od_ilqr_info"""
    _fields_ = [("iterations", C.c_int), ("al_iterations", C.c_int), ("done", C.c_int), ("al_done", C.c_int),
                ("bad_linearisations", C.c_int), ("reg", C.c_double), ("rho", C.c_double), ("max_dJ", C.c_double),
                ("max_violation", C.c_double)]


class ODError(RuntimeError):
    pass


_VP = C.c_void_p
_IP = C.c_void_p   # int* passed as raw address (0 = NULL)

# every symbol declared in include/od_mi355x.h: (restype, argtypes)
SIGNATURES = {
    "od_version": (C.c_int, []),
    "od_last_error": (C.c_char_p, []),
    "od_model_dims": (C.c_int, [C.c_int] + [C.POINTER(C.c_int)] * 5),
    "od_default_friction": (C.c_int, [C.c_int, C.POINTER(C.c_double), C.c_int]),
    "od_num_models": (C.c_int, []),
    "od_model_id": (C.c_int, [C.c_char_p]),
    "od_model_name": (C.c_char_p, [C.c_int]),
    "od_default_options": (C.c_int, [C.c_int, C.POINTER(Options)]),
    "od_create": (C.c_int, [C.c_int, C.c_int, C.POINTER(Options), C.c_double, C.POINTER(_VP)]),
    "od_destroy": (C.c_int, [_VP]),
    "od_get_device": (C.c_int, [_VP, C.POINTER(C.c_int)]),
    "od_set_options": (C.c_int, [_VP, C.POINTER(Options)]),
    "od_get_options": (C.c_int, [_VP, C.POINTER(Options)]),
    "od_set_timestep": (C.c_int, [_VP, C.c_double]),
    "od_set_friction": (C.c_int, [_VP, C.POINTER(C.c_double), C.c_int]),
    "od_set_u_max": (C.c_int, [_VP, C.c_double]),
    "od_set_layout": (C.c_int, [_VP, C.c_int]),
    "od_set_projection_stall_exit": (C.c_int, [_VP, C.c_int]),
    "od_set_mixed_precision": (C.c_int, [_VP, C.c_int]),
    "od_set_stream": (C.c_int, [_VP, _VP]),
    "od_set_launch_config": (C.c_int, [_VP, C.c_int, C.c_int]),
    "od_set_cooperative": (C.c_int, [_VP, C.c_int]),
    "od_uses_cooperative": (C.c_int, [_VP, C.c_long]),
    "od_get_grad_iterates": (C.c_int, [_VP, C.c_long, _VP]),
    "od_synchronize": (C.c_int, [_VP]),
    "od_step": (C.c_int, [_VP, C.c_long, _VP, _VP, _VP, _IP, _IP]),
    "od_step_grad": (C.c_int, [_VP, C.c_long, _VP, _VP, _VP, _VP, _VP, _IP, _IP]),
    "od_step_grad_compact": (C.c_int, [_VP, C.c_long, _VP, _VP, _VP, _VP, _IP, _IP]),
    "od_rollout": (C.c_int, [_VP, C.c_long, C.c_int, _VP, _VP, _VP, _VP, _VP, _IP, _IP]),
    "od_rollout_compact": (C.c_int, [_VP, C.c_long, C.c_int, _VP, _VP, _VP, _VP, _IP, _IP]),
    "od_rollout_policy": (C.c_int, [_VP, C.c_long, C.c_int, C.c_int, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _IP, _IP]),
    "od_quad_cost": (C.c_int, [_VP, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "od_ilqr_backward": (C.c_int, [_VP, C.c_long, C.c_int, C.c_int, C.c_int, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, C.c_double, _VP, _VP, _VP, _IP]),
    "od_ilqr_default_options": (C.c_int, [C.POINTER(IlqrOptions)]),
    "od_ilqr_create": (C.c_int, [_VP, C.c_long, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(IlqrOptions), C.POINTER(_VP)]),
    "od_ilqr_destroy": (C.c_int, [_VP]),
    "od_ilqr_set_objective": (C.c_int, [_VP, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    "od_ilqr_set_constraints": (C.c_int, [_VP, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                          C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "od_ilqr_set_parameter_stage": (C.c_int, [_VP, C.POINTER(IlqrParameterStage)]),
    "od_ilqr_set_gradient_bundle": (C.c_int, [_VP, C.c_int, _VP]),
    "od_num_constraints": (C.c_int, []),
    "od_constraint_id": (C.c_int, [C.c_char_p]),
    "od_constraint_name": (C.c_char_p, [C.c_int]),
    "od_constraint_dims": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "od_ilqr_init": (C.c_int, [_VP, _VP, _VP]),
    "od_ilqr_iterate": (C.c_int, [_VP, C.c_int]),
    "od_ilqr_al_update": (C.c_int, [_VP]),
    "od_ilqr_solve": (C.c_int, [_VP, _VP, _VP]),
    "od_ilqr_get": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP]),
    "od_ilqr_get_history": (C.c_int, [_VP, _VP, C.c_int]),
    "od_ilqr_get_status": (C.c_int, [_VP, _VP, _VP, _VP]),
    "od_ilqr_get_trace": (C.c_int, [_VP, _VP, _VP, _VP, C.c_int]),
    "od_ilqr_get_info": (C.c_int, [_VP, C.POINTER(IlqrInfo)]),
    "od_bundle_workspace_bytes": (C.c_size_t, [_VP, C.c_long, C.c_int]),
    "od_bundle_grad": (C.c_int, [_VP, C.c_long, C.c_int, _VP, _VP, _VP, _VP, _VP, C.c_size_t, _IP]),
    "od_ls_fit": (C.c_int, [_VP, C.c_long, C.c_int, C.c_int, C.c_int, _VP, _VP, _VP, _IP]),
    "od_raw_grad_dims": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "od_ip_solve": (C.c_int, [_VP, C.c_long, _VP, _VP, _VP, _VP, _IP, _IP]),
    "od_rocket": (C.c_int, [_VP, C.c_long, C.c_int, _VP, _VP, _VP, _VP, _VP, _VP, _IP]),
    "od_soc_project": (C.c_int, [_VP, C.c_long, _VP, _VP, _VP, _IP]),
    "od_soc_project_full": (C.c_int, [_VP, C.c_long, _VP, _VP, _VP, _IP, _IP]),
    "od_model_indices": (C.c_int, [C.c_int, C.c_int, _IP, C.c_int]),
    "od_step_full": (C.c_int, [_VP, C.c_long, _VP, _VP, _VP, _VP, _IP, _IP]),
    "od_rocket_rollout": (C.c_int, [_VP, C.c_long, C.c_int, C.c_int, _VP, C.c_int, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _IP]),
    "od_f_host": (C.c_int, [_VP, _VP, _VP, _VP]),
    "od_fx_host": (C.c_int, [_VP, _VP, _VP, _VP]),
    "od_fu_host": (C.c_int, [_VP, _VP, _VP, _VP]),
    "od_ffxfu_host": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP]),
    "od_rocket_host": (C.c_int, [_VP, C.c_int, _VP, _VP, _VP, _VP, _VP, _VP, _IP]),
    "od_soc_project_host": (C.c_int, [_VP, _VP, _VP, _VP, _IP]),
    "od_bundle_grad_host": (C.c_int, [_VP, C.c_int, _VP, _VP, _VP, _VP, _IP]),
    "od_comm_unique_id": (C.c_int, [_VP]),
    "od_comm_create": (C.c_int, [_VP, _VP, C.c_int, C.c_int, C.POINTER(_VP)]),
    "od_comm_info": (C.c_int, [_VP, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "od_comm_destroy": (C.c_int, [_VP]),
    "od_allgather_compact": (C.c_int, [_VP, _VP, C.c_long, C.c_int, _VP, _VP, _VP, _VP]),
    "od_comm_allgather": (C.c_int, [_VP, _VP, _VP, _VP, C.c_size_t]),
}

# include/od_mi355x.h: OD_ABI_VERSION this mirror was written against (struct layouts above, defaults the wrappers rely on)
ABI_VERSION = 101
COMM_ID_BYTES = 128


class Library:
    """A loaded libod_mi355x.so with typed entry points; errors raise ODError."""

    def __init__(self, path=None):
        self.path = path or DEFAULT_PATH
        if not os.path.exists(self.path):
            raise ODError(
                "HIP extension %s is missing: build it with `make -C %s -j8` (hipcc, gfx950). "
                "There is no CPU fallback." % (self.path, os.path.join(_HERE, "csrc")))
        self.cdll = C.CDLL(self.path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(self.cdll, name)   # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        got = self.cdll.od_version()
        if got != ABI_VERSION:
            raise ODError("%s reports ABI version %d, this binding was written against %d (include/od_mi355x.h: OD_ABI_VERSION): rebuild the "
                          "library -- a stale one reads od_ilqr_options of another size" % (self.path, got, ABI_VERSION))
        # the library's own registry: the eight models of the reference plus whatever the generator added (--add)
        self.model_ids = {self.cdll.od_model_name(i).decode(): i for i in range(self.cdll.od_num_models())}

    def model_id(self, name):
        if name not in self.model_ids:
            raise ODError("model %r is not in this library (has: %s)" % (name, ", ".join(self.model_ids)))
        return self.model_ids[name]

    def check(self, rc):
        if rc != 0:
            raise ODError("libod_mi355x error %d: %s" % (rc, self.cdll.od_last_error().decode()))

    def model_dims(self, model):
        v = [C.c_int() for _ in range(5)]
        self.check(self.cdll.od_model_dims(self.model_id(model), *[C.byref(x) for x in v]))
        return dict(zip(["nq", "nu", "nz", "ntheta", "nfric"], [x.value for x in v]))

    def model_indices(self, model):
        """z indices (0-based) of the next configuration, the impact impulses gamma and the friction impulses b"""
        out = {}
        for key, which in (("q", 0), ("gamma", 1), ("b", 2)):
            buf = (C.c_int * 16)()
            n = self.cdll.od_model_indices(self.model_id(model), which, buf, 16)
            if n < 0:
                self.check(n)
            out[key] = [buf[i] for i in range(n)]
        return out

    def raw_grad_dims(self, model):
        a, b = C.c_int(), C.c_int()
        self.check(self.cdll.od_raw_grad_dims(self.model_id(model), C.byref(a), C.byref(b)))
        return a.value, b.value

    def default_options(self, model):
        o = Options()
        self.check(self.cdll.od_default_options(self.model_id(model), C.byref(o)))
        return o


def on_device(device):
    """context manager: `device` is the current HIP device inside (od_create makes its handle on the current device)"""
    import contextlib
    import torch
    device = torch.device(device)
    if device.type == "cuda":
        return torch.cuda.device(device)
    return contextlib.nullcontext()


_default = None


def default_library():
    global _default
    if _default is None:
        _default = Library()
    return _default
