"""optimization_dynamics_amd -- MI355X-native batched implicit-dynamics engine.

Drop-in for the hot path of thowell/optimization_dynamics (the per-timestep interior-point solve
and its implicit-function gradient behind the iLQR dynamics callbacks): hand-written HIP kernels
for gfx950 behind a C ABI (include/od_mi355x.h), with this package mirroring the reference's
Julia API (same names, argument meaning, in-place semantics).
"""
from ._lib import Library, ODError, Options, default_library  # noqa: F401
from .dynamics import ImplicitDynamics, f, ffxfu, fu, fx, state_to_configuration  # noqa: F401
from .gradient_bundle import GradientBundle, MInfo, f_gb, fu_gb, fx_gb, gradient_, gradient_batch  # noqa: F401
from .ilqr import ILQR, QuadraticObjective  # noqa: F401
from .interior_point import InteriorPoint  # noqa: F401
from .ls import LeastSquares, update_  # noqa: F401
from .models import (acrobot_impact, acrobot_nominal, cartpole_friction, cartpole_frictionless,  # noqa: F401
                     hopper, planarpush, rocket)
from .rocket import (RocketDynamics, RocketInfo, f_rocket, f_rocket_proj, fu_rocket, fu_rocket_proj, fx_rocket,  # noqa: F401
                     fx_rocket_proj, soc_projection, soc_projection_gradient)
