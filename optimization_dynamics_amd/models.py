"""Model objects mirroring the reference's exported model instances
(`acrobot_impact`, `cartpole_friction`, ... src/OptimizationDynamics.jl:75-79; `hopper` is
`RoboDojo.hopper`, examples/hopper.jl:14).  They carry only what the host API needs: dimensions and
the mutable friction vector (`cartpole_friction.friction .= [0.35; 0.35]`, examples/cartpole.jl:21);
the residual math itself lives in the generated device code (csrc/gen)."""
import ctypes as C

import numpy as np


class Model:
    def __init__(self, name, nq, nu, nw, nc, friction=(), **params):
        self.name = name
        self.nq, self.nu, self.nw, self.nc = nq, nu, nw, nc
        self.friction = np.array(friction, dtype=np.float64)   # friction_coefficients(model)
        for k, v in params.items():
            setattr(self, k, v)

    def __repr__(self):
        return "Model(%s, nq=%d, nu=%d)" % (self.name, self.nq, self.nu)


# src/models/acrobot/model.jl:159-163
acrobot_impact = Model("acrobot_impact", 2, 1, 0, 2)
acrobot_nominal = Model("acrobot_nominal", 2, 1, 0, 0)
# src/models/cartpole/model.jl:132-133
cartpole_friction = Model("cartpole_friction", 2, 1, 0, 2, friction=[0.1, 0.1])
cartpole_frictionless = Model("cartpole_frictionless", 2, 1, 0, 2)
# src/models/planar_push/model.jl:190-200
planarpush = Model("planar_push", 5, 2, 0, 5)
# src/models/rocket/model.jl:35-48
rocket = Model("rocket_dynamics", 12, 3, 0, 0)
# RoboDojo.hopper (un-vendored; constants recalled, see codegen/models.py HOPPER_PARAMS)
hopper = Model("hopper", 4, 2, 0, 4, friction=[0.5, 0.5], foot_radius=0.05, body_radius=0.1,
               gravity=9.81, mass_body=3.0, mass_foot=1.0)

BY_NAME = {m.name: m for m in [acrobot_impact, acrobot_nominal, cartpole_friction, cartpole_frictionless,
                               planarpush, rocket, hopper]}


def from_library(lib, name):
    """a Model for any model of the loaded library -- in particular one added with
    `python -m optimization_dynamics_amd.codegen --add spec.py`; friction = the generated defaults"""
    if name in BY_NAME:
        return BY_NAME[name]
    d = lib.model_dims(name)
    ix = lib.model_indices(name)
    fr = (C.c_double * 4)()
    lib.check(lib.cdll.od_default_friction(lib.model_id(name), fr, 4))
    m = Model(name, d["nq"], d["nu"], 0, len(ix["gamma"]), friction=[fr[i] for i in range(d["nfric"])])
    BY_NAME[name] = m
    return m
