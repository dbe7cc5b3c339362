"""A ninth model for the generator test (tests/test_model_generator.py): a pendulum with one joint limit, written the
way a user would write a new model for `python -m optimization_dynamics_amd.codegen --add tests/specs/pendulum_limit.py`.

Same recipe as the reference's models (src/models/acrobot/model.jl:90-157): variational midpoint integrator, one
signed distance phi(q) = q_max - q >= 0 with its impact impulse gamma and slack s, relaxed complementarity.
z = [q2; gamma; s] (3), theta = [q0; q1; u; h] (4)."""
import sympy as sp

from optimization_dynamics_amd.codegen.models import F, IP_DEFAULT, ModelSpec, _syms, midpoint_del

Q_MAX = 0.8


def spec() -> ModelSpec:
    nq, nu = 1, 1
    nz, nth = 3, 4
    z, th, k = _syms("z", nz), _syms("th", nth), sp.Symbol("kappa", real=True)
    m, l, g, damping = 1.0, 0.5, 9.81, 0.1

    def M(q):
        return sp.Matrix([[m * l * l]])

    def C(q, qd):                       # bias: gravity + viscous damping
        return sp.Matrix([m * g * l * sp.sin(q[0]) + damping * qd[0]])

    q0, q1 = sp.Matrix(th[0:1]), sp.Matrix(th[1:2])
    u1, h = th[2], th[3]
    q2, gam, s = sp.Matrix(z[0:1]), z[1], z[2]
    d, qm2, vm2 = midpoint_del(M, C, h, q0, q1, q2)
    phi = Q_MAX - q2[0]
    dyn = d[0] + u1 + sp.diff(phi, q2[0]) * gam          # impulse along the constraint normal
    r = [dyn, s - phi, gam * s - k]
    return ModelSpec(
        name="pendulum_limit", model_id=-1, nq=nq, nu=nu, nz=nz, nth=nth, z=z, th=th, kappa=k, r=r,
        ort=([1], [2]), soc=[], equr=[0, 1], ortr=[2], socri=[], bil=[2],
        z_init=[("q", 0), 1.0, 1.0], kind="mech", nfric=0, idx_zq=[0], idx_gamma=[1],
        elim=[(1, 2), (2, 1)], floor_pivots=[(2, 1)],
        opts=dict(IP_DEFAULT, kappa_tol=1e-4, kappa_grad_tol=1e-3),
    )
