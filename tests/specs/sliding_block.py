"""A tenth model for the generator test (tests/test_model_generator.py): a point mass above a floor with Coulomb
friction -- one contact AND one friction cone, so the generated cooperative kernels carry a cone lane whose psi row
reads its partner contact's impulse (the structure of the reference's hopper and planar-push models, in miniature).

Recipe of the reference's models with friction (src/models/cartpole/model.jl:99-112 for the cone rows; hopper for
psi - mu*gamma): variational midpoint integrator; signed distance phi = y >= 0 with impact impulse gamma and slack s;
friction impulse b along x inside the cone |b| <= psi = mu*gamma, dual (s_psi, s_b) with s_b = tangential velocity.
z = [q2(2); gamma; s; psi; b; s_psi; s_b] (8), theta = [q0(2); q1(2); u(2); mu; h] (8)."""
import sympy as sp

from optimization_dynamics_amd.codegen.models import IP_DEFAULT, ModelSpec, _syms, cone_product, midpoint_del


def spec() -> ModelSpec:
    nq, nu = 2, 2
    nz, nth = 8, 8
    z, th, k = _syms("z", nz), _syms("th", nth), sp.Symbol("kappa", real=True)
    m, g, drag = 1.5, 9.81, 0.05

    def M(q):
        return sp.Matrix([[m, 0], [0, m]])

    def C(q, qd):                       # bias: gravity + a little viscous drag
        return sp.Matrix([drag * qd[0], m * g + drag * qd[1]])

    q0, q1 = sp.Matrix(th[0:2]), sp.Matrix(th[2:4])
    u1 = sp.Matrix(th[4:6])
    mu, h = th[6], th[7]
    q2 = sp.Matrix(z[0:2])
    gam, s, psi, b, spsi, sb = z[2], z[3], z[4], z[5], z[6], z[7]
    d, qm2, vm2 = midpoint_del(M, C, h, q0, q1, q2)
    dyn = d + u1 + sp.Matrix([b, gam])                  # friction along x, normal impulse along y
    vT = (q2[0] - q1[0]) / h
    r = list(dyn) + [s - q2[1], psi - mu * gam, vT - sb, gam * s - k]
    r += list(cone_product([psi, b], [spsi, sb]) - sp.Matrix([k, 0]))
    return ModelSpec(
        name="sliding_block", model_id=-1, nq=nq, nu=nu, nz=nz, nth=nth, z=z, th=th, kappa=k, r=r,
        ort=([2], [3]), soc=[([4, 5], [6, 7])], equr=[0, 1, 2, 3, 4], ortr=[5], socri=[[6, 7]], bil=[5, 6, 7],
        z_init=[("q", 0), ("q", 1), 1.0, 1.0, 1.0, 0.1, 1.0, 0.1], kind="mech", nfric=1, fric_default=[0.5],
        idx_zq=[0, 1], idx_gamma=[2], idx_b=[5],
        # slack row -> s, psi row -> psi, velocity row -> s_b, bilinear row -> gamma (pivot s, floored),
        # cone: tail row -> b after the runtime role swap (head <-> tail, b <-> s_psi)
        elim=[(2, 3), (3, 4), (4, 7), (5, 2), (7, 5)], floor_pivots=[(5, 2)],
        swaps=[((6, 7), (5, 6))],
        opts=dict(IP_DEFAULT, kappa_tol=1e-4, kappa_grad_tol=1e-3),
    )
