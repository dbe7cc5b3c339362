"""Symbolic (sympy) statements of the optimization-based dynamics residuals.

Counterpart of the reference's build-time codegen (``src/models/*/codegen.jl`` +
``deps/build.jl:27-48``): each model is a residual r(z; theta, kappa) whose Jacobians
rz, rtheta are derived symbolically and emitted as straight-line C (``emit.py``).

Every function cites the reference lines it restates.  Index sets are 0-based here
(the reference is 1-based).  The hopper lives in the un-vendored RoboDojo.jl; its
statement below follows the structural pins in the reference (``examples/hopper.jl``,
``examples/comparisons/hopper.jl``) and the recalled RoboDojo source -- constants are
UNVERIFIED ("parity unpinned").
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import sympy as sp

R = sp.Rational
F = sp.Float


@dataclass
class ModelSpec:
    name: str
    model_id: int
    nq: int            # configuration dim (mechanical) / state dim (rocket)
    nu: int
    nz: int
    nth: int
    z: List[sp.Symbol]
    th: List[sp.Symbol]
    kappa: sp.Symbol
    r: List[sp.Expr]
    # cone structure (reference: IndicesOptimization positional fields
    # (nz, nD, ortz, ortD, socz, socD, equr, ortr, socr, socri, bil),
    # e.g. src/models/cartpole/simulator_friction.jl:22-33)
    ort: Tuple[List[int], List[int]]                 # (primal members, dual members)
    soc: List[Tuple[List[int], List[int]]]           # per cone (primal idx, dual idx)
    equr: List[int]
    ortr: List[int]
    socri: List[List[int]]
    bil: List[int]
    # initial guess for z (reference initialize_z!): entries are ('q', i) -> current
    # configuration component i, or a float constant
    z_init: List[object] = field(default_factory=list)
    # theta layout
    kind: str = "mech"      # "mech": theta=[q0;q1;u;fric;h]   "rocket": theta=[x;u;h]  "proj": theta=[u;umax]
    nfric: int = 0
    fric_default: List[float] = field(default_factory=list)
    # which z entries are the next configuration, which theta slots feed the gradient
    idx_zq: List[int] = field(default_factory=list)
    # which z entries are the impact impulses gamma and the friction impulses b (RoboDojo `process!` /
    # sim.traj.gamma, sim.traj.b; sized by nc / nb in src/dynamics.jl:36-46)
    idx_gamma: List[int] = field(default_factory=list)
    idx_b: List[int] = field(default_factory=list)
    # device elimination order: list of (row, col) static pivots
    elim: List[Tuple[int, int]] = field(default_factory=list)
    # static pivots whose value is a strictly positive cone variable that may underflow towards 0
    # (orthant slack): floored at OD_PIVOT_FLOOR in the factorisation
    floor_pivots: List[Tuple[int, int]] = field(default_factory=list)
    # runtime role swaps for second-order-cone blocks: ((row_a,row_b),(col_a,col_b)); rows/cols are
    # exchanged when |A[row_a,col_b]| > |A[row_b,col_a]| (psi vs s_psi: sticking vs sliding mode)
    swaps: List[Tuple[Tuple[int, int], Tuple[int, int]]] = field(default_factory=list)
    # the dense tail may be factored without runtime pivoting inside the interior-point iterations (measured per
    # model against the pivoted factorisation: identical iteration counts and iterates); the implicit-gradient solve
    # at the converged point always pivots
    static_tail: bool = False
    # a second elimination order for those iterations that continues through the cone leftovers (after the role swap every
    # pivot of it is positive at interior points: second tail row on b_2 with pivot s_psi resp. s_psi - b_2 s_b2 / psi, then
    # the head row with the Schur complement of the cone block), leaving the configuration block as the only dense tail
    elim_state: List[Tuple[int, int]] = field(default_factory=list)
    state_tail_pivot: bool = True      # whether that configuration tail keeps runtime partial pivoting
    # default solver options (reference src/dynamics.jl:25-33 etc.)
    opts: Dict[str, float] = field(default_factory=dict)
    notes: str = ""


def _syms(prefix, n):
    # symbols are named as the C array elements they are printed as (z[3], th[7], ...)
    if prefix in ("z", "th"):
        return [sp.Symbol(f"{prefix}[{i}]", real=True) for i in range(n)]
    return [sp.Symbol(f"{prefix}{i}", real=True) for i in range(n)]


def lagrangian_derivatives(M, C, q, v):
    """RoboDojo.lagrangian_derivatives [RECALL, consistent with every call site, e.g.
    src/models/acrobot/model.jl:97-100]: D1L = -C(q, v), D2L = M(q) v."""
    D1L = -C(q, v)
    D2L = M(q) * v
    return D1L, D2L


def cone_product(a, b):
    """RoboDojo.cone_product, pinned by usage at src/models/cartpole/model.jl:111-112:
    [a'b ; a0*b1: + b0*a1:]."""
    a = sp.Matrix(a)
    b = sp.Matrix(b)
    out = [(a.T * b)[0, 0]]
    for i in range(1, a.shape[0]):
        out.append(a[0] * b[i] + b[0] * a[i])
    return sp.Matrix(out)


def midpoint_del(M, C, h, q0, q1, q2):
    """Variational midpoint integrator, src/models/acrobot/model.jl:90-100."""
    qm1 = (q0 + q1) * F(0.5)
    vm1 = (q1 - q0) / h
    qm2 = (q1 + q2) * F(0.5)
    vm2 = (q2 - q1) / h
    D1L1, D2L1 = lagrangian_derivatives(M, C, qm1, vm1)
    D1L2, D2L2 = lagrangian_derivatives(M, C, qm2, vm2)
    d = F(0.5) * h * D1L1 + D2L1 + F(0.5) * h * D1L2 - D2L2
    return d, qm2, vm2


IP_DEFAULT = dict(r_tol=1e-8, kappa_tol=1e-4, max_iter=100, max_ls=25, eps_min=0.25,
                  kappa_reg=1e-3, gamma_reg=0.1, undercut=float("inf"))


# ---------------------------------------------------------------------------------
# acrobot (src/models/acrobot/model.jl)
# ---------------------------------------------------------------------------------
def _acrobot_common():
    m1, J1, l1, lc1 = 1.0, 0.333, 1.0, 0.5     # model.jl:159-160
    m2, J2, l2, lc2 = 1.0, 0.333, 1.0, 0.5
    g = 9.81

    def M(x):  # model.jl:41-51
        a = J1 + J2 + m2 * l1 * l1 + 2.0 * m2 * l1 * lc2 * sp.cos(x[1])
        b = J2 + m2 * l1 * lc2 * sp.cos(x[1])
        c = J2
        return sp.Matrix([[a, b], [b, c]])

    def tau(x):  # model.jl:53-61
        a = (-1.0 * m1 * g * lc1 * sp.sin(x[0])
             - m2 * g * (l1 * sp.sin(x[0]) + lc2 * sp.sin(x[0] + x[1])))
        b = -1.0 * m2 * g * lc2 * sp.sin(x[0] + x[1])
        return sp.Matrix([a, b])

    def c(q, qd):  # model.jl:63-71
        a = -2.0 * m2 * l1 * lc2 * sp.sin(q[1]) * qd[1]
        b = -1.0 * m2 * l1 * lc2 * sp.sin(q[1]) * qd[1]
        cc = m2 * l1 * lc2 * sp.sin(q[1]) * qd[0]
        return sp.Matrix([[a, b], [cc, 0]])

    def C(q, qd):  # model.jl:77-79
        return c(q, qd) * qd - tau(q)

    return M, C


def acrobot_impact() -> ModelSpec:
    nq, nu, nc = 2, 1, 2
    nz, nth = 6, 6
    z, th, k = _syms("z", nz), _syms("th", nth), sp.Symbol("kappa", real=True)
    M, C = _acrobot_common()
    q0, q1 = sp.Matrix(th[0:2]), sp.Matrix(th[2:4])     # model.jl:126-129
    u1, h = th[4], th[5]
    q2, lam, s = sp.Matrix(z[0:2]), sp.Matrix(z[2:4]), sp.Matrix(z[4:6])  # :131-133
    d, qm2, vm2 = midpoint_del(M, C, h, q0, q1, q2)
    phi = sp.Matrix([F(0.5) * sp.pi - q2[1], q2[1] + F(0.5) * sp.pi])  # :81-83
    P = phi.jacobian(q2)                                                 # :85-88
    B = sp.Matrix([0, 1])                                                # :73-75
    dyn = d + B * u1 + P.T * lam - h * F(0.5) * vm2                      # :100-103
    r = list(dyn) + list(s - phi) + [lam[i] * s[i] - k for i in range(nc)]  # :135-141
    return ModelSpec(
        name="acrobot_impact", model_id=0, nq=nq, nu=nu, nz=nz, nth=nth, z=z, th=th, kappa=k, r=r,
        # simulator_impact.jl:16-31
        ort=([2, 3], [4, 5]), soc=[], equr=[0, 1, 2, 3], ortr=[4, 5], socri=[], bil=[4, 5],
        z_init=[("q", 0), ("q", 1), 1.0, 1.0, 1.0, 1.0],                   # :34-38
        kind="mech", nfric=0, idx_zq=[0, 1], idx_gamma=[2, 3],
        elim=[(2, 4), (3, 5), (4, 2), (5, 3)], floor_pivots=[(4, 2), (5, 3)],
        opts=dict(IP_DEFAULT, kappa_tol=1e-4, kappa_grad_tol=1e-3),       # examples/acrobot.jl:15-23
    )


def acrobot_nominal() -> ModelSpec:
    nq, nu = 2, 1
    nz, nth = 2, 6
    z, th, k = _syms("z", nz), _syms("th", nth), sp.Symbol("kappa", real=True)
    M, C = _acrobot_common()
    q0, q1 = sp.Matrix(th[0:2]), sp.Matrix(th[2:4])
    u1, h = th[4], th[5]
    q2 = sp.Matrix(z[0:2])
    d, qm2, vm2 = midpoint_del(M, C, h, q0, q1, q2)
    B = sp.Matrix([0, 1])
    dyn = d + B * u1 - h * F(0.5) * vm2                                  # model.jl:106-119
    return ModelSpec(
        name="acrobot_nominal", model_id=1, nq=nq, nu=nu, nz=nz, nth=nth, z=z, th=th, kappa=k,
        r=list(dyn), ort=([], []), soc=[], equr=[0, 1], ortr=[], socri=[], bil=[],
        z_init=[("q", 0), ("q", 1)], kind="mech", nfric=0, idx_zq=[0, 1], elim=[],
        opts=dict(IP_DEFAULT, kappa_tol=1.0, kappa_grad_tol=1.0),         # examples/acrobot.jl:25-27
    )


# ---------------------------------------------------------------------------------
# cartpole (src/models/cartpole/model.jl)
# ---------------------------------------------------------------------------------
def _cartpole_common():
    mc, mp, l, g = 1.0, 0.2, 0.5, 9.81                                   # model.jl:132-133

    def M(x):  # :28-32
        return sp.Matrix([[mc + mp, mp * l * sp.cos(x[1])],
                          [mp * l * sp.cos(x[1]), mp * l ** 2.0]])

    def C(q, qd):  # :43-49
        Cm = sp.Matrix([[0, -1.0 * mp * qd[1] * l * sp.sin(q[1])], [0, 0]])
        G = sp.Matrix([0, mp * g * l * sp.sin(q[1])])
        return -Cm * qd + G

    return M, C, (mc, mp, l, g)


def cartpole_friction() -> ModelSpec:
    nq, nu, nc = 2, 1, 2
    nz, nth = 10, 8
    z, th, k = _syms("z", nz), _syms("th", nth), sp.Symbol("kappa", real=True)
    M, C, (mc, mp, l, g) = _cartpole_common()
    q0, q1 = sp.Matrix(th[0:2]), sp.Matrix(th[2:4])                      # :86-91
    u1, mu_s, mu_a, h = th[4], th[5], th[6], th[7]
    q2 = sp.Matrix(z[0:2])                                               # :93-97
    psi, b, spsi, sb = z[2:4], z[4:6], z[6:8], z[8:10]
    vT1 = (q2[0] - q1[0]) / h                                            # :99-100
    vT2 = (q2[1] - q1[1]) / h
    d, qm2, vm2 = midpoint_del(M, C, h, q0, q1, q2)
    B = sp.Matrix([1, 0])
    dyn = d + B * u1 + sp.Matrix(b)                                      # :51-64 (P = I)
    r = list(dyn) + [
        sb[0] - vT1,                                                     # :107
        psi[0] - mu_s * (mp + mc) * g * h,                               # :108
        sb[1] - vT2,                                                     # :109
        psi[1] - mu_a * (mp * g * l) * h,                                # :110
    ]
    r += list(cone_product([psi[0], b[0]], [spsi[0], sb[0]]) - sp.Matrix([k, 0]))   # :111
    r += list(cone_product([psi[1], b[1]], [spsi[1], sb[1]]) - sp.Matrix([k, 0]))   # :112
    return ModelSpec(
        name="cartpole_friction", model_id=2, nq=nq, nu=nu, nz=nz, nth=nth, z=z, th=th, kappa=k, r=r,
        # simulator_friction.jl:22-33
        ort=([], []), soc=[([2, 4], [6, 8]), ([3, 5], [7, 9])],
        equr=list(range(6)), ortr=[], socri=[[6, 7], [8, 9]], bil=[6, 7, 8, 9],
        z_init=[("q", 0), ("q", 1), 1.0, 1.0, 0.1, 0.1, 1.0, 1.0, 0.1, 0.1],   # :36-42
        kind="mech", nfric=2, fric_default=[0.1, 0.1], idx_zq=[0, 1], idx_b=[4, 5],
        # rows: 2: sb0-vT1 (pivot sb0=z8), 3: psi0-.. (pivot psi0=z2), 4: sb1 (z9), 5: psi1 (z3)
        # cone rows (6,7) / (8,9): one local pivot (tail row -> b, coefficient s_psi) after a runtime
        # role swap (head<->tail, b<->s_psi) when psi > s_psi; the other row/unknown goes to the tail
        elim=[(2, 8), (4, 9), (3, 2), (5, 3), (7, 4), (9, 5)],
        swaps=[((6, 7), (4, 6)), ((8, 9), (5, 7))],
        opts=dict(IP_DEFAULT, kappa_tol=1e-4, kappa_grad_tol=1e-4),       # examples/cartpole.jl:20
    )


def cartpole_frictionless() -> ModelSpec:
    nq, nu = 2, 1
    nz, nth = 2, 6
    z, th, k = _syms("z", nz), _syms("th", nth), sp.Symbol("kappa", real=True)
    M, C, _ = _cartpole_common()
    q0, q1 = sp.Matrix(th[0:2]), sp.Matrix(th[2:4])                      # :119-122
    u1, h = th[4], th[5]
    q2 = sp.Matrix(z[0:2])
    d, qm2, vm2 = midpoint_del(M, C, h, q0, q1, q2)
    dyn = d + sp.Matrix([1, 0]) * u1                                     # :66-79
    return ModelSpec(
        name="cartpole_frictionless", model_id=3, nq=nq, nu=nu, nz=nz, nth=nth, z=z, th=th, kappa=k,
        r=list(dyn), ort=([], []), soc=[], equr=[0, 1], ortr=[], socri=[], bil=[],
        z_init=[("q", 0), ("q", 1)], kind="mech", nfric=0, idx_zq=[0, 1], elim=[],
        opts=dict(IP_DEFAULT, kappa_tol=1.0, kappa_grad_tol=1.0),
    )


# ---------------------------------------------------------------------------------
# planar push (src/models/planar_push/model.jl)
# ---------------------------------------------------------------------------------
def planar_push() -> ModelSpec:
    nq, nu = 5, 2
    nz, nth = 35, 13
    z, th, k = _syms("z", nz), _syms("th", nth), sp.Symbol("kappa", real=True)
    r_dim = 0.1                                                          # model.jl:24
    mu_surface, mu_pusher, gravity = 0.5, 0.5, 9.81                      # :43-45
    mass_block, mass_pusher = 1.0, 10.0                                  # :46-47
    inertia = 1.0 / 12.0 * mass_block * ((2.0 * r_dim) ** 2 + (2.0 * r_dim) ** 2)  # :48
    cc = [(r_dim, r_dim), (-r_dim, r_dim), (r_dim, -r_dim), (-r_dim, -r_dim)]      # :34-39

    def rot(x):  # :62-64
        return sp.Matrix([[sp.cos(x), -sp.sin(x)], [sp.sin(x), sp.cos(x)]])

    def sd_2d_box(p, pose):  # :26-31
        D = rot(-pose[2]) * (sp.Matrix(p) - sp.Matrix(pose[0:2]))
        return (D[0] ** 10 + D[1] ** 10) ** R(1, 10) - r_dim

    def phi_func(q):  # :65-72
        return sd_2d_box(q[3:5], q[0:3])

    def p_func(x):  # :87-96
        pos = sp.Matrix(x[0:2])
        Rm = rot(x[2])
        out = []
        for c in cc:
            out += list(pos + Rm * sp.Matrix(c))
        return sp.Matrix(out)

    q0, q1 = sp.Matrix(th[0:5]), sp.Matrix(th[5:10])                     # :129-132
    u1 = sp.Matrix(th[10:12])
    h = th[12]
    q2 = sp.Matrix(z[0:5])                                               # :134-141
    gam, s1 = z[5], z[6]
    psi = z[7:12]
    b1 = sp.Matrix(z[12:21])
    spsi = z[21:26]
    sb1 = sp.Matrix(z[26:35])

    phi = phi_func(list(q2))
    N = sp.Matrix([phi]).jacobian(q2).T                                  # :143-144 (vec)
    # P_func :98-119
    P_block = p_func(list(q2)).jacobian(q2)
    N_pusher = N[3:5, 0]
    nrm = sp.sqrt(N_pusher[0] ** 2 + N_pusher[1] ** 2)
    n_dir = N_pusher / nrm
    t_dir = sp.Matrix([-n_dir[1], n_dir[0]])
    rr = sp.Matrix(q2[3:5]) - sp.Matrix(q2[0:2])
    m = rr[0] * t_dir[1] - rr[1] * t_dir[0]
    Prow = sp.Matrix([[t_dir[0], t_dir[1], m, -t_dir[0], -t_dir[1]]])
    P = P_block.col_join(Prow)                                           # 9 x 5
    vT = P * (q2 - q1) / h                                               # :148

    Mm = sp.diag(mass_block, mass_block, inertia, mass_pusher, mass_pusher)  # :51-52

    def M(q):
        return Mm

    def C(q, qd):  # :54-56
        return sp.zeros(5, 1)

    d, qm2, vm2 = midpoint_del(M, C, h, q0, q1, q2)                      # :150-156
    Bm = sp.Matrix([[0, 0], [0, 0], [0, 0], [1, 0], [0, 1]])             # :74-80
    dyn = d + Bm * u1 + N * gam + P.T * b1                               # :158-161
    r = list(dyn)
    r += [s1 - phi]                                                      # :166
    for i in range(4):                                                   # :168-174
        r += [psi[i] - mu_surface * mass_block * gravity * h * 0.25]
    r += [psi[4] - mu_pusher * gam]                                      # :176
    r += list(vT - sb1)                                                  # :178
    r += [gam * s1 - k]                                                  # :180
    for i in range(4):                                                   # :181-184
        r += list(cone_product([psi[i], b1[2 * i], b1[2 * i + 1]],
                               [spsi[i], sb1[2 * i], sb1[2 * i + 1]]) - sp.Matrix([k, 0, 0]))
    r += list(cone_product([psi[4], b1[8]], [spsi[4], sb1[8]]) - sp.Matrix([k, 0]))  # :185
    assert len(r) == 35
    # simulator.jl:19-50 (0-based)
    soc = []
    for i in range(4):
        soc.append(([7 + i, 12 + 2 * i, 13 + 2 * i], [21 + i, 26 + 2 * i, 27 + 2 * i]))
    soc.append(([11, 20], [25, 34]))
    socri = [[21, 22, 23], [24, 25, 26], [27, 28, 29], [30, 31, 32], [33, 34]]
    z_init = [("q", i) for i in range(5)] + [1.0, 1.0] + [1.0] * 5 + [0.1] * 9 + [1.0] * 5 + [0.1] * 9
    # static pivots: s1 row 5 -> z6; psi rows 6..10 -> z7..z11; vT rows 11..19 -> sb z26..z34;
    # bilinear row 20 -> gamma z5; cones: tail rows pivot on b (coef spsi), head on spsi.
    elim = [(5, 6)] + [(6 + i, 7 + i) for i in range(5)] + [(11 + i, 26 + i) for i in range(9)] + [(20, 5)]
    swaps = []
    for i in range(4):
        base = 21 + 3 * i
        elim += [(base + 1, 12 + 2 * i)]
        swaps += [((base, base + 1), (12 + 2 * i, 21 + i))]
    elim += [(34, 20)]
    swaps += [((33, 34), (20, 25))]
    # state program: per 3-d cone also the second tail row on b_2 and the (possibly exchanged) head row on s_psi, the 2-d
    # cone's head row likewise; what is left is the 5 x 5 configuration block, which keeps its partial pivoting
    elim_state = list(elim)
    for i in range(4):
        base = 21 + 3 * i
        elim_state += [(base + 2, 13 + 2 * i), (base, 21 + i)]
    elim_state += [(33, 25)]
    return ModelSpec(
        name="planar_push", model_id=4, nq=nq, nu=nu, nz=nz, nth=nth, z=z, th=th, kappa=k, r=r,
        ort=([5], [6]), soc=soc, equr=list(range(20)), ortr=[20], socri=socri, bil=list(range(20, 35)),
        z_init=z_init, kind="mech", nfric=0, idx_zq=list(range(5)), idx_gamma=[5], idx_b=list(range(12, 21)),
        elim=elim, swaps=swaps, elim_state=elim_state, state_tail_pivot=True,
        floor_pivots=[(20, 5)],
        opts=dict(IP_DEFAULT, kappa_tol=1e-4, kappa_grad_tol=1e-2),       # examples/planar_push.jl:21-22
    )


# ---------------------------------------------------------------------------------
# rocket (src/models/rocket/model.jl, codegen.jl)
# ---------------------------------------------------------------------------------
def _mrp_rotate(r, v):
    """Rotations.jl (pinned 1.0.2, Project.toml:32) `MRP(r...) * v`: the MRP is mapped to the
    unit quaternion (w, x) = ((1-|r|^2), 2 r)/(1+|r|^2) and the vector rotated actively:
    R = (w^2 - x'x) I + 2 x x' + 2 w [x]_x."""
    r = sp.Matrix(r)
    v = sp.Matrix(v)
    n2 = (r.T * r)[0, 0]
    w = (1 - n2) / (1 + n2)
    x = 2 * r / (1 + n2)
    return (w ** 2 - (x.T * x)[0, 0]) * v + 2 * x * (x.T * v)[0, 0] + 2 * w * x.cross(v)


def rocket_dynamics() -> ModelSpec:
    nx, nu = 12, 3
    nz, nth = 12, 16
    z, th, k = _syms("z", nz), _syms("th", nth), sp.Symbol("kappa", real=True)
    mass, length = 1.0, 1.0                                              # model.jl:37-38
    Ixx = 1.0 / 12.0 * mass * length ** 2.0
    inertia = sp.diag(Ixx, Ixx, 1.0e-5)                                  # :40
    inertia_inv = sp.diag(1.0 / Ixx, 1.0 / Ixx, 1.0 / 1.0e-5)            # :41
    grav = sp.Matrix([0, 0, -9.81])                                      # :47

    def f(zz, u):  # :14-33
        r_ = sp.Matrix(zz[3:6])
        v = sp.Matrix(zz[6:9])
        om = sp.Matrix(zz[9:12])
        Fb = sp.Matrix(u[0:3])
        tau = sp.Matrix([length * u[1], -length * u[0], 0])
        kin = F(0.25) * ((1 - (r_.T * r_)[0, 0]) * om - 2 * om.cross(r_) + 2 * (om.T * r_)[0, 0] * r_)
        acc = grav + (1.0 / mass) * _mrp_rotate(r_, Fb)
        dom = inertia_inv * (tau - om.cross(inertia * om))
        return sp.Matrix(list(v) + list(kin) + list(acc) + list(dom))

    y = sp.Matrix(z)                                                     # codegen.jl:15-22
    x = sp.Matrix(th[0:12])
    u = th[12:15]
    h = th[15]
    xm = (x + y) * F(0.5)
    r = list(y - (x + h * f(list(xm), u)))
    return ModelSpec(
        name="rocket_dynamics", model_id=5, nq=nx, nu=nu, nz=nz, nth=nth, z=z, th=th, kappa=k, r=r,
        ort=([], []), soc=[], equr=list(range(12)), ortr=[], socri=[], bil=[],   # simulator.jl:34-49
        z_init=[("q", i) for i in range(12)], kind="rocket", idx_zq=list(range(12)),
        # rz = I - (h/2) df/dx(mid) is diagonally dominant for the step sizes used -> static diagonal pivots
        elim=[(i, i) for i in range(12)],
        # src/models/rocket/dynamics.jl:21-27 (other fields RoboDojo defaults [RECALL])
        opts=dict(r_tol=1e-8, kappa_tol=1.0, max_iter=100, max_ls=25, eps_min=0.25,
                  kappa_reg=1e-3, gamma_reg=0.1, undercut=5.0, kappa_grad_tol=1.0),
    )


def rocket_projection() -> ModelSpec:
    nz, nth = 10, 4
    z, th, k = _syms("z", nz), _syms("th", nth), sp.Symbol("kappa", real=True)
    u = z[0:3]; p = z[3]; s = z[4]; w = z[5]; y = z[6]; v = z[7:10]      # codegen.jl:46-51
    ub = th[0:3]; uu = th[3]
    idx = [2, 0, 1]
    r = [u[0] - ub[0] - v[0], u[1] - ub[1] - v[1], u[2] - ub[2] - v[2] - (y + p),   # :57
         uu - u[2] - s,                                                               # :58
         -y - w,                                                                      # :59
         w * s - k,                                                                   # :60
         p * u[2] - k]                                                                # :61
    r += list(cone_product([u[i] for i in idx], [v[i] for i in idx]) - sp.Matrix([k, 0, 0]))  # :62
    return ModelSpec(
        name="rocket_projection", model_id=6, nq=3, nu=0, nz=nz, nth=nth, z=z, th=th, kappa=k, r=r,
        # src/models/rocket/dynamics.jl:52-63 (0-based): ort primal {z4=s, z2=u3}, dual {z5=w, z3=p}
        ort=([4, 2], [5, 3]), soc=[([2, 0, 1], [9, 7, 8])],
        equr=[0, 1, 2, 3, 4], ortr=[5, 6], socri=[[7, 8, 9]], bil=[5, 6, 7, 8, 9],
        # dynamics.jl:169-172: z .= 0.1; z[3] += 1; z[10] += 1; z[7] = 0 (1-based)
        z_init=[0.1, 0.1, 1.1, 0.1, 0.1, 0.1, 0.0, 0.1, 0.1, 1.1],
        kind="proj", idx_zq=[0, 1, 2],
        # pivots: row3 (uu-u3-s) on s=z4 [coef -1]; row4 (-y-w) on w=z5 [-1]; rows0,1 on v1,v2 (z7,z8) [-1];
        # row2 on v3 (z9) [-1]; remaining handled by the runtime-pivoted dense tail.  (Tried in round 3: two more static pivots, the
        # slack s for row5 and u3 for row6, leaving a 3 x 3 tail.  With eps_min = 0 the step goes all the way to the boundary
        # (tau = 1), s or u3 become exactly 0 on some solves and the static pivot divides by it -- the pivoted 5 x 5 survives that.)
        elim=[(3, 4), (4, 5), (0, 7), (1, 8), (2, 9)],
        # dynamics.jl:77-86
        opts=dict(r_tol=1e-8, kappa_tol=1e-4, max_iter=100, max_ls=25, eps_min=0.0,
                  kappa_reg=0.0, gamma_reg=0.0, undercut=float("inf"), kappa_grad_tol=1e-4),
    )


# ---------------------------------------------------------------------------------
# hopper (RoboDojo.jl src/robots/hopper -- NOT in /root/reference; restated from the
# structural pins in examples/hopper.jl:14,38-50,178,270 and
# examples/comparisons/hopper.jl:6-37,57-162 plus recalled RoboDojo source. UNVERIFIED.)
# ---------------------------------------------------------------------------------
HOPPER_PARAMS = dict(mass_body=3.0, mass_foot=1.0, inertia_body=0.75, inertia_foot=0.25,
                     body_radius=0.1, foot_radius=0.05, leg_len_max=1.0, leg_len_min=0.25,
                     friction_body_world=0.5, friction_foot_world=0.5, gravity=9.81)


def hopper() -> ModelSpec:
    P = HOPPER_PARAMS
    nq, nu, nc = 4, 2, 4
    nz, nth = 20, 13
    z, th, k = _syms("z", nz), _syms("th", nth), sp.Symbol("kappa", real=True)
    mb, mf, Jb, Jf = P["mass_body"], P["mass_foot"], P["inertia_body"], P["inertia_foot"]
    g = P["gravity"]

    def kin_foot(q):  # [RECALL] kinematics_foot
        return sp.Matrix([q[0] + q[3] * sp.sin(q[2]), q[1] - q[3] * sp.cos(q[2])])

    def kin_foot_jac(q):
        return sp.Matrix([[1, 0, q[3] * sp.cos(q[2]), sp.sin(q[2])],
                          [0, 1, q[3] * sp.sin(q[2]), -sp.cos(q[2])]])

    qs, qds = _syms("q_", 4), _syms("qd_", 4)
    vb = sp.Matrix(qds[0:2])
    vf = kin_foot_jac(qs) * sp.Matrix(qds)
    L = (F(0.5) * mb * (vb.T * vb)[0, 0] + F(0.5) * Jb * qds[2] ** 2 - mb * g * qs[1]
         + F(0.5) * mf * (vf.T * vf)[0, 0] + F(0.5) * Jf * qds[2] ** 2 - mf * g * kin_foot(qs)[1])
    # RoboDojo codegen_dynamics [RECALL]: M = d2L/dqd2, C = (d2L/dqd dq) qd - dL/dq
    dLdqd = sp.Matrix([L]).jacobian(qds).T
    Msym = sp.simplify(dLdqd.jacobian(qds))
    Csym = sp.simplify(dLdqd.jacobian(qs) * sp.Matrix(qds) - sp.Matrix([L]).jacobian(qs).T)

    def M(q):
        return Msym.subs(dict(zip(qs, q)), simultaneous=True)

    def C(q, qd):
        return Csym.subs(dict(zip(qs + qds, list(q) + list(qd))), simultaneous=True)

    q0, q1 = sp.Matrix(th[0:4]), sp.Matrix(th[4:8])
    u1 = sp.Matrix(th[8:10])
    mu_body, mu_foot, h = th[10], th[11], th[12]
    q2 = sp.Matrix(z[0:4])
    gam = z[4:8]; sg = z[8:12]; psi = z[12:14]; b1 = z[14:16]; spsi = z[16:18]; sb1 = z[18:20]

    # signed distances: body-ground, foot-ground, leg min, leg max (comparisons/hopper.jl:74,236-237)
    phi = sp.Matrix([q2[1] - P["body_radius"], kin_foot(q2)[1] - P["foot_radius"],
                     q2[3] - P["leg_len_min"], P["leg_len_max"] - q2[3]])
    # contact jacobian rows: body (x,z), foot (x,z), leg limits (comparisons/hopper.jl:25-30)
    J = sp.Matrix([[1, 0, 0, 0], [0, 1, 0, 0]]).col_join(kin_foot_jac(q2)).col_join(
        sp.Matrix([[0, 0, 0, 1], [0, 0, 0, -1]]))
    lam = J.T * sp.Matrix([b1[0], gam[0], b1[1], gam[1], gam[2], gam[3]])
    lam[2] += P["body_radius"] * b1[0]                                   # comparisons/hopper.jl:30
    d, qm2, vm2 = midpoint_del(M, C, h, q0, q1, q2)
    Bq = sp.Matrix([[0, 0, 1, 0], [-sp.sin(qm2[2]), sp.cos(qm2[2]), 0, 1]])   # input_jacobian [RECALL]
    dyn = d + Bq.T * u1 + lam
    v = (q2 - q1) / h
    vT_body = v[0] + P["body_radius"] * v[2]                             # comparisons/hopper.jl:152-155
    vT_foot = (kin_foot_jac(q2) * v)[0]
    r = list(dyn)
    r += list(sp.Matrix(sg) - phi)
    r += [psi[0] - mu_body * gam[0], psi[1] - mu_foot * gam[1]]
    r += [vT_body - sb1[0], vT_foot - sb1[1]]
    r += [gam[i] * sg[i] - k for i in range(4)]
    r += list(cone_product([psi[0], b1[0]], [spsi[0], sb1[0]]) - sp.Matrix([k, 0]))
    r += list(cone_product([psi[1], b1[1]], [spsi[1], sb1[1]]) - sp.Matrix([k, 0]))
    assert len(r) == 20
    z_init = [("q", i) for i in range(4)] + [1.0] * 4 + [1.0] * 4 + [1.0] * 2 + [0.1] * 2 + [1.0] * 2 + [0.1] * 2
    elim = ([(4 + i, 8 + i) for i in range(4)]        # slack rows -> s_gamma
            + [(8, 12), (9, 13)]                       # psi rows -> psi
            + [(10, 18), (11, 19)]                     # tangential velocity rows -> sb
            + [(12 + i, 4 + i) for i in range(4)]      # bilinear rows -> gamma (pivot s_gamma)
            + [(17, 14), (19, 15)])                   # cones: tail row -> b (after the runtime role swap)
    return ModelSpec(
        name="hopper", model_id=7, nq=nq, nu=nu, nz=nz, nth=nth, z=z, th=th, kappa=k, r=r,
        ort=([4, 5, 6, 7], [8, 9, 10, 11]), soc=[([12, 14], [16, 18]), ([13, 15], [17, 19])],
        equr=list(range(12)), ortr=[12, 13, 14, 15], socri=[[16, 17], [18, 19]], bil=list(range(12, 20)),
        z_init=z_init, kind="mech", nfric=2,
        fric_default=[P["friction_body_world"], P["friction_foot_world"]],
        idx_zq=[0, 1, 2, 3], idx_gamma=[4, 5, 6, 7], idx_b=[14, 15],
        elim=elim, floor_pivots=[(12 + i, 4 + i) for i in range(4)],
        swaps=[((16, 17), (14, 16)), ((18, 19), (15, 17))],
        # measured (CPU build, 25 600 rollout knots + 12 288 knots): without any runtime pivoting in the iterations the
        # iteration counts are identical and the states agree to 3e-14
        static_tail=True, elim_state=elim + [(16, 16), (18, 17)], state_tail_pivot=False,
        opts=dict(IP_DEFAULT, kappa_tol=1e-4, kappa_grad_tol=1e-3),       # examples/hopper.jl:42
        notes="RoboDojo hopper, restated from recall; constants unverified",
    )


ALL_MODELS: Dict[str, Callable[[], ModelSpec]] = {
    "acrobot_impact": acrobot_impact,
    "acrobot_nominal": acrobot_nominal,
    "cartpole_friction": cartpole_friction,
    "cartpole_frictionless": cartpole_frictionless,
    "planar_push": planar_push,
    "rocket_dynamics": rocket_dynamics,
    "rocket_projection": rocket_projection,
    "hopper": hopper,
}


# ---------------------------------------------------------------------------------
# user models (python -m optimization_dynamics_amd.codegen --add spec.py)
# ---------------------------------------------------------------------------------
def _load_spec_module(path):
    import importlib.util
    spec = importlib.util.spec_from_file_location("od_user_spec_" + os.path.splitext(os.path.basename(path))[0], path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if not hasattr(mod, "spec"):
        raise ValueError("%s does not define spec() -> ModelSpec" % path)
    return mod


def _registry(udir):
    p = os.path.join(udir, "registry.json")
    return json.load(open(p)) if os.path.exists(p) else {"models": []}


def register_user_model(path, udir):
    """copy the spec file into the package and give its model the next free id; returns the model name"""
    import shutil
    mod = _load_spec_module(path)
    m = mod.spec()
    if not isinstance(m, ModelSpec):
        raise ValueError("spec() of %s does not return a ModelSpec" % path)
    if m.name in ALL_MODELS:
        raise ValueError("model name %r is one of the built-in models" % m.name)
    if m.kind != "mech":
        raise ValueError("user models are mechanical models (theta = [q0; q1; u; friction; h])")
    os.makedirs(udir, exist_ok=True)
    reg = _registry(udir)
    names = [e["name"] for e in reg["models"]]
    if m.name not in names:
        reg["models"].append({"name": m.name, "file": m.name + ".py"})
    dst = os.path.join(udir, m.name + ".py")
    if os.path.abspath(path) != os.path.abspath(dst):
        shutil.copyfile(path, dst)
    json.dump(reg, open(os.path.join(udir, "registry.json"), "w"), indent=1)
    return m.name


def user_model_id(name, udir):
    names = [e["name"] for e in _registry(udir)["models"]]
    return len(ALL_MODELS) + names.index(name)


def all_models(udir=None) -> Dict[str, Callable[[], ModelSpec]]:
    """built-in models plus the registered user models (ids continue after the built-in ones)"""
    udir = udir or os.path.join(os.path.dirname(os.path.abspath(__file__)), "user_models")
    out = dict(ALL_MODELS)
    for k, e in enumerate(_registry(udir)["models"]):
        def make(e=e, k=k):
            m = _load_spec_module(os.path.join(udir, e["file"])).spec()
            m.model_id = len(ALL_MODELS) + k
            return m
        out[e["name"]] = make
    return out
