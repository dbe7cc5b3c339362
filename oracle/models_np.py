"""Hand-written numpy restatement of the reference residuals -- TEST INFRASTRUCTURE ONLY.

Purpose: pin the *generated* residual/Jacobian code (oracle/gen/*.h and
optimization_dynamics_amd/csrc/gen/*.h, both emitted from codegen/models.py) against an
independent, line-by-line numeric transliteration of the Julia sources.  Jacobians are checked
against complex-step / finite differences of these functions (tests/test_models.py).

Each function cites the reference lines it follows (paths relative to /root/reference).
All functions accept real or complex arrays (complex-step differentiation).
"""
import numpy as np


def cone_product(a, b):
    # RoboDojo.cone_product as pinned by src/models/cartpole/model.jl:111-112
    a = np.asarray(a)
    b = np.asarray(b)
    return np.concatenate([[a @ b], a[0] * b[1:] + b[0] * a[1:]])


def lagrangian_derivatives(M, C, q, v):
    # [RECALL] RoboDojo: D1L = -C(q,v), D2L = M(q) v  (consistent with acrobot/model.jl:97-100)
    return -C(q, v), M(q) @ v


def _del(M, C, h, q0, q1, q2):
    # src/models/acrobot/model.jl:90-100
    qm1 = 0.5 * (q0 + q1)
    vm1 = (q1 - q0) / h
    qm2 = 0.5 * (q1 + q2)
    vm2 = (q2 - q1) / h
    D1L1, D2L1 = lagrangian_derivatives(M, C, qm1, vm1)
    D1L2, D2L2 = lagrangian_derivatives(M, C, qm2, vm2)
    return 0.5 * h * D1L1 + D2L1 + 0.5 * h * D1L2 - D2L2, qm2, vm2


# ------------------------------------------------------------------ acrobot
def _acrobot_MC():
    m1, J1, l1, lc1, m2, J2, l2, lc2, g = 1.0, 0.333, 1.0, 0.5, 1.0, 0.333, 1.0, 0.5, 9.81  # model.jl:159-160

    def M(x):  # :41-51
        a = J1 + J2 + m2 * l1 * l1 + 2.0 * m2 * l1 * lc2 * np.cos(x[1])
        b = J2 + m2 * l1 * lc2 * np.cos(x[1])
        return np.array([[a, b], [b, J2 + 0 * b]])

    def tau(x):  # :53-61
        a = -1.0 * m1 * g * lc1 * np.sin(x[0]) - m2 * g * (l1 * np.sin(x[0]) + lc2 * np.sin(x[0] + x[1]))
        b = -1.0 * m2 * g * lc2 * np.sin(x[0] + x[1])
        return np.array([a, b])

    def c(q, qd):  # :63-71
        a = -2.0 * m2 * l1 * lc2 * np.sin(q[1]) * qd[1]
        b = -1.0 * m2 * l1 * lc2 * np.sin(q[1]) * qd[1]
        cc = m2 * l1 * lc2 * np.sin(q[1]) * qd[0]
        return np.array([[a, b], [cc, 0 * cc]])

    def C(q, qd):  # :77-79
        return c(q, qd) @ qd - tau(q)

    return M, C


def acrobot_impact(z, th, kappa):
    M, C = _acrobot_MC()
    q0, q1, u1, h = th[0:2], th[2:4], th[4], th[5]          # model.jl:126-129
    q2, lam, s = z[0:2], z[2:4], z[4:6]                     # :131-133
    d, qm2, vm2 = _del(M, C, h, q0, q1, q2)
    phi = np.array([0.5 * np.pi - q2[1], q2[1] + 0.5 * np.pi])   # :81-83
    P = np.array([[0.0, -1.0], [0.0, 1.0]])                 # jacobian of phi, :85-88
    dyn = d + np.array([0.0, 1.0]) * u1 + P.T @ lam - h * 0.5 * vm2   # :100-103
    return np.concatenate([dyn, s - phi, lam * s - kappa])  # :135-141


def acrobot_nominal(z, th, kappa):
    M, C = _acrobot_MC()
    q0, q1, u1, h = th[0:2], th[2:4], th[4], th[5]
    q2 = z[0:2]
    d, qm2, vm2 = _del(M, C, h, q0, q1, q2)
    return d + np.array([0.0, 1.0]) * u1 - h * 0.5 * vm2    # model.jl:106-119


# ------------------------------------------------------------------ cartpole
def _cartpole_MC():
    mc, mp, l, g = 1.0, 0.2, 0.5, 9.81                      # model.jl:132

    def M(x):  # :28-32
        return np.array([[mc + mp + 0 * x[1], mp * l * np.cos(x[1])], [mp * l * np.cos(x[1]), mp * l ** 2.0 + 0 * x[1]]])

    def C(q, qd):  # :43-49
        Cm = np.array([[0.0 * qd[1], -1.0 * mp * qd[1] * l * np.sin(q[1])], [0.0 * qd[1], 0.0 * qd[1]]])
        G = np.array([0.0 * q[1], mp * g * l * np.sin(q[1])])
        return -Cm @ qd + G

    return M, C, (mc, mp, l, g)


def cartpole_friction(z, th, kappa):
    M, C, (mc, mp, l, g) = _cartpole_MC()
    q0, q1, u1 = th[0:2], th[2:4], th[4]                    # model.jl:86-91
    mu_s, mu_a, h = th[5], th[6], th[7]
    q2, psi, b, spsi, sb = z[0:2], z[2:4], z[4:6], z[6:8], z[8:10]   # :93-97
    vT1 = (q2[0] - q1[0]) / h
    vT2 = (q2[1] - q1[1]) / h
    d, _, _ = _del(M, C, h, q0, q1, q2)
    dyn = d + np.array([1.0, 0.0]) * u1 + b                 # :51-64
    return np.concatenate([
        dyn,
        [sb[0] - vT1, psi[0] - mu_s * (mp + mc) * g * h, sb[1] - vT2, psi[1] - mu_a * (mp * g * l) * h],  # :107-110
        cone_product([psi[0], b[0]], [spsi[0], sb[0]]) - np.array([kappa, 0.0]),   # :111
        cone_product([psi[1], b[1]], [spsi[1], sb[1]]) - np.array([kappa, 0.0]),   # :112
    ])


def cartpole_frictionless(z, th, kappa):
    M, C, _ = _cartpole_MC()
    q0, q1, u1, h = th[0:2], th[2:4], th[4], th[5]
    d, _, _ = _del(M, C, h, q0, q1, z[0:2])
    return d + np.array([1.0, 0.0]) * u1                    # model.jl:66-79,116-129


# ------------------------------------------------------------------ planar push
_R_DIM = 0.1


def _rot(x):
    return np.array([[np.cos(x), -np.sin(x)], [np.sin(x), np.cos(x)]])


def _pp_phi(q):
    # sd_2d_box, src/models/planar_push/model.jl:26-31
    D = _rot(-q[2]) @ (q[3:5] - q[0:2])
    return (D[0] ** 10 + D[1] ** 10) ** (1 / 10) - _R_DIM


def _pp_pfunc(q):
    # p_func :87-96
    cc = [np.array([_R_DIM, _R_DIM]), np.array([-_R_DIM, _R_DIM]), np.array([_R_DIM, -_R_DIM]), np.array([-_R_DIM, -_R_DIM])]
    Rm = _rot(q[2])
    return np.concatenate([q[0:2] + Rm @ c for c in cc])


def _cs_jac(fun, x, m):
    """complex-step Jacobian of fun: R^n -> R^m at real x"""
    n = len(x)
    J = np.zeros((m, n))
    for j in range(n):
        xc = np.array(x, dtype=complex)
        xc[j] += 1e-30j
        J[:, j] = np.imag(np.atleast_1d(fun(xc))) / 1e-30
    return J


def planar_push(z, th, kappa):
    """real inputs only (inner Jacobians N, P are themselves complex-step derivatives)"""
    mu_surface, mu_pusher, gravity, mass_block, mass_pusher = 0.5, 0.5, 9.81, 1.0, 10.0   # model.jl:43-47
    inertia = 1.0 / 12.0 * mass_block * ((2.0 * _R_DIM) ** 2 + (2.0 * _R_DIM) ** 2)
    q0, q1, u1, h = th[0:5], th[5:10], th[10:12], th[12]    # :129-132
    q2 = z[0:5]
    gam, s1 = z[5], z[6]
    psi, b1, spsi, sb1 = z[7:12], z[12:21], z[21:26], z[26:35]   # :137-141
    phi = _pp_phi(q2)
    N = _cs_jac(_pp_phi, q2, 1)[0]                          # :143-144
    P_block = _cs_jac(_pp_pfunc, q2, 8)                     # :99-100
    Np = N[3:5]
    n_dir = Np / np.sqrt(Np[0] ** 2.0 + Np[1] ** 2.0)       # :110
    t_dir = np.array([-n_dir[1], n_dir[0]])
    rr = q2[3:5] - q2[0:2]
    m = rr[0] * t_dir[1] - rr[1] * t_dir[0]
    P = np.vstack([P_block, [t_dir[0], t_dir[1], m, -t_dir[0], -t_dir[1]]])   # :116-118
    vT = P @ (q2 - q1) / h                                  # :148
    Mm = np.diag([mass_block, mass_block, inertia, mass_pusher, mass_pusher])
    d, _, _ = _del(lambda q: Mm, lambda q, v: np.zeros(5), h, q0, q1, q2)
    Bm = np.array([[0, 0], [0, 0], [0, 0], [1.0, 0], [0, 1.0]])
    dyn = d + Bm @ u1 + N * gam + P.T @ b1                  # :158-161
    out = [dyn, [s1 - phi]]
    out.append([psi[i] - mu_surface * mass_block * gravity * h * 0.25 for i in range(4)])   # :168-174
    out.append([psi[4] - mu_pusher * gam])                  # :176
    out.append(vT - sb1)                                    # :178
    out.append([gam * s1 - kappa])                          # :180
    for i in range(4):                                      # :181-184
        out.append(cone_product([psi[i], b1[2 * i], b1[2 * i + 1]], [spsi[i], sb1[2 * i], sb1[2 * i + 1]]) - np.array([kappa, 0, 0]))
    out.append(cone_product([psi[4], b1[8]], [spsi[4], sb1[8]]) - np.array([kappa, 0]))   # :185
    return np.concatenate([np.atleast_1d(o) for o in out])


# ------------------------------------------------------------------ rocket
def _mrp_matrix(r):
    """Rotations.jl 1.0.2 MRP -> rotation matrix via the unit quaternion
    (w, v) = ((1 - |r|^2), 2 r) / (1 + |r|^2), written out entry-wise (independent of the
    vector form used in codegen/models.py)."""
    n2 = r[0] ** 2 + r[1] ** 2 + r[2] ** 2
    w = (1 - n2) / (1 + n2)
    x, y, zz = 2 * r[0] / (1 + n2), 2 * r[1] / (1 + n2), 2 * r[2] / (1 + n2)
    return np.array([
        [1 - 2 * (y * y + zz * zz), 2 * (x * y - w * zz), 2 * (x * zz + w * y)],
        [2 * (x * y + w * zz), 1 - 2 * (x * x + zz * zz), 2 * (y * zz - w * x)],
        [2 * (x * zz - w * y), 2 * (y * zz + w * x), 1 - 2 * (x * x + y * y)],
    ])


def rocket_f(zz, u):
    # src/models/rocket/model.jl:14-33
    mass, length = 1.0, 1.0
    Ixx = 1.0 / 12.0 * mass * length ** 2.0
    inertia = np.array([Ixx, Ixx, 1.0e-5])
    inertia_inv = np.array([1.0 / Ixx, 1.0 / Ixx, 1.0 / 1.0e-5])
    grav = np.array([0.0, 0.0, -9.81])
    r, v, om = zz[3:6], zz[6:9], zz[9:12]
    Fb = u[0:3]
    tau = np.array([length * u[1], -length * u[0], 0.0 * u[0]])
    kin = 0.25 * ((1.0 - r @ r) * om - 2.0 * np.cross(om, r) + 2.0 * (om @ r) * r)
    acc = grav + (1.0 / mass) * (_mrp_matrix(r) @ Fb)
    dom = inertia_inv * (tau - np.cross(om, inertia * om))
    return np.concatenate([v, kin, acc, dom])


def rocket_dynamics(z, th, kappa):
    # src/models/rocket/codegen.jl:14-23
    x, u, h = th[0:12], th[12:15], th[15]
    return z - (x + h * rocket_f(0.5 * (x + z), u))


def rocket_projection(z, th, kappa):
    # src/models/rocket/codegen.jl:45-64
    u, p, s, w, y, v = z[0:3], z[3], z[4], z[5], z[6], z[7:10]
    ub, uu = th[0:3], th[3]
    idx = [2, 0, 1]
    return np.concatenate([
        u - ub - v - np.array([0.0, 0.0, 1.0]) * (y + p),
        [uu - u[2] - s, -y - w, w * s - kappa, p * u[2] - kappa],
        cone_product(u[idx], v[idx]) - np.array([kappa, 0.0, 0.0]),
    ])


# ------------------------------------------------------------------ hopper (RoboDojo, recalled)
HOPPER = dict(mass_body=3.0, mass_foot=1.0, inertia_body=0.75, inertia_foot=0.25, body_radius=0.1,
              foot_radius=0.05, leg_len_max=1.0, leg_len_min=0.25, gravity=9.81)


def _hopper_MC():
    """Mass matrix and bias derived BY HAND from the recalled RoboDojo Lagrangian
    L = 1/2 mb |v_b|^2 + 1/2 Jb w^2 - mb g z + 1/2 mf |J_f(q) qd|^2 + 1/2 Jf w^2 - mf g (z - r cos t)
    (independent of the sympy derivation in codegen/models.py)."""
    P = HOPPER
    mb, mf, Jb, Jf, g = P["mass_body"], P["mass_foot"], P["inertia_body"], P["inertia_foot"], P["gravity"]

    def Jf_(q):
        return np.array([[1.0, 0.0, q[3] * np.cos(q[2]), np.sin(q[2])],
                         [0.0, 1.0, q[3] * np.sin(q[2]), -np.cos(q[2])]])

    def M(q):
        J = Jf_(q)
        Mb = np.diag([mb, mb, Jb + Jf, 0.0]).astype(J.dtype)
        return Mb + mf * J.T @ J

    def C(q, qd):
        # C = (d(M qd)/dq) qd - dL/dq ;  only q3=t and q4=r enter M.
        t, r = q[2], q[3]
        J = Jf_(q)
        dJdt = np.array([[0, 0, -r * np.sin(t), np.cos(t)], [0, 0, r * np.cos(t), np.sin(t)]])
        dJdr = np.array([[0, 0, np.cos(t), 0 * t], [0, 0, np.sin(t), 0 * t]])
        dMdt = mf * (dJdt.T @ J + J.T @ dJdt)
        dMdr = mf * (dJdr.T @ J + J.T @ dJdr)
        Mdot_qd = (dMdt @ qd) * qd[2] + (dMdr @ qd) * qd[3]
        # dL/dq: kinetic part 1/2 qd' dM/dq_i qd, potential part
        dL = np.zeros(4, dtype=Mdot_qd.dtype)
        dL[1] = -(mb + mf) * g
        dL[2] = 0.5 * qd @ dMdt @ qd - mf * g * r * np.sin(t)
        dL[3] = 0.5 * qd @ dMdr @ qd + mf * g * np.cos(t)
        return Mdot_qd - dL

    return M, C, Jf_


def hopper(z, th, kappa):
    P = HOPPER
    M, C, Jf_ = _hopper_MC()
    q0, q1, u1 = th[0:4], th[4:8], th[8:10]
    mu_b, mu_f, h = th[10], th[11], th[12]
    q2 = z[0:4]
    gam, sg, psi, b1, spsi, sb1 = z[4:8], z[8:12], z[12:14], z[14:16], z[16:18], z[18:20]
    kf = np.array([q2[0] + q2[3] * np.sin(q2[2]), q2[1] - q2[3] * np.cos(q2[2])])
    phi = np.array([q2[1] - P["body_radius"], kf[1] - P["foot_radius"], q2[3] - P["leg_len_min"], P["leg_len_max"] - q2[3]])
    J = np.vstack([np.array([[1.0, 0, 0, 0], [0, 1.0, 0, 0]]), Jf_(q2), np.array([[0, 0, 0, 1.0], [0, 0, 0, -1.0]])])
    lam = J.T @ np.array([b1[0], gam[0], b1[1], gam[1], gam[2], gam[3]])   # comparisons/hopper.jl:25-29
    lam[2] = lam[2] + P["body_radius"] * b1[0]                                # :30
    d, qm2, vm2 = _del(M, C, h, q0, q1, q2)
    Bq = np.array([[0, 0, 1.0, 0], [-np.sin(qm2[2]), np.cos(qm2[2]), 0, 1.0]])
    dyn = d + Bq.T @ u1 + lam
    v = (q2 - q1) / h
    vT_body = v[0] + P["body_radius"] * v[2]                                  # comparisons/hopper.jl:152-155
    vT_foot = (Jf_(q2) @ v)[0]
    return np.concatenate([
        dyn, sg - phi,
        [psi[0] - mu_b * gam[0], psi[1] - mu_f * gam[1]],
        [vT_body - sb1[0], vT_foot - sb1[1]],
        gam * sg - kappa,
        cone_product([psi[0], b1[0]], [spsi[0], sb1[0]]) - np.array([kappa, 0.0]),
        cone_product([psi[1], b1[1]], [spsi[1], sb1[1]]) - np.array([kappa, 0.0]),
    ])


RESIDUALS = {
    "acrobot_impact": acrobot_impact, "acrobot_nominal": acrobot_nominal,
    "cartpole_friction": cartpole_friction, "cartpole_frictionless": cartpole_frictionless,
    "planar_push": planar_push, "rocket_dynamics": rocket_dynamics,
    "rocket_projection": rocket_projection, "hopper": hopper,
}
COMPLEX_OK = {k: k != "planar_push" for k in RESIDUALS}
