"""Tables + glue for the general cooperative solver (8 lanes per problem, cones of dimension 2 and 3), csrc/od_coop3.h.

Same contract as codegen/coop.py: the block structure every contact-implicit model of the reference has
(src/models/*/model.jl: `residual`) is read off the symbolic Jacobian and CHECKED entry by entry; a model that does not
have it is rejected (returns None) and keeps the lane-per-problem kernels.  No arithmetic is generated here: residual
and Jacobian values come from the model's own ``eval_r`` / ``eval_rz`` (gen/<model>.h); this file emits where each
value goes.  A friction cone of dimension 3 (planar push, src/models/planar_push/model.jl:181-184) brings two friction
unknowns b1, b2 with their slacks, two tangential-velocity rows and two tail rows into its lane.
"""
from __future__ import annotations

import io
from typing import Optional

import sympy as sp

from .coop import NotCoop, _need
from .emit import Derived, _DevicePrinter
from .models import ModelSpec


def _dpp_block8(ops, indent="    ", max_operands=28):
    """C++ for a list of cross-lane operations between the 8 lanes of a problem (two problems per DPP row).  Device: one
    inline-asm block per chunk -- `s_nop 1` (DPP read-after-VALU-write is not interlocked), every operation for lanes
    0..7 (row_newbcast:L, bank_mask 0x3), `s_nop 0`, every operation for lanes 8..15 (row_newbcast:L+8, bank_mask 0xc):
    a 0xc fmac directly after a 0x3 fmac on the same accumulator reads a stale value (tools/ubench/dpp_bank_mask3.hip).
    Host test build: the Row8Emu calls.
    ops: ("fmac" | "fnmac", lane, acc, src, mul)   acc (+)= src[lane] * (+-)mul
         ("bc", lane, dst, src)                      dst = src[lane]"""
    out = []
    chunks, cur, cur_syms = [], [], set()
    for op in ops:
        syms = set(op[2:])
        if cur and len(cur_syms | syms) > max_operands:
            chunks.append(cur)
            cur, cur_syms = [], set()
        cur.append(op)
        cur_syms |= syms
    if cur:
        chunks.append(cur)
    out.append("#if defined(__HIP_DEVICE_COMPILE__)")
    for ch in chunks:
        outs, ins = [], []
        for op in ch:
            if op[2] not in outs:
                outs.append(op[2])
        for op in ch:
            for e in (op[3:] if op[0] != "bc" else op[3:4]):
                if e not in ins and e not in outs:
                    ins.append(e)
        num = {e: k for k, e in enumerate(outs + ins)}
        lines = ["s_nop 1"]
        for half, mask in ((0, "0x3"), (8, "0xc")):
            if half:
                lines.append("s_nop 0")
            for op in ch:
                if op[0] == "bc":
                    lines.append("v_mov_b64_dpp %%%d, %%%d row_newbcast:%d row_mask:0xf bank_mask:%s" % (num[op[2]], num[op[3]], op[1] + half, mask))
                else:
                    neg = "-" if op[0] == "fnmac" else ""
                    lines.append("v_fmac_f64_dpp %%%d, %%%d, %s%%%d row_newbcast:%d row_mask:0xf bank_mask:%s"
                                 % (num[op[2]], num[op[3]], neg, num[op[4]], op[1] + half, mask))
        is_bc = {op[2] for op in ch if op[0] == "bc"}
        cons_out = ", ".join(('"=&v"(%s)' if e in is_bc else '"+v"(%s)') % e for e in outs)
        cons_in = ", ".join('"v"(%s)' % e for e in ins)
        out.append(indent + 'asm("' + ('\\n\\t"\n' + indent + '    "').join(lines) + '"')
        out.append(indent + "    : %s : %s);" % (cons_out, cons_in))
    out.append("#else")
    for op in ops:
        if op[0] == "bc":
            out.append(indent + "%s = RO::template bc<%d>(%s);" % (op[2], op[1], op[3]))
        else:
            out.append(indent + "RO::template %s<%d>(%s, %s, %s);" % (op[0], op[1], op[2], op[3], op[4]))
    out.append("#endif")
    return "\n".join(out) + "\n"


def emit_coop3(m: ModelSpec, d: Derived) -> Optional[str]:
    try:
        return _emit(m, d)
    except NotCoop as e:
        print("  (no 8-lane cooperative kernel for %s: %s)" % (m.name, e))
        return None


def _emit(m: ModelSpec, d: Derived) -> str:
    _need(m.kind == "mech", "not a mechanical model")
    nq, nz = m.nq, m.nz
    z, rz, r0 = m.z, d.rz, list(d.r0)
    zq = list(m.idx_zq)
    _need(len(zq) == nq, "idx_zq")
    piv_of_col = {c: r for (r, c) in m.elim}
    NC = len(m.ort[0])
    NK = len(m.soc)
    _need(NC + NK > 0, "no cones")
    _need(NC + NK <= 8, "more than 8 contact / cone blocks")
    dims = []
    for p, dd in m.soc:
        _need(len(p) == len(dd) and len(p) in (2, 3), "cone dimension not 2 or 3")
        dims.append(len(p))
    DIM3 = any(k == 3 for k in dims)
    ZG, ZS = list(m.ort[0]), list(m.ort[1])
    RBIL = list(m.ortr)
    RSL = []
    for i in range(NC):
        _need(ZS[i] in piv_of_col, "slack row of contact %d" % i)
        RSL.append(piv_of_col[ZS[i]])
    ZPSI = [p[0] for p, _ in m.soc]
    ZB1 = [p[1] for p, _ in m.soc]
    ZB2 = [p[2] if len(p) == 3 else -1 for p, _ in m.soc]
    ZSPSI = [dd[0] for _, dd in m.soc]
    ZSB1 = [dd[1] for _, dd in m.soc]
    ZSB2 = [dd[2] if len(dd) == 3 else -1 for _, dd in m.soc]
    RCA = [rr[0] for rr in m.socri]
    RCB1 = [rr[1] for rr in m.socri]
    RCB2 = [rr[2] if len(rr) == 3 else -1 for rr in m.socri]
    RPSI, RV1, RV2 = [], [], []
    for c in range(NK):
        _need(ZPSI[c] in piv_of_col and ZSB1[c] in piv_of_col, "psi / velocity row of cone %d" % c)
        RPSI.append(piv_of_col[ZPSI[c]])
        RV1.append(piv_of_col[ZSB1[c]])
        if dims[c] == 3:
            _need(ZSB2[c] in piv_of_col, "second velocity row of cone %d" % c)
            RV2.append(piv_of_col[ZSB2[c]])
        else:
            RV2.append(-1)
    used = set(RSL + RBIL + RPSI + RV1 + RCA + RCB1) | {r for r in RV2 + RCB2 if r >= 0}
    RDYN = [i for i in range(nz) if i not in used]
    _need(len(RDYN) == nq, "dynamics rows")
    allz = zq + ZG + ZS + ZPSI + ZB1 + ZSPSI + ZSB1 + [k for k in ZB2 + ZSB2 if k >= 0]
    _need(sorted(allz) == list(range(nz)), "variables do not partition")
    qcols = set(zq)

    def nzcols(row):
        return {j for j in range(nz) if rz[row, j] != 0}

    # ---- structure checks -------------------------------------------------------------------------------------
    for i in range(NC):
        _need(rz[RSL[i], ZS[i]] == 1, "slack pivot")
        _need(nzcols(RSL[i]) <= qcols | {ZS[i]}, "slack row pattern")
        _need(rz[RBIL[i], ZG[i]] == z[ZS[i]] and rz[RBIL[i], ZS[i]] == z[ZG[i]], "bilinear row")
        _need(nzcols(RBIL[i]) == {ZG[i], ZS[i]}, "bilinear row pattern")
    partner = []
    for c in range(NK):
        ps, sp_ = ZPSI[c], ZSPSI[c]
        _need(rz[RPSI[c], ps] == 1, "psi pivot")
        others = nzcols(RPSI[c]) - {ps}
        _need(others <= set(ZG) and len(others) <= 1, "psi row pattern")
        partner.append(ZG.index(next(iter(others))) if others else -1)
        if others:
            _need(not (rz[RPSI[c], ZG[partner[c]]].free_symbols & set(z)), "psi row coefficient depends on z")
        bs = [(ZB1[c], ZSB1[c], RV1[c], RCB1[c])] + ([(ZB2[c], ZSB2[c], RV2[c], RCB2[c])] if dims[c] == 3 else [])
        A = RCA[c]
        _need(rz[A, ps] == z[sp_] and rz[A, sp_] == z[ps], "cone head row")
        cols = {ps, sp_}
        for (b, sb, rv, rb) in bs:
            _need(rz[rv, sb] in (-1, 1), "velocity pivot")
            _need(nzcols(rv) <= qcols | {sb}, "velocity row pattern")
            _need(rz[A, b] == z[sb] and rz[A, sb] == z[b], "cone head row")
            _need(rz[rb, ps] == z[sb] and rz[rb, sb] == z[ps] and rz[rb, b] == z[sp_] and rz[rb, sp_] == z[b], "cone tail row")
            _need(nzcols(rb) == {ps, b, sp_, sb}, "cone tail row pattern")
            cols |= {b, sb}
        _need(nzcols(A) == cols, "cone head row pattern")
    for k in RDYN:
        _need(nzcols(k) <= qcols | set(ZG) | set(ZB1) | {b for b in ZB2 if b >= 0}, "dynamics row couples to something else than q, gamma, b")
    # role swap candidates as in the serial code: |psi| > |s_psi| exchanges the head row with the FIRST tail row
    _need(len(m.swaps) == NK, "swap spec count")
    for c, ((ra, rb), (ca, cb)) in enumerate(m.swaps):
        _need((ra, rb) == (RCA[c], RCB1[c]) and (ca, cb) == (ZB1[c], ZSPSI[c]), "swap spec")
    # the state program of a 3-d cone: (B1 -> b1), (B2 -> b2), (A -> s_psi) after the swap -- what od_coop3.h eliminates
    order = list(m.elim_state) if m.elim_state else list(m.elim)
    pos = {rc: k for k, rc in enumerate(order)}
    for c in range(NK):
        if dims[c] == 3:
            _need((RCB1[c], ZB1[c]) in pos and (RCB2[c], ZB2[c]) in pos and (RCA[c], ZSPSI[c]) in pos, "state program lacks the cone pivots")
            _need(pos[(RCB1[c], ZB1[c])] < pos[(RCB2[c], ZB2[c])] < pos[(RCA[c], ZSPSI[c])], "cone pivot order")
    # lanes: contact i -> lane i, cone c -> lane NC + c; a psi row reads gamma of its partner contact SH lanes below
    shifts = {NC + c - partner[c] for c in range(NK) if partner[c] >= 0}
    _need(len(shifts) <= 1, "cones reach their partner contacts at different lane distances")
    SH = shifts.pop() if shifts else 0
    partner_bits = sum(1 << (NC + c) for c in range(NK) if partner[c] >= 0)

    nzidx = {ij: k for k, ij in enumerate(d.rz_nz)}
    pr = _DevicePrinter()

    def aref(i, j):
        e = rz[i, j]
        if e == 0:
            return "0.0"
        if e.is_Number:
            return repr(float(e))
        return "a[%d]" % nzidx[(i, j)]

    def lane_of_z(k):
        if k in ZG:
            return ZG.index(k), "P0"
        if k in ZS:
            return ZS.index(k), "D0"
        for name, tab in (("P0", ZPSI), ("P1", ZB1), ("P2", ZB2), ("D0", ZSPSI), ("D1", ZSB1), ("D2", ZSB2)):
            if k in tab:
                return NC + tab.index(k), name
        raise KeyError(k)

    zset = set(z)
    zidx = {s: i for i, s in enumerate(z)}

    def force_vars(exprs):
        out = set()
        for e in exprs:
            for s in e.free_symbols:
                if s in zset and zidx[s] not in qcols:
                    out.add(zidx[s])
        return sorted(out)

    fv_r = force_vars([r0[k] for k in RDYN])
    for i in range(NC):
        _need(force_vars([r0[RSL[i]]]) == [ZS[i]] and sp.diff(r0[RSL[i]], z[ZS[i]]) == 1, "slack residual")
        _need(sp.expand(r0[RBIL[i]] - z[ZG[i]] * z[ZS[i]]) == 0, "bilinear residual")
    for c in range(NK):
        _need(force_vars([r0[RV1[c]]]) == [ZSB1[c]] and sp.diff(r0[RV1[c]], z[ZSB1[c]]) == rz[RV1[c], ZSB1[c]], "velocity residual")
        head = z[ZPSI[c]] * z[ZSPSI[c]] + z[ZB1[c]] * z[ZSB1[c]]
        _need(sp.expand(r0[RCB1[c]] - (z[ZPSI[c]] * z[ZSB1[c]] + z[ZB1[c]] * z[ZSPSI[c]])) == 0, "cone tail residual")
        if dims[c] == 3:
            _need(force_vars([r0[RV2[c]]]) == [ZSB2[c]] and sp.diff(r0[RV2[c]], z[ZSB2[c]]) == rz[RV2[c], ZSB2[c]], "velocity residual 2")
            head = head + z[ZB2[c]] * z[ZSB2[c]]
            _need(sp.expand(r0[RCB2[c]] - (z[ZPSI[c]] * z[ZSB2[c]] + z[ZB2[c]] * z[ZSPSI[c]])) == 0, "cone tail residual 2")
        _need(sp.expand(r0[RCA[c]] - head) == 0, "cone head residual")
        want = z[ZPSI[c]] + (rz[RPSI[c], ZG[partner[c]]] * z[ZG[partner[c]]] if partner[c] >= 0 else 0)
        _need(not (sp.expand(r0[RPSI[c]] - want).free_symbols & zset), "psi residual")
    used_rz = ([rz[k, j] for k in RDYN for j in range(nz)] + [rz[RSL[i], j] for i in range(NC) for j in zq]
               + [rz[RV1[c], j] for c in range(NK) for j in zq] + [rz[RV2[c], j] for c in range(NK) if RV2[c] >= 0 for j in zq])
    fv_rz = force_vars(used_rz)
    for k in fv_r + fv_rz:
        _need(lane_of_z(k)[1] in ("P0", "P1", "P2"), "the dynamics rows read a dual variable")

    PN = [[k for k in range(nq) if rz[RDYN[k], ZG[i]] != 0] for i in range(NC)]
    PJ = [[j for j in range(nq) if rz[RSL[i], zq[j]] != 0] for i in range(NC)]
    PNB1 = [[k for k in range(nq) if rz[RDYN[k], ZB1[c]] != 0] for c in range(NK)]
    PNB2 = [[k for k in range(nq) if ZB2[c] >= 0 and rz[RDYN[k], ZB2[c]] != 0] for c in range(NK)]
    PJV1 = [[j for j in range(nq) if rz[RV1[c], zq[j]] != 0] for c in range(NK)]
    PJV2 = [[j for j in range(nq) if RV2[c] >= 0 and rz[RV2[c], zq[j]] != 0] for c in range(NK)]
    UPJ = [any(j in PJ[i] for i in range(NC)) for j in range(nq)]
    UPV = [any(j in PJV1[c] or j in PJV2[c] for c in range(NK)) for j in range(nq)]

    def lanes8(f, default):
        t = [default] * 8
        for r in range(NC + NK):
            t[r] = f(r)
        return t

    def jf_const(j, second):
        def f(r):
            if r < NC:
                e = sp.Integer(0) if second else rz[RSL[r], zq[j]]
            else:
                row = RV2[r - NC] if second else RV1[r - NC]
                e = rz[row, zq[j]] if row >= 0 else sp.Integer(0)
            return float(e) if e.is_Number else 0.0
        return lanes8(f, 0.0)

    zi = [None if isinstance(e, tuple) else float(e) for e in m.z_init]

    def zfield(field, tab_contact, default):
        def f(r):
            if r < NC:
                k = tab_contact.get(field, -1)
                k = k[r] if isinstance(k, list) else k
            else:
                k = {"P0": ZPSI, "P1": ZB1, "P2": ZB2, "D0": ZSPSI, "D1": ZSB1, "D2": ZSB2}[field][r - NC]
            return k
        return f

    contact_tab = {"P0": ZG, "D0": ZS}

    def zinit(field):
        g = zfield(field, contact_tab, None)

        def f(r):
            k = g(r)
            return zi[k] if k is not None and k >= 0 else 0.0
        return lanes8(f, {"P0": 1.0, "D0": 1.0}.get(field, 0.0))

    def zindex(field):
        g = zfield(field, contact_tab, None)
        return lanes8(lambda r: (g(r) if g(r) is not None and g(r) >= 0 else -1), -1)

    o = io.StringIO()
    n = m.name
    w = o.write
    w("// GENERATED by optimization_dynamics_amd.codegen.coop3 -- do not edit.\n")
    w("// Lane roles and value routing of model %s for the 8-lanes-per-problem cooperative solver (csrc/od_coop3.h):\n" % n)
    w("// contact i in lane i, cone c in lane %d + c.  Structure checked against the symbolic Jacobian at generation.\n" % NC)
    w("#pragma once\n#include \"%s.h\"\n#include \"../od_coop3.h\"\n\nnamespace od {\n\n" % n)
    w("struct Coop3_%s {\n" % n)
    w("  using M = Model_%s;\n" % n)
    w("  static constexpr int NQ = %d, NC = %d, NK = %d, SH = %d;\n" % (nq, NC, NK, SH))
    w("  static constexpr bool DIM3 = %s;          // a cone of dimension 3 among them\n" % ("true" if DIM3 else "false"))
    w("  static constexpr unsigned PARTNER_BITS = %du;   // cone lanes whose psi row reads a contact force\n" % partner_bits)

    def arr(name, xs, ty="int"):
        xs = list(xs) or [0]
        w("  static constexpr %s %s[%d] = {%s};\n" % (ty, name, len(xs), ", ".join(str(x).lower() if ty == "bool" else str(x) for x in xs)))

    arr("ZQ", zq); arr("RDYN", RDYN)
    arr("UPJ", UPJ, "bool"); arr("UPV", UPV, "bool")
    for nm, second in (("JFA", False), ("JFB", True)):
        w("  static constexpr double %s[%d][8] = {\n" % (nm, nq))
        for j in range(nq):
            w("    {%s},\n" % ", ".join(repr(v) for v in jf_const(j, second)))
        w("  };\n")
    arr("CVA", [repr(v) for v in lanes8(lambda r: 0.0 if r < NC else float(rz[RV1[r - NC], ZSB1[r - NC]]), 0.0)], "double")
    arr("CVB", [repr(v) for v in lanes8(lambda r: 0.0 if (r < NC or RV2[r - NC] < 0) else float(rz[RV2[r - NC], ZSB2[r - NC]]), 0.0)], "double")
    for fld in ("P0", "P1", "P2", "D0", "D1", "D2"):
        arr("ZI_" + fld, [repr(v) for v in zinit(fld)], "double")
    for fld in ("P0", "P1", "P2", "D0", "D1", "D2"):
        arr("IDX_" + fld, zindex(fld))

    def gather(name, fv):
        w("  template <class RO, class V> OD_HD static void %s(const V& P0, const V& P1, const V& P2, double* zr) {\n" % name)
        ops = []
        for k in fv:
            lane, fld = lane_of_z(k)
            ops.append(("bc", lane, "zr[%d]" % k, fld))
        if ops:
            w(_dpp_block8(ops))
        w("  }\n")

    w("  // replicated copies of the contact forces the dynamics rows (gather_r) / their Jacobian (gather_rz) read\n")
    gather("gather_r", fv_r)
    gather("gather_rz", fv_rz)
    w("  // the same copies from per-role arrays (lane-parallel line-search trials, od_coop3.h::c3_trials_lanes)\n")
    w("  template <class V> OD_HD static void scatter_r(const V* P0, const V* P1, const V* P2, V* zr) {\n")
    for k in fv_r:
        lane, fld = lane_of_z(k)
        w("    zr[%d] = %s[%d];\n" % (k, fld, lane))
    w("  }\n")
    arr("E1ROWA", RSL + RV1)
    arr("E1ROWB", [-1] * NC + RV2)
    w("  // aux expressions of the lane: -(phi_i) for contact i, the first / second tangential velocity for cone c (rows of the\n")
    w("  // serial residual evaluated with s_i = s_b = 0)\n")
    for nm, rows in (("pick_e1a", RSL + RV1), ("pick_e1b", [-1] * NC + RV2)):
        w("  template <class RO, class L_> OD_HD static typename RO::V %s(const L_& L, const double* rr) {\n" % nm)
        w("    typename RO::V e = typename RO::V(0.0);\n")
        for r in range(NC + NK):
            if rows[r] >= 0:
                w("    e = RO::sel(L.role[%d], rr[%d], e);\n" % (r, rows[r]))
        w("    return e;\n  }\n")
    w("  // the lane's aux-row Jacobians w.r.t. q: constant entries from JFA / JFB, the others from the evaluated rz\n")
    w("  template <class RO, class L_> OD_HD static void build_jf(const L_& L, const double* a, typename RO::V* JFa, typename RO::V* JFb) {\n")
    for j in range(nq):
        w("    JFa[%d] = L.jfa[%d];\n    JFb[%d] = L.jfb[%d];\n" % (j, j, j, j))
        for r in range(NC + NK):
            rowa = RSL[r] if r < NC else RV1[r - NC]
            rowb = -1 if r < NC else RV2[r - NC]
            for nm, row in (("JFa", rowa), ("JFb", rowb)):
                if row >= 0:
                    e = rz[row, zq[j]]
                    if e != 0 and not e.is_Number:
                        w("    %s[%d] = RO::sel(L.role[%d], a[%d], %s[%d]);\n" % (nm, j, r, nzidx[(row, zq[j])], nm, j))
    w("  }\n")
    w("  OD_HD static void dqq_from(const double* a, double* dqq) {\n")
    for k in range(nq):
        for j in range(nq):
            w("    dqq[%d] = %s;\n" % (k + nq * j, aref(RDYN[k], zq[j])))
    w("  }\n")
    w("  template <class NV, class NB> OD_HD static void couplings(const double* a, NV& nv, NB& nbv) {\n")
    for i in range(NC):
        for k in range(nq):
            w("    nv[%d][%d] = %s;\n" % (i, k, aref(RDYN[k], ZG[i])))
    for c in range(NK):
        for k in range(nq):
            w("    nbv[%d][0][%d] = %s;\n" % (c, k, aref(RDYN[k], ZB1[c])))
            w("    nbv[%d][1][%d] = %s;\n" % (c, k, aref(RDYN[k], ZB2[c]) if ZB2[c] >= 0 else "0.0"))
    w("  }\n")
    consts = {}

    def coupling(kind, row, col, name):
        e = rz[row, col]
        if e.is_Number:
            c = float(e)
            if c < 0:
                kind = "fnmac" if kind == "fmac" else "fmac"
            sym = consts.setdefault(abs(c), "kc%d_" % len(consts))
            return kind, sym
        return kind, name

    def const_decls():
        return "".join("    const double %s = %r;\n" % (sym, c) for c, sym in consts.items())

    w("  // Schur complement of the dynamics rows: d gamma_i = ty_i + t_i . dq,  d b_{c,m} = Wy_m,c - Wb_m,c . dq\n")
    w("  template <class RO, class F> OD_HD static void schur(const F& f, const typename RO::V* Wb1, const typename RO::V* Wb2, double* dqq) {\n")
    ops = []
    consts.clear()
    for i in range(NC):
        for k in PN[i]:
            kind, mul = coupling("fmac", RDYN[k], ZG[i], "f.nv[%d][%d]" % (i, k))
            for j in PJ[i]:
                ops.append((kind, i, "dqq[%d]" % (k + nq * j), "f.t[%d]" % j, mul))
    for c in range(NK):
        wj = sorted(set(PJV1[c]) | set(PJV2[c]) | (set(PJ[partner[c]]) if partner[c] >= 0 else set()))
        for mth, (PNB, ZB, Wn) in enumerate(((PNB1, ZB1, "Wb1"), (PNB2, ZB2, "Wb2"))):
            for k in PNB[c]:
                kind, mul = coupling("fnmac", RDYN[k], ZB[c], "f.nbv[%d][%d][%d]" % (c, mth, k))
                for j in wj:
                    ops.append((kind, NC + c, "dqq[%d]" % (k + nq * j), "%s[%d]" % (Wn, j), mul))
    w(const_decls())
    w(_dpp_block8(ops))
    w("  }\n")
    w("  template <class RO, class F> OD_HD static void rhs_update(const F& f, const typename RO::V& ty, const typename RO::V& Wy1, const typename RO::V& Wy2, double* rd) {\n")
    ops = []
    consts.clear()
    for i in range(NC):
        for k in PN[i]:
            kind, mul = coupling("fnmac", RDYN[k], ZG[i], "f.nv[%d][%d]" % (i, k))
            ops.append((kind, i, "rd[%d]" % k, "ty", mul))
    for c in range(NK):
        for mth, (PNB, ZB, Wn) in enumerate(((PNB1, ZB1, "Wy1"), (PNB2, ZB2, "Wy2"))):
            for k in PNB[c]:
                kind, mul = coupling("fnmac", RDYN[k], ZB[c], "f.nbv[%d][%d][%d]" % (c, mth, k))
                ops.append((kind, NC + c, "rd[%d]" % k, Wn, mul))
    w(const_decls())
    w(_dpp_block8(ops))
    w("  }\n")
    w("  // psi rows: psi_c + g[c] * gamma_partner + gc[c] (theta only)\n")
    w("  template <class T> OD_HD static void eval_gcoef(const T* th, T* g, T* gc) {\n")
    for c in range(NK):
        ge = rz[RPSI[c], ZG[partner[c]]] if partner[c] >= 0 else sp.Integer(0)
        gce = r0[RPSI[c]] - z[ZPSI[c]] - (ge * z[ZG[partner[c]]] if partner[c] >= 0 else 0)
        w("    g[%d] = %s;\n    gc[%d] = %s;\n" % (c, pr.doprint(sp.nsimplify(ge) if ge == 0 else ge), c, pr.doprint(sp.expand(gce))))
    w("  }\n")
    w("};\n\n}  // namespace od\n")
    return o.getvalue()
