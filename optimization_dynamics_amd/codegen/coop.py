"""Tables + glue for the cooperative (16 lanes per problem) solver, csrc/od_coop.h.

The cooperative solver keeps the configuration q and the dynamics rows replicated in every lane of a DPP row and
gives every contact (orthant pair with its slack and bilinear rows) and every friction cone (with its psi-,
tangential-velocity- and cone-product rows) a lane of its own.  That needs the block structure every contact-implicit
model of the reference has (src/models/*/model.jl: `residual`), which this module reads off the symbolic Jacobian
and CHECKS entry by entry -- a model that does not have it is rejected (returns None) and keeps the lane-per-problem
kernels only.  No new arithmetic is generated here: residual and Jacobian values come from the model's own
``eval_r`` / ``eval_rz`` (gen/<model>.h); this file emits where each value goes.
"""
from __future__ import annotations

import io
from typing import Dict, List, Optional

import sympy as sp

from .emit import Derived, _DevicePrinter
from .models import ModelSpec


class NotCoop(Exception):
    pass


def _need(cond, msg):
    if not cond:
        raise NotCoop(msg)


def _dpp_block(ops, indent="    ", max_operands=28):
    """C++ for a list of cross-lane operations, in two flavours: on the device ONE inline-asm block per chunk (a
    single `s_nop 1` covers the DPP read-after-write hazard of all its instructions: they read registers written
    before the block only), in the host test build the RowEmu calls.
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
            if op[0] == "bc":
                if op[2] not in outs:
                    outs.append(op[2])
            elif op[2] not in outs:
                outs.append(op[2])
        for op in ch:
            for e in (op[3:] if op[0] != "bc" else op[3:4]):
                if e not in ins and e not in outs:
                    ins.append(e)
        num = {e: k for k, e in enumerate(outs + ins)}
        lines = ["s_nop 1"]
        for op in ch:
            if op[0] == "bc":
                lines.append("v_mov_b64_dpp %%%d, %%%d row_newbcast:%d row_mask:0xf bank_mask:0xf" % (num[op[2]], num[op[3]], op[1]))
            else:
                neg = "-" if op[0] == "fnmac" else ""
                lines.append("v_fmac_f64_dpp %%%d, %%%d, %s%%%d row_newbcast:%d row_mask:0xf bank_mask:0xf"
                             % (num[op[2]], num[op[3]], neg, num[op[4]], op[1]))
        is_bc = {op[2] for op in ch if op[0] == "bc"}
        cons_out = ", ".join(('"=&v"(%s)' if e in is_bc else '"+v"(%s)') % e for e in outs)
        cons_in = ", ".join('"v"(%s)' % e for e in ins)
        out.append(indent + 'asm("' + '\\n\\t"\n%s    "'.join([lines[0]] + lines[1:]) .replace("%s", indent) + '"')
        out.append(indent + "    : %s : %s);" % (cons_out, cons_in))
    out.append("#else")
    for op in ops:
        if op[0] == "bc":
            out.append(indent + "%s = RO::template bc<%d>(%s);" % (op[2], op[1], op[3]))
        else:
            out.append(indent + "RO::template %s<%d>(%s, %s, %s);" % (op[0], op[1], op[2], op[3], op[4]))
    out.append("#endif")
    return "\n".join(out) + "\n"


def emit_coop(m: ModelSpec, d: Derived) -> Optional[str]:
    try:
        return _emit(m, d)
    except NotCoop as e:
        print("  (no cooperative kernel for %s: %s)" % (m.name, e))
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
    for p, dd in m.soc:
        _need(len(p) == 2 and len(dd) == 2, "cone dimension != 2")
    ZG, ZS = list(m.ort[0]), list(m.ort[1])
    RBIL = list(m.ortr)
    RSL = []
    for i in range(NC):
        _need(ZS[i] in piv_of_col, "slack row of contact %d" % i)
        RSL.append(piv_of_col[ZS[i]])
    ZPSI = [p[0] for p, _ in m.soc]
    ZB = [p[1] for p, _ in m.soc]
    ZSPSI = [dd[0] for _, dd in m.soc]
    ZSB = [dd[1] for _, dd in m.soc]
    RCA = [rr[0] for rr in m.socri]
    RCB = [rr[1] for rr in m.socri]
    RPSI, RVEL = [], []
    for c in range(NK):
        _need(ZPSI[c] in piv_of_col and ZSB[c] in piv_of_col, "psi / velocity row of cone %d" % c)
        RPSI.append(piv_of_col[ZPSI[c]])
        RVEL.append(piv_of_col[ZSB[c]])
    used = set(RSL + RBIL + RPSI + RVEL + RCA + RCB)
    RDYN = [i for i in range(nz) if i not in used]
    _need(len(RDYN) == nq, "dynamics rows")
    _need(sorted(zq + ZG + ZS + ZPSI + ZB + ZSPSI + ZSB) == list(range(nz)), "variables do not partition")
    # the static elimination order of the lane-per-problem code must be this block order
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
        ps, b, sp_, sb = ZPSI[c], ZB[c], ZSPSI[c], ZSB[c]
        _need(rz[RPSI[c], ps] == 1, "psi pivot")
        others = nzcols(RPSI[c]) - {ps}
        _need(others <= set(ZG) and len(others) <= 1, "psi row pattern")
        partner.append(ZG.index(next(iter(others))) if others else -1)
        if others:
            _need(not (rz[RPSI[c], ZG[partner[c]]].free_symbols & set(z)), "psi row coefficient depends on z")
        _need(rz[RVEL[c], sb] in (-1, 1), "velocity pivot")
        _need(nzcols(RVEL[c]) <= qcols | {sb}, "velocity row pattern")
        A, B = RCA[c], RCB[c]
        _need(rz[A, ps] == z[sp_] and rz[A, sp_] == z[ps] and rz[A, b] == z[sb] and rz[A, sb] == z[b], "cone head row")
        _need(rz[B, ps] == z[sb] and rz[B, sb] == z[ps] and rz[B, b] == z[sp_] and rz[B, sp_] == z[b], "cone tail row")
        _need(nzcols(A) == {ps, b, sp_, sb} and nzcols(B) == {ps, b, sp_, sb}, "cone row pattern")
    for k in RDYN:
        _need(nzcols(k) <= qcols | set(ZG) | set(ZB), "dynamics row couples to something else than q, gamma, b")
    # role swap candidates as in the serial code: |psi| > |s_psi|
    for c, ((ra, rb), (ca, cb)) in enumerate(m.swaps):
        _need((ra, rb) == (RCA[c], RCB[c]) and (ca, cb) == (ZB[c], ZSPSI[c]), "swap spec")
    _need(len(m.swaps) == NK, "swap spec count")
    # lanes: contact i -> lane i, cone c -> lane NC + c; a psi row reads gamma of its partner contact SH lanes below
    shifts = {NC + c - partner[c] for c in range(NK) if partner[c] >= 0}
    _need(len(shifts) <= 1, "cones reach their partner contacts at different lane distances")
    SH = shifts.pop() if shifts else 0

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
        """(lane, field, clamped field) holding z[k]"""
        if k in ZG:
            return ZG.index(k), "P0"
        if k in ZS:
            return ZS.index(k), "D0"
        if k in ZPSI:
            return NC + ZPSI.index(k), "P0"
        if k in ZB:
            return NC + ZB.index(k), "P1"
        if k in ZSPSI:
            return NC + ZSPSI.index(k), "D0"
        return NC + ZSB.index(k), "D1"

    zset = set(z)
    zidx = {s: i for i, s in enumerate(z)}

    def force_vars(exprs):
        out = set()
        for e in exprs:
            for s in e.free_symbols:
                if s in zset and zidx[s] not in qcols:
                    out.add(zidx[s])
        return sorted(out)

    # what the replicated residual / Jacobian evaluation reads besides q
    e1_rows = RSL + RVEL
    fv_r = force_vars([r0[k] for k in RDYN])
    for i in range(NC):
        _need(force_vars([r0[RSL[i]]]) == [ZS[i]] and sp.diff(r0[RSL[i]], z[ZS[i]]) == 1, "slack residual")
    for c in range(NK):
        _need(force_vars([r0[RVEL[c]]]) == [ZSB[c]] and sp.diff(r0[RVEL[c]], z[ZSB[c]]) == rz[RVEL[c], ZSB[c]], "velocity residual")
        want = z[ZPSI[c]] + (rz[RPSI[c], ZG[partner[c]]] * z[ZG[partner[c]]] if partner[c] >= 0 else 0)
        rest = sp.expand(r0[RPSI[c]] - want)
        _need(not (rest.free_symbols & zset), "psi residual")
        _need(sp.expand(r0[RCA[c]] - (z[ZPSI[c]] * z[ZSPSI[c]] + z[ZB[c]] * z[ZSB[c]])) == 0, "cone head residual")
        _need(sp.expand(r0[RCB[c]] - (z[ZPSI[c]] * z[ZSB[c]] + z[ZB[c]] * z[ZSPSI[c]])) == 0, "cone tail residual")
    for i in range(NC):
        _need(sp.expand(r0[RBIL[i]] - z[ZG[i]] * z[ZS[i]]) == 0, "bilinear residual")
    used_rz = ([rz[k, j] for k in RDYN for j in range(nz)] + [rz[RSL[i], j] for i in range(NC) for j in zq]
               + [rz[RVEL[c], j] for c in range(NK) for j in zq])
    fv_rz = force_vars(used_rz)

    PN = [[k for k in range(nq) if rz[RDYN[k], ZG[i]] != 0] for i in range(NC)]
    PJ = [[j for j in range(nq) if rz[RSL[i], zq[j]] != 0] for i in range(NC)]
    PNB = [[k for k in range(nq) if rz[RDYN[k], ZB[c]] != 0] for c in range(NK)]
    PJV = [[j for j in range(nq) if rz[RVEL[c], zq[j]] != 0] for c in range(NK)]
    UPJ = [any(j in PJ[i] for i in range(NC)) for j in range(nq)]
    UPV = [any(j in PJV[c] for c in range(NK)) for j in range(nq)]

    def lanes16(f, default):
        """16-lane table: role lanes 0..NC+NK-1 from f(role), their mirrors 8.., the rest `default`"""
        t = [default] * 16
        for r in range(NC + NK):
            t[r] = t[r + 8] = f(r)
        return t

    def jfc(j):
        def f(r):
            e = rz[RSL[r], zq[j]] if r < NC else rz[RVEL[r - NC], zq[j]]
            return float(e) if e.is_Number else 0.0
        return lanes16(f, 0.0)

    zi = []
    for e in m.z_init:
        zi.append(None if isinstance(e, tuple) else float(e))

    def zinit(field):
        def f(r):
            if r < NC:
                return {"P0": zi[ZG[r]], "D0": zi[ZS[r]], "P1": 0.0, "D1": 0.0}[field]
            c = r - NC
            return {"P0": zi[ZPSI[c]], "P1": zi[ZB[c]], "D0": zi[ZSPSI[c]], "D1": zi[ZSB[c]]}[field]
        return lanes16(f, {"P0": 1.0, "D0": 1.0, "P1": 0.0, "D1": 0.0}[field])

    def zindex(field):
        def f(r):
            if r < NC:
                return {"P0": ZG[r], "D0": ZS[r], "P1": -1, "D1": -1}[field]
            c = r - NC
            return {"P0": ZPSI[c], "P1": ZB[c], "D0": ZSPSI[c], "D1": ZSB[c]}[field]
        t = lanes16(f, -1)
        return t[:8] + [-1] * 8          # only the first half of the row stores

    o = io.StringIO()
    n = m.name
    w = o.write
    w("// GENERATED by optimization_dynamics_amd.codegen.coop -- do not edit.\n")
    w("// Lane roles and value routing of model %s for the cooperative solver (csrc/od_coop.h): contact i in lane i,\n" % n)
    w("// cone c in lane %d + c, lanes 8..15 mirror 0..7.  Structure checked against the symbolic Jacobian at generation.\n" % NC)
    w("#pragma once\n#include \"%s.h\"\n#include \"../od_coop.h\"\n\nnamespace od {\n\n" % n)
    w("struct Coop_%s {\n" % n)
    w("  using M = Model_%s;\n" % n)
    w("  static constexpr int NQ = %d, NC = %d, NK = %d, SH = %d;\n" % (nq, NC, NK, SH))
    w("  // chosen automatically for small batches (csrc/od_model_tu.inc: up to 8192 problems with four or more contacts / cones to\n")
    w("  // share out, up to 4096 with fewer -- the acrobot's two contacts pay through the row-private backtracking of a solve\n")
    w("  // that jams: 65 536 knots of the joint-limit workload 0.40 ms against 1.20 lane-per-problem)\n")
    w("  static constexpr bool AUTO = true;\n")

    def arr(name, xs, ty="int"):
        xs = list(xs) or [0]
        w("  static constexpr %s %s[%d] = {%s};\n" % (ty, name, len(xs), ", ".join(str(x).lower() if ty == "bool" else str(x) for x in xs)))

    arr("ZQ", zq); arr("RDYN", RDYN)
    arr("UPJ", UPJ, "bool"); arr("UPV", UPV, "bool")
    w("  static constexpr double JFC[%d][16] = {\n" % nq)
    for j in range(nq):
        w("    {%s},\n" % ", ".join(repr(v) for v in jfc(j)))
    w("  };\n")
    arr("CV", [repr(v) for v in lanes16(lambda r: 0.0 if r < NC else float(rz[RVEL[r - NC], ZSB[r - NC]]), 0.0)], "double")
    for fld in ("P0", "P1", "D0", "D1"):
        arr("ZI_" + fld, [repr(v) for v in zinit(fld)], "double")
    for fld in ("P0", "P1", "D0", "D1"):
        arr("IDX_" + fld, zindex(fld))

    def gather(name, fv, clamp):
        w("  template <class RO, class V> OD_HD static void %s(const V& P0, const V& P1, const V& D0, const V& D1, double* zr) {\n" % name)
        ops = []
        for k in fv:
            lane, fld = lane_of_z(k)
            ops.append(("bc", lane, "zr[%d]" % k, fld))
        w(_dpp_block(ops))
        w("  }\n")

    w("  // replicated copies of the contact forces the dynamics rows (gather_r) / their Jacobian (gather_rz) read\n")
    gather("gather_r", fv_r, False)
    gather("gather_rz", fv_rz, True)
    w("  // the same copies from per-role arrays (parallel line-search trials, od_coop.h::coop_trials_lanes)\n")
    w("  template <class V> OD_HD static void scatter_r(const V* P0, const V* P1, const V* D0, const V* D1, V* zr) {\n")
    for k in fv_r:
        lane, fld = lane_of_z(k)
        w("    zr[%d] = %s[%d];\n" % (k, fld, lane))
    w("  }\n")
    arr("E1ROW", e1_rows)
    w("  // aux-1 expression of the lane: -(phi_i) for contact i, the tangential velocity for cone c (rows of the serial\n")
    w("  // residual evaluated with s_i = s_b = 0)\n")
    w("  template <class RO, class L_> OD_HD static typename RO::V pick_e1(const L_& L, const double* rr) {\n")
    w("    typename RO::V e = typename RO::V(0.0);\n")
    for r in range(NC + NK):
        w("    e = RO::sel(L.role[%d], rr[%d], e);\n" % (r, e1_rows[r]))
    w("    return e;\n  }\n")
    w("  // the lane's aux-row Jacobian w.r.t. q: constant entries from JFC, the others from the evaluated rz\n")
    w("  template <class RO, class L_> OD_HD static void build_jf(const L_& L, const double* a, typename RO::V* JF) {\n")
    for j in range(nq):
        w("    JF[%d] = L.jfc[%d];\n" % (j, j))
        for r in range(NC + NK):
            e = rz[RSL[r], zq[j]] if r < NC else rz[RVEL[r - NC], zq[j]]
            if e != 0 and not e.is_Number:
                row = RSL[r] if r < NC else RVEL[r - NC]
                w("    JF[%d] = RO::sel(L.role[%d], a[%d], JF[%d]);\n" % (j, r, nzidx[(row, zq[j])], j))
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
            w("    nbv[%d][%d] = %s;\n" % (c, k, aref(RDYN[k], ZB[c])))
    w("  }\n")
    # couplings that are compile-time constants are passed as shared literal operands (one register pair for every 1.0
    # in a block instead of one per entry); a negative constant flips fmac <-> fnmac
    consts = {}

    def coupling(kind, row, col, name):
        """(op kind, multiplier operand) for  acc (+|-)= src * rz[row, col]"""
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

    w("  // Schur complement of the dynamics rows: d gamma_i = ty_i + t_i . dq,  d b_c = Wy_c - W_c . dq\n")
    w("  template <class RO, class F> OD_HD static void schur(const F& f, const typename RO::V* W, double* dqq) {\n")
    ops = []
    consts.clear()
    for i in range(NC):
        for k in PN[i]:
            kind, mul = coupling("fmac", RDYN[k], ZG[i], "f.nv[%d][%d]" % (i, k))
            for j in PJ[i]:
                ops.append((kind, i, "dqq[%d]" % (k + nq * j), "f.t[%d]" % j, mul))
    for c in range(NK):
        wj = sorted(set(PJV[c]) | (set(PJ[partner[c]]) if partner[c] >= 0 else set()))
        for k in PNB[c]:
            kind, mul = coupling("fnmac", RDYN[k], ZB[c], "f.nbv[%d][%d]" % (c, k))
            for j in wj:
                ops.append((kind, NC + c, "dqq[%d]" % (k + nq * j), "W[%d]" % j, mul))
    w(const_decls())
    w(_dpp_block(ops))
    w("  }\n")
    w("  template <class RO, class F> OD_HD static void rhs_update(const F& f, const typename RO::V& ty, const typename RO::V& Wy, double* rd) {\n")
    ops = []
    consts.clear()
    for i in range(NC):
        for k in PN[i]:
            kind, mul = coupling("fnmac", RDYN[k], ZG[i], "f.nv[%d][%d]" % (i, k))
            ops.append((kind, i, "rd[%d]" % k, "ty", mul))
    for c in range(NK):
        for k in PNB[c]:
            kind, mul = coupling("fnmac", RDYN[k], ZB[c], "f.nbv[%d][%d]" % (c, k))
            ops.append((kind, NC + c, "rd[%d]" % k, "Wy", mul))
    w(const_decls())
    w(_dpp_block(ops))
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
