"""Straight-line C emission for the residual models (sympy -> CSE -> C).

Two flavours are produced from the SAME symbolic statement:

* ``oracle`` : plain C99, ``double`` only, dense column-major Jacobians -- consumed by the CPU
  restatement under ``oracle/`` (test infrastructure).
* ``device`` : templated ``OD_HD`` (host/device) inline functions on sparse Jacobian value arrays,
  plus a statically ordered sparse elimination of the KKT matrix with a small runtime-pivoted
  dense tail -- consumed by the HIP kernels under ``optimization_dynamics_amd/csrc``.

This is the counterpart of ``Symbolics.build_function(...)[2]`` in the reference's
``src/models/*/codegen.jl`` (e.g. acrobot/codegen.jl:18-31).
"""
from __future__ import annotations

import io
from typing import Dict, List, Sequence, Tuple

import sympy as sp
from sympy.printing.c import C99CodePrinter

from .models import ModelSpec


# --------------------------------------------------------------------------------------
# printers
# --------------------------------------------------------------------------------------
class _OraclePrinter(C99CodePrinter):
    def _print_Pow(self, expr):
        b, e = expr.base, expr.exp
        if e.is_Integer and 2 <= int(e) <= 12:
            return "od_powi(%s, %d)" % (self._print(b), int(e))
        if e.is_Integer and -12 <= int(e) <= -1:
            if int(e) == -1:
                return "(1.0/(%s))" % self._print(b)
            return "(1.0/od_powi(%s, %d))" % (self._print(b), -int(e))
        return super()._print_Pow(expr)


class _DevicePrinter(C99CodePrinter):
    """Prints with scalar type ``T``: literals wrapped, math functions overloaded (od_sin...)."""

    def _print_Float(self, expr):
        s = super()._print_Float(expr)
        return "T(%s)" % s

    def _print_Integer(self, expr):
        return "T(%d)" % int(expr)

    def _print_Rational(self, expr):
        return "(T(%d)/T(%d))" % (int(expr.p), int(expr.q))

    def _print_Pi(self, expr):
        return "T(3.14159265358979323846)"

    def _print_od_rootinv(self, expr):
        return "od_rootinv<%d>(%s)" % (int(expr.args[1]), self._print(expr.args[0]))

    def _print_Pow(self, expr):
        b, e = expr.base, expr.exp
        if e.is_Integer and 2 <= int(e) <= 64:
            return "od_powi<%d>(%s)" % (int(e), self._print(b))
        if e.is_Integer and -12 <= int(e) <= -1:
            if int(e) == -1:
                return "od_rcp(%s)" % self._print(b)
            return "od_rcp(od_powi<%d>(%s))" % (-int(e), self._print(b))
        if e == sp.Rational(1, 2):
            return "od_sqrt(%s)" % self._print(b)
        if e == -sp.Rational(1, 2):
            return "od_rsqrt(%s)" % self._print(b)
        return "od_pow(%s, %s)" % (self._print(b), self._print(e))

    def _print_sin(self, expr):
        return "od_sin(%s)" % self._print(expr.args[0])

    def _print_cos(self, expr):
        return "od_cos(%s)" % self._print(expr.args[0])

    def _print_Abs(self, expr):
        return "od_abs(%s)" % self._print(expr.args[0])


class od_rootinv(sp.Function):
    """od_rootinv(b, Q) = b**(-1/Q) for b > 0 (csrc/od_math.h: single-precision seed + division-free Newton steps)"""
    nargs = 2
    is_real = True


def rewrite_roots(exprs):
    """Fractional powers b**(p/q) (q > 2, or q = 2 with |p| > 1) of one base through ONE root u = b**(-1/Q), Q the least
    common denominator over the base's exponents:  b**(P/Q) = u**(-P) for P < 0,  b**c * u**(c Q - P) for P > 0
    (c = ceil(P/Q)).  The signed distance of the planar push, (x**10 + y**10)**(1/10), and its first and second
    derivatives need exponents 1/10, -9/10, -9/5, -19/10, -14/5 of the same sum: one root instead of five calls of the
    library's pow per Jacobian evaluation (each a few hundred instructions)."""
    import math
    exprs = [sp.sympify(e) for e in exprs]

    def frac(e):
        return e.is_Pow and e.exp.is_Rational and not e.exp.is_Integer and not (e.exp.q == 2 and abs(e.exp.p) == 1)

    # innermost first: the base of an outer fractional power changes when the roots inside it are rewritten
    for _ in range(8):
        Q = {}
        for e in exprs:
            for pw in e.atoms(sp.Pow):
                if frac(pw) and not any(frac(x) for x in pw.base.atoms(sp.Pow)):
                    Q[pw.base] = math.lcm(Q.get(pw.base, 1), int(pw.exp.q))
        if not Q:
            return exprs

        def sub(pw):
            if pw.base not in Q:
                return pw
            b, q = pw.base, Q[pw.base]
            P = int(pw.exp.p) * (q // int(pw.exp.q))
            u = od_rootinv(b, sp.Integer(q))
            if P < 0:
                return u ** (-P)
            c = -((-P) // q)
            return b ** c * u ** (c * q - P)

        exprs = [e.replace(lambda x: frac(x) and x.base in Q, sub) for e in exprs]
    raise AssertionError("nested fractional powers deeper than expected")


def _cse_block(exprs: Sequence[sp.Expr], printer, scalar: str, prefix: str) -> Tuple[List[str], List[str]]:
    """CSE the expressions; returns (statement lines, printed result expressions)."""
    syms = sp.numbered_symbols(prefix)
    repl, red = sp.cse(list(exprs), symbols=syms)
    lines = []
    for s, e in repl:
        lines.append("const %s %s = %s;" % (scalar, s.name, printer.doprint(e)))
    outs = [printer.doprint(e) for e in red]
    return lines, outs


def count_ops(exprs: Sequence[sp.Expr]) -> int:
    repl, red = sp.cse(list(exprs))
    return int(sum(sp.count_ops(e) for _, e in repl) + sum(sp.count_ops(e) for e in red))


# --------------------------------------------------------------------------------------
# symbolic derivation shared by both flavours
# --------------------------------------------------------------------------------------
class Derived:
    def __init__(self, m: ModelSpec):
        self.m = m
        r = sp.Matrix(m.r)
        self.r = r
        self.rz = r.jacobian(sp.Matrix(m.z))
        self.rth = r.jacobian(sp.Matrix(m.th))
        # kappa enters linearly with coefficient -1 on the cone "head" rows only
        dk = r.diff(m.kappa)
        self.kappa_rows = [i for i in range(m.nz) if dk[i] != 0]
        heads = list(m.ortr) + [s[0] for s in m.socri]
        assert sorted(self.kappa_rows) == sorted(heads), (m.name, self.kappa_rows, heads)
        for i in self.kappa_rows:
            assert dk[i] == -1
        self.r0 = r.subs(m.kappa, 0)
        self.rz_nz = [(i, j) for i in range(m.nz) for j in range(m.nz) if self.rz[i, j] != 0]
        # gradient columns of theta actually consumed by the callers
        if m.kind == "mech":
            self.grad_cols = list(range(2 * m.nq + m.nu))
        elif m.kind == "rocket":
            self.grad_cols = list(range(m.nq + m.nu))
        else:
            self.grad_cols = list(range(3))
        self.rth_nz = [(i, j) for j in self.grad_cols for i in range(m.nz) if self.rth[i, j] != 0]
        # can the regularisation clamp (z_ort -> max(z_ort, reg) inside rz) be applied after
        # evaluation?  yes iff orthant variables enter rz only through their own bilinear rows.
        ortv = set(m.ort[0]) | set(m.ort[1])
        ok = True
        self.ort_entries = []  # (nnz index, partner variable index) for bilinear-row entries
        for k, (i, j) in enumerate(self.rz_nz):
            e = self.rz[i, j]
            deps = {m.z.index(s) for s in e.free_symbols if s in m.z}
            if deps & ortv:
                if i in m.ortr and e in [m.z[v] for v in ortv]:
                    self.ort_entries.append((k, m.z.index(e)))
                else:
                    ok = False
        self.reg_posthoc = ok


# --------------------------------------------------------------------------------------
# oracle flavour
# --------------------------------------------------------------------------------------
def emit_oracle(m: ModelSpec, d: Derived) -> str:
    pr = _OraclePrinter()
    o = io.StringIO()
    n = m.name
    o.write("/* GENERATED by optimization_dynamics_amd.codegen (oracle flavour) -- do not edit.\n")
    o.write(" * Model %s: nz=%d ntheta=%d.  Test infrastructure only. */\n" % (n, m.nz, m.nth))

    def func(sig, exprs, outname, dense_idx=None, zero=None):
        o.write("static void %s {\n" % sig)
        if zero:
            o.write("  for (int i_ = 0; i_ < %d; ++i_) %s[i_] = 0.0;\n" % (zero, outname))
        lines, outs = _cse_block(exprs, pr, "double", "x")
        for ln in lines:
            o.write("  " + ln + "\n")
        for k, e in enumerate(outs):
            idx = dense_idx[k] if dense_idx is not None else k
            o.write("  %s[%d] = %s;\n" % (outname, idx, e))
        o.write("}\n\n")

    def S(e):
        return e

    func("%s_r(const double* z, const double* th, double kappa, double* r)" % n,
         [S(e) for e in d.r], "r")
    func("%s_rz(const double* z, const double* th, double* rz)" % n,
         [S(d.rz[i, j]) for (i, j) in d.rz_nz], "rz",
         dense_idx=[i + m.nz * j for (i, j) in d.rz_nz], zero=m.nz * m.nz)
    nzth = [(i, j) for j in range(m.nth) for i in range(m.nz) if d.rth[i, j] != 0]
    func("%s_rth(const double* z, const double* th, double* rth)" % n,
         [S(d.rth[i, j]) for (i, j) in nzth], "rth",
         dense_idx=[i + m.nz * j for (i, j) in nzth], zero=m.nz * m.nth)
    return o.getvalue()


def _c_int_list(xs):
    return "{" + ", ".join(str(int(x)) for x in xs) + "}" if len(xs) else "{0}"


def emit_oracle_table(m: ModelSpec) -> str:
    """Static description consumed by oracle/ip_oracle.c (struct od_oracle_model)."""
    n = m.name
    o = io.StringIO()
    soc_flat_p, soc_flat_d, soc_off, socr_flat = [], [], [0], []
    for (p, dd), rr in zip(m.soc, m.socri):
        soc_flat_p += p
        soc_flat_d += dd
        socr_flat += rr
        soc_off.append(len(soc_flat_p))
    zi_kind = [0 if isinstance(e, tuple) else 1 for e in m.z_init]
    zi_idx = [e[1] if isinstance(e, tuple) else 0 for e in m.z_init]
    zi_val = [0.0 if isinstance(e, tuple) else float(e) for e in m.z_init]
    o.write("static const int %s_ort1[] = %s;\n" % (n, _c_int_list(m.ort[0])))
    o.write("static const int %s_ort2[] = %s;\n" % (n, _c_int_list(m.ort[1])))
    o.write("static const int %s_ortr[] = %s;\n" % (n, _c_int_list(m.ortr)))
    o.write("static const int %s_soc1[] = %s;\n" % (n, _c_int_list(soc_flat_p)))
    o.write("static const int %s_soc2[] = %s;\n" % (n, _c_int_list(soc_flat_d)))
    o.write("static const int %s_socr[] = %s;\n" % (n, _c_int_list(socr_flat)))
    o.write("static const int %s_socoff[] = %s;\n" % (n, _c_int_list(soc_off)))
    o.write("static const int %s_equr[] = %s;\n" % (n, _c_int_list(m.equr)))
    o.write("static const int %s_bil[] = %s;\n" % (n, _c_int_list(m.bil)))
    o.write("static const int %s_zq[] = %s;\n" % (n, _c_int_list(m.idx_zq)))
    o.write("static const int %s_zikind[] = %s;\n" % (n, _c_int_list(zi_kind)))
    o.write("static const int %s_ziidx[] = %s;\n" % (n, _c_int_list(zi_idx)))
    o.write("static const double %s_zival[] = {%s};\n" % (n, ", ".join(repr(v) for v in zi_val)))
    fd = m.fric_default if m.fric_default else [0.0]
    o.write("static const double %s_fric[] = {%s};\n" % (n, ", ".join(repr(float(v)) for v in fd)))
    kind = {"mech": 0, "rocket": 1, "proj": 2}[m.kind]
    op = m.opts
    und = "INFINITY" if op["undercut"] == float("inf") else repr(float(op["undercut"]))
    o.write("static const od_oracle_model %s_model = {\n" % n)
    o.write('  "%s", %d, %d, %d, %d, %d, %d, %d,\n' % (n, m.model_id, kind, m.nq, m.nu, m.nz, m.nth, m.nfric))
    o.write("  %d, %s_ort1, %s_ort2, %s_ortr,\n" % (len(m.ort[0]), n, n, n))
    o.write("  %d, %s_socoff, %s_soc1, %s_soc2, %s_socr,\n" % (len(m.soc), n, n, n, n))
    o.write("  %d, %s_equr, %d, %s_bil, %d, %s_zq,\n" % (len(m.equr), n, len(m.bil), n, len(m.idx_zq), n))
    o.write("  %s_zikind, %s_ziidx, %s_zival, %s_fric,\n" % (n, n, n, n))
    o.write("  {%r, %r, %r, %d, %d, %r, %r, %r, %s},\n" % (
        op["r_tol"], op["kappa_tol"], op["kappa_grad_tol"], op["max_iter"], op["max_ls"],
        op["eps_min"], op["kappa_reg"], op["gamma_reg"], und))
    o.write("  %s_r, %s_rz, %s_rth\n};\n\n" % (n, n, n))
    return o.getvalue()


# --------------------------------------------------------------------------------------
# device flavour
# --------------------------------------------------------------------------------------
def _lit(c):
    return "T(%r)" % float(c)


class _Elim:
    """Static-order sparse Gaussian elimination on the structural pattern of rz, preceded by
    runtime row/column role swaps for the second-order-cone blocks, followed by a dense tail.

    Entries whose value is a compile-time constant (the +-1 of slack / psi / tangential-velocity rows,
    friction-free zeros, ...) are tracked and folded: no reciprocal, no multiply, no factor slot."""

    def __init__(self, nz, pattern, order, floor_pivots=(), swaps=(), tail_last=(), const_entries=None):
        self.nz = nz
        self.order = list(order)
        self.swaps = list(swaps)
        pat = set(pattern)
        cval = dict(const_entries or {})          # (i,j) -> float for compile-time constant entries
        self.init_zero = set()
        self.swap_lines: List[str] = []
        S = self.swap_lines
        for k, ((ra, rb), (ca, cb)) in enumerate(self.swaps):
            assert (ra, cb) in pat and (rb, ca) in pat
            S.append("const bool sw%d = od_abs(a_%d_%d) > od_abs(a_%d_%d); f.sw[%d] = sw%d;" % (k, ra, cb, rb, ca, k, k))
            cols = sorted({j for (i, j) in pat if i in (ra, rb)})
            for j in cols:
                for i in (ra, rb):
                    if (i, j) not in pat:
                        pat.add((i, j)); self.init_zero.add((i, j)); cval[(i, j)] = 0.0
                if cval.get((ra, j)) is not None and cval.get((ra, j)) == cval.get((rb, j)):
                    continue
                cval.pop((ra, j), None); cval.pop((rb, j), None)
                S.append("{ const T u_ = a_%d_%d, w_ = a_%d_%d; a_%d_%d = sw%d ? w_ : u_; a_%d_%d = sw%d ? u_ : w_; }"
                         % (ra, j, rb, j, ra, j, k, rb, j, k))
            rows = sorted({i for (i, j) in pat if j in (ca, cb)})
            for i in rows:
                for j in (ca, cb):
                    if (i, j) not in pat:
                        pat.add((i, j)); self.init_zero.add((i, j)); cval[(i, j)] = 0.0
                if cval.get((i, ca)) is not None and cval.get((i, ca)) == cval.get((i, cb)):
                    continue
                cval.pop((i, ca), None); cval.pop((i, cb), None)
                S.append("{ const T u_ = a_%d_%d, w_ = a_%d_%d; a_%d_%d = sw%d ? w_ : u_; a_%d_%d = sw%d ? u_ : w_; }"
                         % (i, ca, i, cb, i, ca, k, i, cb, k))
        self.pattern0 = set(pat)
        self.const0 = dict(cval)                  # constants at the start of the elimination
        rows = list(range(nz))
        cols = list(range(nz))
        self.slots = 0
        self.fac_lines: List[str] = []
        # solve program: values are either ('c', float) or ('s', slot)
        self.fwd = []   # (pr, [(row i, val)])
        self.bwd = []   # (pr, pc, ipval, [(col j, val)])
        L = self.fac_lines

        def ref(i, j):
            return _lit(cval[(i, j)]) if (i, j) in cval else "a_%d_%d" % (i, j)

        for (pr, pc) in self.order:
            assert (pr, pc) in pat, ("structurally zero pivot", pr, pc)
            assert pr in rows and pc in cols
            rows.remove(pr)
            cols.remove(pc)
            L.append("{")
            if (pr, pc) in cval:
                assert cval[(pr, pc)] != 0.0
                ipc = 1.0 / cval[(pr, pc)]
                ipval = ("c", ipc)
            else:
                ipc = None
                sl = self._slot()
                if (pr, pc) in floor_pivots:
                    L.append("  const T ip_ = od_rcp(od_max(a_%d_%d, T(OD_PIVOT_FLOOR))); f.v[%d] = ip_;" % (pr, pc, sl))
                else:
                    L.append("  const T ip_ = od_rcp(a_%d_%d); f.v[%d] = ip_;" % (pr, pc, sl))
                ipval = ("s", sl)
            fw = []
            prow = [j for j in cols if (pr, j) in pat and cval.get((pr, j)) != 0.0]
            for i in rows:
                if (i, pc) not in pat or cval.get((i, pc)) == 0.0:
                    continue
                # multiplier l = a_i_pc * ip
                if (i, pc) in cval and ipc is not None:
                    lc = cval[(i, pc)] * ipc
                    lval = ("c", lc)
                    lexpr = None
                else:
                    lc = None
                    sl = self._slot()
                    if ipc is not None:
                        if ipc == 1.0:
                            lexpr0 = "a_%d_%d" % (i, pc)
                        elif ipc == -1.0:
                            lexpr0 = "-a_%d_%d" % (i, pc)
                        else:
                            lexpr0 = "a_%d_%d * %s" % (i, pc, _lit(ipc))
                    else:
                        lexpr0 = "%s * ip_" % ref(i, pc)
                    L.append("  { const T l_ = %s; f.v[%d] = l_;" % (lexpr0, sl))
                    lval = ("s", sl)
                    lexpr = "l_"
                fw.append((i, lval))
                for j in prow:
                    uc = cval.get((pr, j))
                    # product l * u
                    if lc is not None and uc is not None:
                        pc_ = lc * uc
                        if (i, j) in pat and (i, j) not in cval:
                            L.append("    a_%d_%d -= %s;" % (i, j, _lit(pc_)))
                        else:
                            cval[(i, j)] = cval.get((i, j), 0.0) - pc_
                            pat.add((i, j))
                        continue
                    if lc is not None:
                        prod = ("a_%d_%d" % (pr, j)) if lc == 1.0 else (("-a_%d_%d" % (pr, j)) if lc == -1.0 else "%s * a_%d_%d" % (_lit(lc), pr, j))
                    elif uc is not None:
                        prod = "l_" if uc == 1.0 else ("-l_" if uc == -1.0 else "l_ * %s" % _lit(uc))
                    else:
                        prod = "l_ * a_%d_%d" % (pr, j)
                    if (i, j) in pat:
                        if (i, j) in cval:      # constant becomes runtime
                            c0 = cval.pop((i, j))
                            L.append("    a_%d_%d = %s - (%s);" % (i, j, _lit(c0), prod))
                        else:
                            L.append("    a_%d_%d -= %s;" % (i, j, prod))
                    else:
                        L.append("    a_%d_%d = -(%s);" % (i, j, prod))
                        pat.add((i, j))
                if lexpr is not None:
                    L.append("  }")
            us = []
            for j in prow:
                if (pr, j) in cval:
                    us.append((j, ("c", cval[(pr, j)])))
                else:
                    sl = self._slot()
                    L.append("  f.v[%d] = a_%d_%d;" % (sl, pr, j))
                    us.append((j, ("s", sl)))
            L.append("}")
            self.fwd.append((pr, fw))
            self.bwd.append((pr, pc, ipval, us))
        # rows that share their index with a `tail_last` column (dynamics row i <-> configuration q_i) go
        # last, in that order: the natural pivots then sit on the diagonal and the runtime search seldom
        # has to exchange rows (od_lu_factor skips an exchange no lane of the wavefront needs)
        self.tail_rows = [r for r in rows if r not in tail_last] + [r for r in tail_last if r in rows]
        self.tail_cols = [c for c in cols if c not in tail_last] + [c for c in cols if c in tail_last]
        self.m = len(rows)
        self.tail_base = self.slots
        self.slots += self.m * self.m
        self.final_pattern = pat
        self.final_const = cval

    def _slot(self):
        s = self.slots
        self.slots += 1
        return s


def emit_device(m: ModelSpec, d: Derived) -> str:
    pr = _DevicePrinter()
    n = m.name
    o = io.StringIO()
    o.write("// GENERATED by optimization_dynamics_amd.codegen (device flavour) -- do not edit.\n")
    o.write("// Model %s: nz=%d ntheta=%d  (reference residual statement: see codegen/models.py)\n" % (n, m.nz, m.nth))
    o.write("#pragma once\n#include \"../od_math.h\"\n\nnamespace od {\n\n")

    const_entries = {(i, j): float(d.rz[i, j]) for (i, j) in d.rz_nz if d.rz[i, j].is_Number}
    tail_last = set(m.idx_zq) if (m.soc and m.kind == 'mech') else ()
    el = _Elim(m.nz, d.rz_nz, m.elim, set(m.floor_pivots), m.swaps, tail_last, const_entries)
    # the "state" program of the interior-point iterations (factor<false> / solve<false>): the same elimination continued
    # through the cone leftovers (m.elim_state), so that only the configuration block is left as dense tail; without
    # m.elim_state it is the first program with its tail taken down the diagonal (m.static_tail)
    el_s = None
    if m.elim_state:
        el_s = _Elim(m.nz, d.rz_nz, m.elim_state, set(m.floor_pivots), m.swaps, tail_last, const_entries)
    has_state = bool(el_s is not None or (m.static_tail and el.m > 0))
    els = el_s if el_s is not None else el
    state_tail_piv = bool(m.state_tail_pivot) if el_s is not None else False
    nnz = len(d.rz_nz)
    nnzth = len(d.rth_nz)

    def arr(name, xs, ty="int"):
        xs = list(xs)
        if not xs:
            xs = [0]
        return "  static constexpr %s %s[%d] = {%s};\n" % (ty, name, len(xs), ", ".join(str(x) for x in xs))

    soc_flat_p, soc_flat_d, soc_off, socr_flat = [], [], [0], []
    for (p, dd), rr in zip(m.soc, m.socri):
        soc_flat_p += p
        soc_flat_d += dd
        socr_flat += rr
        soc_off.append(len(soc_flat_p))
    max_soc = max([len(p) for p, _ in m.soc], default=1)

    o.write("struct Model_%s {\n" % n)
    o.write("  static constexpr int ID = %d;\n" % m.model_id)
    o.write("  static constexpr int KIND = %d;  // 0 mech (theta=[q0;q1;u;fric;h]), 1 rocket, 2 projection\n"
            % {"mech": 0, "rocket": 1, "proj": 2}[m.kind])
    o.write("  static constexpr int NQ = %d, NU = %d, NZ = %d, NTH = %d, NFRIC = %d;\n" % (m.nq, m.nu, m.nz, m.nth, m.nfric))
    o.write("  static constexpr int NNZ = %d, NNZTH = %d, NGC = %d;\n" % (nnz, nnzth, len(d.grad_cols)))
    o.write("  static constexpr int NORT = %d, NSOC = %d, MAXSOC = %d, NEQ = %d, NBIL = %d, NZQ = %d;\n"
            % (len(m.ort[0]), len(m.soc), max_soc, len(m.equr), len(m.bil), len(m.idx_zq)))
    o.write("  static constexpr bool REG_POSTHOC = %s;\n" % ("true" if d.reg_posthoc else "false"))
    o.write("  static constexpr int NFACT = %d, MTAIL = %d, TAIL_BASE = %d, NSWAP = %d;\n" % (max(el.slots, els.slots), el.m, el.tail_base, len(el.swaps)))
    o.write("  // the interior-point iterations have their own elimination program, factor<false> / solve<false> (no runtime\n")
    o.write("  // pivoting beyond the cone role swaps%s); gradient solves always use factor<true>\n" % (", pivoted %dx%d configuration tail" % (els.m, els.m) if state_tail_piv else ""))
    o.write("  static constexpr bool STATIC_TAIL = %s;\n" % ("true" if has_state else "false"))
    o.write("  static constexpr int MTAIL_S = %d, TAIL_BASE_S = %d;\n" % (els.m, els.tail_base))
    o.write("  // whether the configuration block of the interior-point iterations is factored with runtime partial pivoting\n")
    o.write("  static constexpr bool STATE_TAIL_PIVOT = %s;\n" % ("true" if ((not has_state) or state_tail_piv) else "false"))
    o.write("  // factor slots the interior-point iterations touch (a kernel that never takes a gradient stores no more)\n")
    o.write("  static constexpr int NFACT_S = %d;\n" % (els.slots if has_state else el.slots))
    o.write(arr("ORT1", m.ort[0]) + arr("ORT2", m.ort[1]) + arr("ORTR", m.ortr))
    o.write(arr("SOCOFF", soc_off) + arr("SOC1", soc_flat_p) + arr("SOC2", soc_flat_d) + arr("SOCR", socr_flat))
    o.write(arr("EQUR", m.equr) + arr("BIL", m.bil) + arr("ZQ", m.idx_zq))
    o.write("  static constexpr int NGAM = %d, NBFR = %d;\n" % (len(m.idx_gamma), len(m.idx_b)))
    o.write(arr("GAM", m.idx_gamma) + arr("BFR", m.idx_b))
    o.write(arr("KROWS", d.kappa_rows))
    o.write("  static constexpr int NKROWS = %d;\n" % len(d.kappa_rows))
    # posthoc clamp entries: (nnz slot, partner variable)
    o.write("  static constexpr int NORTENT = %d;\n" % len(d.ort_entries))
    o.write(arr("ORTENT_SLOT", [k for k, _ in d.ort_entries]) + arr("ORTENT_VAR", [v for _, v in d.ort_entries]))
    zi_kind = [0 if isinstance(e, tuple) else 1 for e in m.z_init]
    zi_idx = [e[1] if isinstance(e, tuple) else 0 for e in m.z_init]
    zi_val = [0.0 if isinstance(e, tuple) else float(e) for e in m.z_init]
    o.write(arr("ZI_KIND", zi_kind) + arr("ZI_IDX", zi_idx) + arr("ZI_VAL", [repr(v) for v in zi_val], "double"))
    o.write(arr("FRIC_DEFAULT", [repr(float(v)) for v in (m.fric_default or [0.0])], "double"))
    # rtheta sparse structure (row, column-position within grad cols)
    o.write(arr("RTH_ROW", [i for i, _ in d.rth_nz]) + arr("RTH_COL", [d.grad_cols.index(j) for _, j in d.rth_nz]))
    op = m.opts
    o.write("  static constexpr double DEF_R_TOL = %r, DEF_KAPPA_EVAL = %r, DEF_KAPPA_GRAD = %r;\n"
            % (op["r_tol"], op["kappa_tol"], op["kappa_grad_tol"]))
    o.write("  static constexpr double DEF_EPS_MIN = %r, DEF_KAPPA_REG = %r, DEF_GAMMA_REG = %r;\n"
            % (op["eps_min"], op["kappa_reg"], op["gamma_reg"]))
    und = "1e300" if op["undercut"] == float("inf") else repr(float(op["undercut"]))
    o.write("  static constexpr double DEF_UNDERCUT = %s;  // 1e300 stands for Inf\n" % und)
    o.write("  static constexpr int DEF_MAX_ITER = %d, DEF_MAX_LS = %d;\n" % (op["max_iter"], op["max_ls"]))
    o.write('  static constexpr const char* NAME = "%s";\n\n' % n)

    # ---- staged evaluation --------------------------------------------------------------------
    # One joint CSE over r(z;theta,0), the nonzeros of rz and of rtheta.  Temporaries are split into
    #   * theta-only ("pre"):   computed once per problem by eval_pre (the q0/q1 half of the discrete
    #                           Euler-Lagrange residual, 1/h, ... are loop invariants of the IP loop)
    #   * trigonometric, z-dependent ("tr"): computed by eval_r at the point it is called on and
    #                           re-used by eval_rz / eval_rth at the same point (the Jacobian is always
    #                           evaluated at the last accepted line-search candidate)
    #   * everything else: recomputed inside the function that needs it.
    r0 = list(d.r0)
    rzv = [d.rz[i, j] for (i, j) in d.rz_nz]
    rthv = [d.rth[i, j] for (i, j) in d.rth_nz]
    # split every residual row into its theta-only addends (one stored value per row: e.g. the
    # (q0, q1) half of the discrete Euler-Lagrange equations) and the z-dependent rest
    row_pre = []          # (symbol, theta-only expression)
    r_loop = []
    for i, e in enumerate(r0):
        indep, dep = e.as_independent(*m.z, as_Add=True)
        if sp.count_ops(indep) >= 2:
            ps = sp.Symbol("pq%d" % i, real=True)
            row_pre.append((ps, indep))
            r_loop.append(ps + dep)
        else:
            r_loop.append(e)
    npre_rows = len(row_pre)
    repl, red = sp.cse(rewrite_roots([e for _, e in row_pre] + r_loop + rzv + rthv), symbols=sp.numbered_symbols("x"))
    red_rowpre = red[:npre_rows]
    red = red[npre_rows:]
    red_r, red_rz, red_rth = red[:m.nz], red[m.nz:m.nz + nnz], red[m.nz + nnz:]
    zs = set(m.z)
    ortv = {m.z[i] for i in (set(m.ort[0]) | set(m.ort[1]))}
    tdef = {sy: e for sy, e in repl}
    dep_z, dep_ort = {}, {}
    for sy, e in repl:
        fs = e.free_symbols
        dep_z[sy] = any((f in zs) or dep_z.get(f, False) for f in fs)
        dep_ort[sy] = any((f in ortv) or dep_ort.get(f, False) for f in fs)
    # z-dependent transcendental values computed by eval_r at the point it is called on and re-used by eval_rz / eval_rth
    # at the same point: sines, cosines and roots
    is_trig = {sy: (dep_z[sy] and isinstance(e, (sp.sin, sp.cos, od_rootinv))) for sy, e in repl}
    for sy in tdef:
        if is_trig[sy]:
            assert not dep_ort[sy], "trig of a clamped (orthant) variable cannot be shared between r and rz"

    def closure(exprs):
        need, stack = set(), [f for e in exprs for f in e.free_symbols if f in tdef]
        while stack:
            t = stack.pop()
            if t in need:
                continue
            need.add(t)
            stack += [f for f in tdef[t].free_symbols if f in tdef]
        return need

    need_any = closure(list(red_r) + list(red_rz) + list(red_rth))
    trig_all = [sy for sy, _ in repl if is_trig[sy] and sy in need_any]
    # theta-only temporaries: the expensive ones are computed once (eval_pre -> pre[]), the cheap ones
    # (a couple of flops) are recomputed where needed -- every stored value pins two VGPRs for the whole
    # interior-point loop.
    PRE_THRESH = 16

    def _w(e):
        c = int(sp.count_ops(e))
        c += 20 * len(e.atoms(sp.sin, sp.cos, od_rootinv))
        c += 3 * sum(1 for p_ in e.atoms(sp.Pow) if p_.exp.is_negative)
        return c

    full_cost = {}
    for sy, e in repl:
        if not dep_z[sy]:
            full_cost[sy] = _w(e) + sum(full_cost[f] for f in e.free_symbols if f in full_cost)
    stored = {sy for sy in full_cost if full_cost[sy] >= PRE_THRESH}
    order = {sy: k for k, (sy, _) in enumerate(repl)}

    def consumer_need(outs, through_trig=False):
        """temporaries a function must compute itself: stop at stored theta-only values (pre[]) and,
        unless it is the producer, at shared trig values (tr[])"""
        need, stack = set(), [f for e in outs for f in e.free_symbols if f in tdef]
        while stack:
            t = stack.pop()
            if t in need:
                continue
            need.add(t)
            if t in stored or (is_trig[t] and not through_trig):
                continue
            stack += [f for f in tdef[t].free_symbols if f in tdef]
        return need

    need_r = consumer_need(list(red_r) + [tdef[t] for t in trig_all] + list(trig_all), through_trig=True)
    need_rz2 = consumer_need(red_rz)
    # only the loop functions (r, rz) justify pinning registers; rtheta is evaluated once per problem
    # and recomputes whatever theta-only values r/rz did not ask for
    stored = {t for n_ in (need_r, need_rz2) for t in n_ if t in stored}
    need_rth2 = consumer_need(red_rth)
    pre_live = sorted(stored, key=lambda t: order[t])
    pre_idx = {t: k for k, t in enumerate(pre_live)}
    row_idx = {ps: len(pre_live) + k for k, (ps, _) in enumerate(row_pre)}
    tr_idx = {t: k for k, t in enumerate(trig_all)}
    pre_need = closure([tdef[t] for t in pre_live] + list(red_rowpre)) | set(pre_live)

    o.write("  static constexpr int NPRE = %d, NTR = %d;\n\n" % (max(1, len(pre_live) + npre_rows), max(1, len(trig_all))))
    o.write("  // theta-only work, once per problem: expensive shared subexpressions + the theta-only addends of each residual row\n")
    o.write("  template <class T> OD_HD static void eval_pre(const T* th, T* pre) {\n")
    for sy, e in repl:
        if sy in pre_need:
            o.write("    const T %s = %s;\n" % (sy.name, pr.doprint(e)))
    for t in pre_live:
        o.write("    pre[%d] = %s;\n" % (pre_idx[t], t.name))
    for (ps, _), e in zip(row_pre, red_rowpre):
        o.write("    pre[%d] = %s;\n" % (row_idx[ps], pr.doprint(e)))
    o.write("  }\n\n")

    def body(need, outs, out_name, produce_trig):
        rows_used = sorted({f for e in outs for f in e.free_symbols if f in row_idx}, key=lambda t: row_idx[t])
        for ps in rows_used:
            o.write("    const T %s = pre[%d];\n" % (ps.name, row_idx[ps]))
        done = set()

        def emit_temp(sy, e):
            if sy in stored:
                o.write("    const T %s = pre[%d];\n" % (sy.name, pre_idx[sy]))
            elif is_trig[sy] and not produce_trig:
                o.write("    const T %s = tr[%d];\n" % (sy.name, tr_idx[sy]))
            elif is_trig[sy] and sy in trig_slot:
                k, fn = trig_slot[sy]
                o.write("    const T %s = %s[%d];\n" % (sy.name, "ts_" if fn == "sin" else "tc_", k))
                o.write("    tr[%d] = %s;\n" % (tr_idx[sy], sy.name))
            else:
                o.write("    const T %s = %s;\n" % (sy.name, pr.doprint(e)))
                if is_trig[sy]:
                    o.write("    tr[%d] = %s;\n" % (tr_idx[sy], sy.name))
            done.add(sy)

        trig_slot = {}
        if produce_trig:
            # the sines / cosines of z-dependent angles, evaluated together through the policy TP: the cooperative
            # solver computes two angles side by side in the two halves of a lane row (od_coop.h::TrigHalves)
            targs = []
            for sy, e in repl:
                if sy in need and is_trig[sy] and sy not in stored and isinstance(e, (sp.sin, sp.cos)):
                    a_ = e.args[0]
                    if a_ not in targs:
                        targs.append(a_)
                    trig_slot[sy] = (targs.index(a_), "sin" if isinstance(e, sp.sin) else "cos")
            if len(targs) >= 2:
                first = set()
                stack = [f for a_ in targs for f in a_.free_symbols if f in tdef]
                while stack:
                    t = stack.pop()
                    if t in first:
                        continue
                    first.add(t)
                    assert not is_trig[t], "an angle that depends on another sine / cosine"
                    if t not in stored:
                        stack += [f for f in tdef[t].free_symbols if f in tdef]
                for sy, e in repl:
                    if sy in first:
                        emit_temp(sy, e)
                o.write("    T ts_[%d], tc_[%d];\n" % (len(targs), len(targs)))
                o.write("    { const T ta_[%d] = {%s}; TP::template sincos_n<%d>(ta_, ts_, tc_); }\n"
                        % (len(targs), ", ".join(pr.doprint(a_) for a_ in targs), len(targs)))
            else:
                trig_slot = {}
        for sy, e in repl:
            if sy not in need or sy in done:
                continue
            emit_temp(sy, e)
        for k, e in enumerate(outs):
            o.write("    %s[%d] = %s;\n" % (out_name, k, pr.doprint(e)))

    o.write("  // r(z; theta, kappa = 0); also produces the shared trigonometric values tr[] at z\n")
    o.write("  template <class TP = TrigDirect, class T> OD_HD static void eval_r(const T* z, const T* th, const T* pre, T* tr, T* r) {\n")
    body(need_r, red_r, "r", True)
    o.write("  }\n\n")
    o.write("  // structural nonzeros of rz at z (orthant variables possibly clamped); tr[] from eval_r at the same point\n")
    o.write("  template <class T> OD_HD static void eval_rz(const T* z, const T* th, const T* pre, const T* tr, T* a) {\n")
    body(need_rz2, red_rz, "a", False)
    o.write("  }\n\n")
    o.write("  template <class T> OD_HD static void eval_rth(const T* z, const T* th, const T* pre, const T* tr, T* g) {\n")
    body(need_rth2, red_rth, "g", False)
    o.write("  }\n\n")
    def _loop_ops(need, outs):
        return int(sum(sp.count_ops(tdef[t]) for t in need if t not in stored and not (is_trig[t] and need is not need_r))
                   + sum(sp.count_ops(e) for e in outs))

    d.stage_stats = dict(ops_pre=int(sum(sp.count_ops(tdef[t]) for t in pre_need)),
                         ops_r_loop=_loop_ops(need_r, red_r), ops_rz_loop=_loop_ops(need_rz2, red_rz),
                         ops_rth_loop=_loop_ops(need_rth2, red_rth), n_pre=len(pre_live) + npre_rows, n_trig=len(trig_all))

    # ---- factor ----
    o.write("  template <class T> struct Fact { T v[NFACT > 0 ? NFACT : 1]; int piv[MTAIL > 0 ? MTAIL : 1]; bool sw[NSWAP > 0 ? NSWAP : 1]; };\n\n")

    def factor_body(e, mode, ind):
        """mode: 'piv' runtime-pivoted tail | 'static' tail down the diagonal"""
        w = lambda t: o.write(ind + t + "\n")
        for k, (i, j) in enumerate(d.rz_nz):
            w("T a_%d_%d = a[%d];" % (i, j, k))
        for (i, j) in sorted(e.final_pattern - set(d.rz_nz)):
            w("T a_%d_%d = T(0);" % (i, j))
        for ln in e.swap_lines:
            w(ln)
        for ln in e.fac_lines:
            w(ln)
        if e.m > 0:
            w("T tl_[%d];" % (e.m * e.m))
        for ii, i in enumerate(e.tail_rows):
            for jj, j in enumerate(e.tail_cols):
                if (i, j) in e.final_const:
                    w("tl_[%d] = %s;" % (ii + e.m * jj, _lit(e.final_const[(i, j)])))
                elif (i, j) in e.final_pattern:
                    w("tl_[%d] = a_%d_%d;" % (ii + e.m * jj, i, j))
                else:
                    w("tl_[%d] = T(0);" % (ii + e.m * jj))
        if e.m > 0:
            if mode == "piv":
                w("const bool ok_ = od_lu_factor<T, %d>(tl_, f.piv);" % e.m)
            else:
                w("const bool ok_ = od_lu_factor_static<T, %d>(tl_);" % e.m)
            o.write("#pragma unroll\n")
            w("for (int i = 0; i < %d; ++i) f.v[%d + i] = tl_[i];" % (e.m * e.m, e.tail_base))
            w("return ok_;")
        else:
            w("return true;")

    o.write("  // factor<true>: statically ordered sparse elimination (%d pivots) + %dx%d runtime-pivoted dense tail\n" % (len(m.elim), el.m, el.m))
    if has_state:
        o.write("  // factor<false>: %d static pivots + %dx%d %s tail (interior-point iterations)\n"
                % (len(els.order), els.m, els.m, "runtime-pivoted" if state_tail_piv else "unpivoted"))
    o.write("  template <bool PIV = true, class T, class F> OD_HD static bool factor(const T* a, F& f) {\n")
    if has_state:
        o.write("    if constexpr (PIV) {\n")
        factor_body(el, "piv", "      ")
        o.write("    } else {\n")
        factor_body(els, "piv" if state_tail_piv else "static", "      ")
        o.write("    }\n")
    else:
        factor_body(el, "piv", "    ")
    o.write("  }\n\n")

    # ---- solve ----
    def _mul(val, operand):
        """code for val * operand; val = ('c', float) | ('s', slot)"""
        if val[0] == "s":
            return "f.v[%d] * %s" % (val[1], operand)
        c = val[1]
        if c == 1.0:
            return operand
        if c == -1.0:
            return "-%s" % operand
        return "%s * %s" % (_lit(c), operand)

    def solve_body(e, mode, ind):
        w = lambda t: o.write(ind + t + "\n")
        for i in range(m.nz):
            w("T y_%d = b[%d];" % (i, i))
        for k, ((ra, rb), (ca, cb)) in enumerate(e.swaps):
            w("{ const T u_ = y_%d, w_ = y_%d; y_%d = f.sw[%d] ? w_ : u_; y_%d = f.sw[%d] ? u_ : w_; }" % (ra, rb, ra, k, rb, k))
        for (prw, fw) in e.fwd:
            for (i, lval) in fw:
                if lval[0] == "c" and lval[1] == 0.0:
                    continue
                w("y_%d -= %s;" % (i, _mul(lval, "y_%d" % prw)))
        if e.m > 0:
            w("T t_[%d];" % e.m)
            for ii, i in enumerate(e.tail_rows):
                w("t_[%d] = y_%d;" % (ii, i))
            if mode == "piv":
                w("od_lu_solve<T, %d>(od_tail_view<T, %d>(f), f.piv, t_);" % (e.m, e.tail_base))
            else:
                w("od_lu_solve_static<T, %d>(od_tail_view<T, %d>(f), t_);" % (e.m, e.tail_base))
            for jj, j in enumerate(e.tail_cols):
                w("const T x_%d = t_[%d];" % (j, jj))
        for (prw, pc, ipval, us) in reversed(e.bwd):
            terms = "".join(" - (%s)" % _mul(uv, "x_%d" % j) for (j, uv) in us if not (uv[0] == "c" and uv[1] == 0.0))
            w("const T x_%d = %s;" % (pc, _mul(ipval, "(y_%d%s)" % (prw, terms))))
        swapped = {}
        for k, ((ra, rb), (ca, cb)) in enumerate(e.swaps):
            swapped[ca] = (cb, k)
            swapped[cb] = (ca, k)
        for j in range(m.nz):
            if j in swapped:
                w("x[%d] = f.sw[%d] ? x_%d : x_%d;" % (j, swapped[j][1], swapped[j][0], j))
            else:
                w("x[%d] = x_%d;" % (j, j))

    o.write("  // x = rz^{-1} b using the stored factors of factor<PIV> (b and x may alias)\n")
    o.write("  template <bool PIV = true, class T, class F> OD_HD static void solve(const F& f, const T* b, T* x) {\n")
    if has_state:
        o.write("    if constexpr (PIV) {\n")
        solve_body(el, "piv", "      ")
        o.write("    } else {\n")
        solve_body(els, "piv" if state_tail_piv else "static", "      ")
        o.write("    }\n")
    else:
        solve_body(el, "piv", "    ")
    o.write("  }\n")
    o.write("};\n\n}  // namespace od\n")
    return o.getvalue()


def stats(m: ModelSpec, d: Derived) -> Dict[str, int]:
    extra = getattr(d, "stage_stats", {})
    return dict(extra, 
        nz=m.nz, nth=m.nth, nnz_rz=len(d.rz_nz), nnz_rth=len(d.rth_nz),
        ops_r=count_ops(list(d.r0)), ops_rz=count_ops([d.rz[i, j] for (i, j) in d.rz_nz]),
        ops_r_rz=count_ops(list(d.r0) + [d.rz[i, j] for (i, j) in d.rz_nz]),
        ops_rth=count_ops([d.rth[i, j] for (i, j) in d.rth_nz]),
    )
