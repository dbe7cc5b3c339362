"""Constraint generator: a symbolic constraint function c(x; p) -> device code for c and its Jacobian dc/dx.

    python -m optimization_dynamics_amd.codegen --add-constraint path/to/spec.py     # register a new constraint and generate it
    python -m optimization_dynamics_amd.codegen --constraints                        # regenerate every registered constraint

IterativeLQR.jl differentiates the user's constraint functions with Symbolics at solver construction (iLQR.Constraint(f, nx, nu;
idx_ineq), examples/hopper.jl:264-266); the device-resident solver (od_ilqr_*) takes affine rows as matrices
(od_ilqr_set_constraints) and NONLINEAR rows as functions generated here: a spec file defines

    def constraint() -> ConstraintSpec        # name, nx, np, expr(x, p) -> list of sympy expressions

and the generator writes csrc/gen/con_<name>.h (struct Con_<name>: NC, NX, NP, eval(x, p, c, cx) with one joint common-subexpression
pass over values and Jacobian) and the registry csrc/gen/con_list.h (name -> id, od_constraint_id).  Built in: `hopper_foot`, the
first-stage constraint of examples/hopper.jl:236-249 on the initial configurations theta = [q1; q2].
"""
import importlib.util
import json
import os
from dataclasses import dataclass
from typing import Callable, List

import sympy as sp

from .emit import _DevicePrinter


@dataclass
class ConstraintSpec:
    name: str
    nx: int                       # length of the vector the rows act on
    np: int                       # parameters (data of the rows: reference points, limits)
    expr: Callable                # (x symbols, p symbols) -> list of sympy expressions, the rows c_i(x; p) (= 0)
    doc: str = ""


def hopper_foot() -> ConstraintSpec:
    """examples/hopper.jl:240-247 (stage1_con without its control limits), on theta = [q1; q2] = u[3:10], p = x1 = [q1_0; q2_0]:
         q1 - q1_0                                    (the first configuration stays where it is)
         kinematics_foot(q1) - kinematics_foot(q1_0)  (implied by the rows above; kept because the reference has them: they enter the
                                                       penalty and the multiplier update)
         kinematics_foot(q2) - kinematics_foot(q2_0)  (the foot does not move between the two initial configurations)
       kinematics_foot(q) = [q1 + q4 sin q3; q2 - q4 cos q3]  (RoboDojo hopper, as restated in codegen/models.py::hopper)"""
    def foot(q):
        return [q[0] + q[3] * sp.sin(q[2]), q[1] - q[3] * sp.cos(q[2])]

    def expr(x, p):
        qa, qb, pa, pb = x[0:4], x[4:8], p[0:4], p[4:8]
        fa, fa0, fb, fb0 = foot(qa), foot(pa), foot(qb), foot(pb)
        return [qa[i] - pa[i] for i in range(4)] + [fa[i] - fa0[i] for i in range(2)] + [fb[i] - fb0[i] for i in range(2)]

    return ConstraintSpec("hopper_foot", 8, 8, expr, "examples/hopper.jl:240-247")


BUILTIN = {"hopper_foot": hopper_foot}


def _registry(udir):
    p = os.path.join(udir, "constraints.json")
    return json.load(open(p)) if os.path.exists(p) else {"constraints": []}


def register(path, udir):
    import shutil
    spec = importlib.util.spec_from_file_location("od_user_constraint", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    c = mod.constraint()
    if not isinstance(c, ConstraintSpec) and not (hasattr(c, "name") and hasattr(c, "expr")):
        raise ValueError("constraint() of %s does not return a ConstraintSpec" % path)
    if c.name in BUILTIN:
        raise ValueError("constraint name %r is built in" % c.name)
    os.makedirs(udir, exist_ok=True)
    reg = _registry(udir)
    if c.name not in [e["name"] for e in reg["constraints"]]:
        reg["constraints"].append({"name": c.name, "file": "con_" + c.name + ".py"})
    dst = os.path.join(udir, "con_" + c.name + ".py")
    if os.path.abspath(path) != os.path.abspath(dst):
        shutil.copyfile(path, dst)
    json.dump(reg, open(os.path.join(udir, "constraints.json"), "w"), indent=1)
    return c.name


def all_constraints(udir):
    out = dict(BUILTIN)
    for e in _registry(udir)["constraints"]:
        def load(path=os.path.join(udir, e["file"])):
            spec = importlib.util.spec_from_file_location("od_user_constraint", path)
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            return mod.constraint()
        out[e["name"]] = load
    return out


def emit(c) -> str:
    x = list(sp.symbols("x0:%d" % c.nx, real=True))
    p = list(sp.symbols("p0:%d" % max(c.np, 1), real=True))
    rows = [sp.sympify(e) for e in c.expr(x, p)]
    nc = len(rows)
    if nc < 1 or nc > 16 or c.nx > 16:
        raise ValueError("1..16 rows on at most 16 variables (OD_IL_MAXCON, OD_IL_N)")
    if c.np < 0 or c.np > 64:
        raise ValueError("at most 64 parameters (the solver's constant block, od_ilqr_set_parameter_stage)")
    J = sp.Matrix(rows).jacobian(sp.Matrix(x))
    nzj = [(i, j) for j in range(c.nx) for i in range(nc) if J[i, j] != 0]
    pr = _DevicePrinter()
    repl, red = sp.cse(rows + [J[i, j] for i, j in nzj], symbols=sp.numbered_symbols("t"))
    o = []
    o.append("// GENERATED by optimization_dynamics_amd.codegen (constraints.py) -- do not edit.")
    o.append("// Constraint %s: %d rows on %d variables, %d parameters.  %s" % (c.name, nc, c.nx, c.np, getattr(c, "doc", "")))
    o.append("#pragma once")
    o.append('#include "../od_math.h"')
    o.append("namespace od {")
    o.append("struct Con_%s {" % c.name)
    o.append("  static constexpr int NC = %d, NX = %d, NP = %d;" % (nc, c.nx, c.np))
    o.append('  static constexpr const char* NAME = "%s";' % c.name)
    o.append("  // c (NC) and cx = dc/dx (NC x NX col-major; may be null)")
    o.append("  template <class T> OD_HD static void eval(const T* x, const T* p, T* c, T* cx) {")
    for i in range(c.nx):
        o.append("    const T x%d = x[%d];" % (i, i))
    used_p = set().union(*[e.free_symbols for e in rows]) & set(p)
    for i in range(c.np):
        if p[i] in used_p:
            o.append("    const T p%d = p[%d];" % (i, i))
    for s_, e in repl:
        o.append("    const T %s = %s;" % (s_.name, pr.doprint(e)))
    for i in range(nc):
        o.append("    c[%d] = %s;" % (i, pr.doprint(red[i])))
    o.append("    if (cx) {")
    o.append("      for (int i_ = 0; i_ < NC * NX; ++i_) cx[i_] = T(0);")
    for k, (i, j) in enumerate(nzj):
        o.append("      cx[%d] = %s;" % (i + nc * j, pr.doprint(red[nc + k])))
    o.append("    }")
    o.append("  }")
    o.append("};")
    o.append("}  // namespace od")
    return "\n".join(o) + "\n"


def generate(root, names=None):
    udir = os.path.join(root, "optimization_dynamics_amd", "codegen", "user_models")
    dev_dir = os.path.join(root, "optimization_dynamics_amd", "csrc", "gen")
    allc = all_constraints(udir)
    order = list(BUILTIN) + [n for n in allc if n not in BUILTIN]
    for n in (names or order):
        with open(os.path.join(dev_dir, "con_" + n + ".h"), "w") as f:
            f.write(emit(allc[n]()))
        print("constraint %-20s -> csrc/gen/con_%s.h" % (n, n), flush=True)
    with open(os.path.join(dev_dir, "con_list.h"), "w") as f:
        f.write("// GENERATED -- registry of constraint functions: X(name, id)\n#pragma once\n")
        for n in order:
            f.write('#include "con_%s.h"\n' % n)
        f.write("#define OD_CONSTRAINT_COUNT %d\n" % len(order))
        f.write("#define OD_FOR_EACH_CONSTRAINT(X) \\\n")
        f.write(" \\\n".join("  X(%s, %d)" % (n, i) for i, n in enumerate(order)) + "\n")
    return order
