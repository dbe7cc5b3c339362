"""Model generator: symbolic residual -> device code + oracle code + every table the build needs.

    python -m optimization_dynamics_amd.codegen                      # regenerate every registered model
    python -m optimization_dynamics_amd.codegen hopper acrobot_impact # only these
    python -m optimization_dynamics_amd.codegen --add path/to/spec.py # register a NEW model and generate it

Counterpart of the reference's build step (deps/build.jl:27-48 running src/models/*/codegen.jl: Symbolics trace ->
jacobians -> build_function -> cached expressions).  A spec file defines `def spec() -> ModelSpec` (see
codegen/models.py for the eight models of the reference and docs in README.md); `--add` copies it into
codegen/user_models/, gives the model the next free id and writes, besides csrc/gen/<name>.h and oracle/gen/<name>.h:
the translation unit csrc/od_model_<name>.hip, the make variable csrc/gen/models.mk, the registries
csrc/gen/model_list.h (launch tables, id -> model) and oracle/gen/models_gen.h.  Nothing else has to be edited:
`make` (or __graft_entry__.build()) then builds the library with the new model, the host side finds it by name
(od_model_id / Library.model_ids())."""
import argparse
import json
import os
import shutil
import sys
import time

from . import models as M
from .coop import emit_coop
from .coop3 import emit_coop3
from .emit import Derived, emit_device, emit_oracle, emit_oracle_table, stats

DEFAULT_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

TU = """#define OD_MODEL %s
#define OD_MODEL_MECH 1
#define OD_MODEL_FP32 0
#include "od_model_tu.inc"
"""


def main(argv):
    ap = argparse.ArgumentParser(prog="python -m optimization_dynamics_amd.codegen")
    ap.add_argument("names", nargs="*", help="models to regenerate (default: all registered)")
    ap.add_argument("--add", action="append", default=[], metavar="SPEC.py", help="register and generate a new model (defines spec() -> ModelSpec)")
    ap.add_argument("--root", default=DEFAULT_ROOT, help="repository root to write into (default: this checkout)")
    ap.add_argument("--add-constraint", action="append", default=[], metavar="SPEC.py",
                    help="register and generate a nonlinear constraint function for the device-resident iLQR solver (defines constraint() -> ConstraintSpec; codegen/constraints.py)")
    ap.add_argument("--constraints", action="store_true", help="regenerate the constraint functions only")
    args = ap.parse_args(argv[1:])
    root = os.path.abspath(args.root)
    if args.add_constraint or args.constraints:
        from . import constraints as CN
        cud = os.path.join(root, "optimization_dynamics_amd", "codegen", "user_models")
        new = [CN.register(pth, cud) for pth in args.add_constraint]
        order = CN.generate(root)
        if new:
            print("registered constraint(s): %s (ids %s).  Rebuild: make -C optimization_dynamics_amd/csrc" % (", ".join(new), ", ".join(str(order.index(n)) for n in new)))
        if not args.names and not args.add:
            return
    udir = os.path.join(root, "optimization_dynamics_amd", "codegen", "user_models")
    dev_dir = os.path.join(root, "optimization_dynamics_amd", "csrc", "gen")
    csrc = os.path.dirname(dev_dir)
    ora_dir = os.path.join(root, "oracle", "gen")
    os.makedirs(dev_dir, exist_ok=True)
    os.makedirs(ora_dir, exist_ok=True)

    added = []
    for path in args.add:
        added.append(M.register_user_model(path, udir))
    all_models = M.all_models(udir)
    names = list(args.names) + [n for n in added if n not in args.names]
    if not names:
        names = list(all_models)

    all_stats = {}
    stats_path = os.path.join(dev_dir, "stats.json")
    if os.path.exists(stats_path):
        all_stats = json.load(open(stats_path))
    for name in names:
        t0 = time.time()
        m = all_models[name]()
        assert m.name == name, (m.name, name)
        d = Derived(m)
        with open(os.path.join(dev_dir, name + ".h"), "w") as f:
            f.write(emit_device(m, d))
        with open(os.path.join(ora_dir, name + ".h"), "w") as f:
            f.write(emit_oracle(m, d))
            f.write(emit_oracle_table(m))
        # cooperative (16 lanes per problem) solver glue, for the models whose block structure allows it
        coop = emit_coop(m, d)
        coop_path = os.path.join(dev_dir, "coop_" + name + ".h")
        if coop is not None:
            with open(coop_path, "w") as f:
                f.write(coop)
        elif os.path.exists(coop_path):
            os.remove(coop_path)
        # the 8-lanes-per-problem form (cones of dimension 2 and 3, csrc/od_coop3.h): for the models the 16-lane form rejects,
        # and beside it for the models with enough contacts / cones to share out (twice the problems per wavefront: larger batches)
        nroles = len(m.ort[0]) + len(m.soc)
        coop3 = emit_coop3(m, d) if (m.kind == "mech" and nroles > 0 and (coop is None or nroles >= 4)) else None
        coop3_path = os.path.join(dev_dir, "coop3_" + name + ".h")
        if coop3 is not None:
            with open(coop3_path, "w") as f:
                f.write(coop3)
        elif os.path.exists(coop3_path):
            os.remove(coop3_path)
        all_stats[name] = stats(m, d)
        print("%-24s %6.1fs  %s" % (name, time.time() - t0, all_stats[name]), flush=True)
    json.dump(all_stats, open(stats_path, "w"), indent=1, sort_keys=True)

    # ---- registries: everything downstream (Makefiles, launch-table switch, oracle table) reads these ----
    ids = {name: all_models[name]().model_id if name in M.ALL_MODELS else M.user_model_id(name, udir) for name in all_models}
    order = sorted(all_models, key=lambda n: ids[n])
    assert [ids[n] for n in order] == list(range(len(order))), "model ids must be 0..n-1: %r" % ids
    with open(os.path.join(ora_dir, "models_gen.h"), "w") as f:
        f.write("/* GENERATED -- registry of oracle models (index = model id) */\n")
        for name in order:
            f.write('#include "%s.h"\n' % name)
        f.write("static const od_oracle_model* const od_oracle_models[] = {\n")
        for name in order:
            f.write("  &%s_model,\n" % name)
        f.write("};\nstatic const int od_oracle_num_models = %d;\n" % len(order))
    with open(os.path.join(dev_dir, "all_models.h"), "w") as f:
        f.write("// GENERATED -- every device model header\n#pragma once\n")
        for name in order:
            f.write('#include "%s.h"\n' % name)
    with open(os.path.join(dev_dir, "model_list.h"), "w") as f:
        f.write("// GENERATED -- registry of device models: X(name, id)\n#pragma once\n")
        f.write("#define OD_MODEL_COUNT %d\n" % len(order))
        f.write("#define OD_FOR_EACH_MODEL(X) \\\n")
        f.write(" \\\n".join("  X(%s, %d)" % (name, ids[name]) for name in order) + "\n")
    with open(os.path.join(dev_dir, "models.mk"), "w") as f:
        f.write("# GENERATED -- model translation units of the library (one od_model_<name>.hip each)\n")
        f.write("MODELS = %s\n" % " ".join(order))
    for name in order:
        tu = os.path.join(csrc, "od_model_%s.hip" % name)
        if not os.path.exists(tu):
            with open(tu, "w") as f:
                f.write(TU % name)
    if added:
        print("registered: %s  (ids %s).  Rebuild: make -C optimization_dynamics_amd/csrc && make -C oracle" % (", ".join(added), ", ".join(str(ids[n]) for n in added)))


if __name__ == "__main__":
    main(sys.argv)
