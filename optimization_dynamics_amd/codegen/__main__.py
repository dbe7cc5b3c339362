import json
import os
import sys
import time

from .coop import emit_coop
from .emit import Derived, emit_device, emit_oracle, emit_oracle_table, stats
from .models import ALL_MODELS

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(argv):
    names = argv[1:] or list(ALL_MODELS)
    dev_dir = os.path.join(ROOT, "optimization_dynamics_amd", "csrc", "gen")
    ora_dir = os.path.join(ROOT, "oracle", "gen")
    os.makedirs(dev_dir, exist_ok=True)
    os.makedirs(ora_dir, exist_ok=True)
    all_stats = {}
    stats_path = os.path.join(dev_dir, "stats.json")
    if os.path.exists(stats_path):
        all_stats = json.load(open(stats_path))
    for name in names:
        t0 = time.time()
        m = ALL_MODELS[name]()
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
        all_stats[name] = stats(m, d)
        print("%-24s %6.1fs  %s" % (name, time.time() - t0, all_stats[name]), flush=True)
    json.dump(all_stats, open(stats_path, "w"), indent=1, sort_keys=True)
    # registry headers
    with open(os.path.join(ora_dir, "models_gen.h"), "w") as f:
        f.write("/* GENERATED -- registry of oracle models */\n")
        for name in ALL_MODELS:
            f.write('#include "%s.h"\n' % name)
        f.write("static const od_oracle_model* const od_oracle_models[] = {\n")
        for name in ALL_MODELS:
            f.write("  &%s_model,\n" % name)
        f.write("};\nstatic const int od_oracle_num_models = %d;\n" % len(ALL_MODELS))
    with open(os.path.join(dev_dir, "all_models.h"), "w") as f:
        f.write("// GENERATED -- registry of device models\n#pragma once\n")
        for name in ALL_MODELS:
            f.write('#include "%s.h"\n' % name)
        f.write("#define OD_FOR_EACH_MODEL(X) \\\n")
        f.write(" \\\n".join("  X(%s)" % name for name in ALL_MODELS) + "\n")


if __name__ == "__main__":
    main(sys.argv)
