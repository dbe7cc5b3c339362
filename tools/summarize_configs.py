"""Condense gpurun_out/prof_cfg_<tag> (tools/profile_configs.sh on the GPU box) into profiles/<tag>_config*_kernel_stats.csv and
profiles/<tag>_configs_pmc.json: per configuration and kernel the average duration (rocprofv3 --kernel-trace --stats), HBM bytes
per dispatch (2 x FETCH_SIZE + WRITE_SIZE, KB -> bytes: the gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md), wavefronts,
VALU instructions, VALU issue utilisation, wait fractions and the executed floating-point lane-slot operations, each from its own
PMC pass -- keyed by a hash of the kernel sources so that a stale record is recognisable.
usage: python tools/summarize_configs.py r4"""
import collections, csv, hashlib, json, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r4"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "prof_cfg_" + tag)
dst = os.path.join(ROOT, "profiles")


def source_hash():
    d = os.path.join(ROOT, "optimization_dynamics_amd", "csrc")
    h = hashlib.sha1()
    for dp, _, fs in sorted(os.walk(d)):
        if "build" in dp:
            continue
        for f in sorted(fs):
            if f.endswith((".h", ".hip", ".inc")):
                h.update(f.encode()); h.update(open(os.path.join(dp, f), "rb").read())
    return h.hexdigest()[:16]


def short(k):
    k = k.replace("(anonymous namespace)::", "").replace("void ", "").replace("od::", "")
    return k.split("(")[0]


def counters(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    if not os.path.exists(path):
        return {}
    for r in csv.DictReader(open(path)):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}


out = {"tag": tag, "source_hash": source_hash(), "how": "tools/profile_configs.sh: rocprofv3 --kernel-trace --stats, then one --pmc pass per counter group", "configs": {}}
for cfg in ("config2", "config3", "config5_f32", "config5_f64", "config5_hover_f32"):
    st = os.path.join(src, cfg + "_trace", "t_kernel_stats.csv")
    if not os.path.exists(st):
        continue
    shutil.copy(st, os.path.join(dst, "%s_%s_kernel_stats.csv" % (tag, cfg)))
    rows = list(csv.DictReader(open(st)))
    fetch = counters(os.path.join(src, cfg + "_fetch", "p_counter_collection.csv"))
    write = counters(os.path.join(src, cfg + "_write", "p_counter_collection.csv"))
    sq = counters(os.path.join(src, cfg + "_sq", "p_counter_collection.csv"))
    fl = counters(os.path.join(src, cfg + "_fl", "p_counter_collection.csv"))
    ks = {}
    for r in rows:
        k = short(r["Name"])
        if float(r["Percentage"]) < 1.0 or k.startswith(("__amd", "at::", "Cijk")):
            continue
        e = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "share_pct": float(r["Percentage"])}
        if k in fetch and k in write:
            e["hbm_bytes_per_dispatch"] = 1024.0 * (2.0 * fetch[k].get("FETCH_SIZE", 0.0) + write[k].get("WRITE_SIZE", 0.0))
        if k in sq:
            s = sq[k]
            e.update(wavefronts=s.get("SQ_WAVES"), valu_insts_per_wave=s["SQ_INSTS_VALU"] / max(s.get("SQ_WAVES", 1.0), 1.0),
                     valu_issue_util=s["SQ_ACTIVE_INST_VALU"] / s["SQ_WAVE_CYCLES"] if s.get("SQ_WAVE_CYCLES") else None,
                     wait_any_frac=s["SQ_WAIT_ANY"] / s["SQ_WAVE_CYCLES"] if s.get("SQ_WAVE_CYCLES") else None,
                     quad_cycles_per_valu_inst=s["SQ_WAVE_CYCLES"] / s["SQ_INSTS_VALU"] if s.get("SQ_INSTS_VALU") else None)
        if k in fl:
            f = fl[k]
            f64 = f.get("SQ_INSTS_VALU_ADD_F64", 0) + f.get("SQ_INSTS_VALU_MUL_F64", 0) + 2 * f.get("SQ_INSTS_VALU_FMA_F64", 0)
            f32 = f.get("SQ_INSTS_VALU_ADD_F32", 0) + f.get("SQ_INSTS_VALU_MUL_F32", 0) + 2 * f.get("SQ_INSTS_VALU_FMA_F32", 0)
            e["executed_lane_slot_flops_per_dispatch"] = {"f64": 64.0 * f64, "f32": 64.0 * f32}
        ks[k] = e
    out["configs"][cfg] = ks
bc = os.path.join(src, "bench_configs.json")
if os.path.exists(bc):
    shutil.copy(bc, os.path.join(dst, "%s_bench_configs.json" % tag))
json.dump(out, open(os.path.join(dst, "%s_configs_pmc.json" % tag), "w"), indent=1)
print(json.dumps(out, indent=1)[:6000])
