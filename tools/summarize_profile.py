"""Condense gpurun_out/prof_<tag> (written by tools/profile_round.sh on the GPU box) into profiles/.
usage: python tools/summarize_profile.py r1"""
import collections, csv, json, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")
shutil.copy(os.path.join(src, "trace", "t_kernel_stats.csv"), os.path.join(dst, tag + "_kernel_stats.csv"))
shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, tag + "_bench.json"))


def per_kernel(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(os.path.join(src, d, "p_counter_collection.csv"))):
        k = r["Kernel_Name"]
        k = "k_rollout_state" if "k_rollout_state" in k else "k_grad_knots" if "k_grad_knots" in k else None
        if k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}, \
           {k: len(next(iter(cs.values()))) for k, cs in acc.items()}


out = ["# round %s -- PMC summaries for `python bench.py --steps 3 --warmup 1 --no-cpu-baseline` (hopper T=100 batch=4096, MI355X)" % tag,
       "# collected with tools/profile_round.sh: separate rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ_*), each with --kernel-trace only"]
tot = 0.0
for d, c, mult in (("pmc_fetch", "FETCH_SIZE", 2.0), ("pmc_write", "WRITE_SIZE", 1.0)):
    m, n = per_kernel(d)
    for k in ("k_rollout_state", "k_grad_knots"):
        out.append("od::%s %s per dispatch [KB]: mean %.1f  (n=%d)" % (k, c, m[k][c], n[k]))
        tot += mult * m[k][c] * 1024
m, n = per_kernel("pmc_sq")
for k in ("k_rollout_state", "k_grad_knots"):
    out.append("%s SQ counters per dispatch (SQ_*_CYCLES in quad-cycles): %s" % (k, json.dumps({c: int(v) for c, v in sorted(m[k].items())})))
out.append("HBM bytes per od_rollout (2*FETCH_SIZE + WRITE_SIZE, both kernels; FETCH_SIZE counts half of wide coalesced reads on gfx950): %.1f MB" % (tot / 1e6))
s = m["k_rollout_state"]
out.append("k_rollout_state: %d wavefronts, %.2f M VALU instructions per wavefront, %.2f quad-cycles per VALU instruction, VALU active %.0f %% of wave cycles, SQ_WAIT_ANY (dependent-issue gaps and memory waits) %.0f %%, issue stalls %.1f %%"
           % (s["SQ_WAVES"], s["SQ_INSTS_VALU"] / s["SQ_WAVES"] / 1e6, s["SQ_WAVE_CYCLES"] / s["SQ_INSTS_VALU"],
              100 * s["SQ_ACTIVE_INST_VALU"] / s["SQ_WAVE_CYCLES"], 100 * s["SQ_WAIT_ANY"] / s["SQ_WAVE_CYCLES"],
              100 * s["SQ_WAIT_INST_ANY"] / s["SQ_WAVE_CYCLES"]))
if os.path.exists(os.path.join(src, "pmc_sq2", "p_counter_collection.csv")):
    m2, _ = per_kernel("pmc_sq2")
    s2 = m2["k_rollout_state"]
    out.append("k_rollout_state, second SQ pass per dispatch: %s" % json.dumps({c: int(v) for c, v in sorted(s2.items())}))
    out.append("k_rollout_state: scalar instructions %.0f %% of wave cycles (a lone wavefront per SIMD issues them in its own time), s_nop and other MISC %.1f %%, %.0f K vector memory instructions per dispatch"
               % (100 * s2["SQ_ACTIVE_INST_SCA"] / s["SQ_WAVE_CYCLES"], 100 * s2["SQ_ACTIVE_INST_MISC"] / s["SQ_WAVE_CYCLES"], s2["SQ_INSTS_VMEM"] / 1e3))
sys.path.insert(0, ROOT)
import bench
rec = {"tag": tag, "workload": "bench.py default (hopper T=100 batch=4096, od_rollout_compact)", "hbm_bytes_per_step": tot,
       "formula": "2*FETCH_SIZE + WRITE_SIZE [KB*1024] over k_rollout_state* and k_grad_knots, mean per dispatch",
       "source_hash": bench.kernel_source_hash()}
# executed (not algorithmic) fp64 work: wavefront-level instruction counts of the fp64 arithmetic classes; one wavefront
# instruction occupies 64 lane slots of the SIMD whatever its lanes hold (a cooperative row has 6 role lanes + their mirrors)
rec["valu_issue_util"] = s["SQ_ACTIVE_INST_VALU"] / s["SQ_WAVE_CYCLES"]
rec["valu_insts_per_step"] = s["SQ_INSTS_VALU"] + m["k_grad_knots"]["SQ_INSTS_VALU"]
if os.path.exists(os.path.join(src, "pmc_f64", "p_counter_collection.csv")):
    mf, _ = per_kernel("pmc_f64")
    tot_f = {c: sum(mf[k].get(c, 0.0) for k in mf) for c in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64", "SQ_INSTS_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU")}
    wave_flops = tot_f["SQ_INSTS_VALU_ADD_F64"] + tot_f["SQ_INSTS_VALU_MUL_F64"] + 2.0 * tot_f["SQ_INSTS_VALU_FMA_F64"]
    rec["fp64_wave_instructions_per_step"] = {k: v for k, v in tot_f.items() if k.endswith("F64")}
    rec["executed_lane_slot_flops_per_step"] = 64.0 * wave_flops          # what the SIMDs spend
    rec["active_lane_fraction"] = tot_f["SQ_THREAD_CYCLES_VALU"] / (64.0 * tot_f["SQ_ACTIVE_INST_VALU"]) if tot_f["SQ_ACTIVE_INST_VALU"] else None
    rec["fp64_fraction_of_valu_instructions"] = (tot_f["SQ_INSTS_VALU_ADD_F64"] + tot_f["SQ_INSTS_VALU_MUL_F64"] + tot_f["SQ_INSTS_VALU_FMA_F64"] + tot_f["SQ_INSTS_VALU_TRANS_F64"]) / tot_f["SQ_INSTS_VALU"]
    out.append("fp64 arithmetic, wavefront-level instructions per step (both kernels): %s; %.1f %% of the VALU instructions; x 64 lanes = %.3e executed lane-slot flops per step (%.0f per unit); active lanes %.0f %%"
               % (json.dumps({k: int(v) for k, v in rec["fp64_wave_instructions_per_step"].items()}), 100 * rec["fp64_fraction_of_valu_instructions"],
                  rec["executed_lane_slot_flops_per_step"], rec["executed_lane_slot_flops_per_step"] / 409600.0, 100 * (rec["active_lane_fraction"] or 0)))
json.dump(rec, open(os.path.join(dst, tag + "_traffic.json"), "w"), indent=1)
notes = os.path.join(dst, tag + "_pmc_notes.txt")
if os.path.exists(notes):
    out.append("")
    out.append(open(notes).read().rstrip())
open(os.path.join(dst, tag + "_pmc_summary.txt"), "w").write("\n".join(out) + "\n")
# the bench line of this profile run was printed before its own PMC passes existed: give it their traffic figure
# (same sources, same command), as bench.py does on every later run through profiles/<tag>_traffic.json
line = json.load(open(os.path.join(src, "bench.json")))
if line["roofline"].get("traffic") is None:
    line["roofline"]["traffic"] = tot
    line["roofline"]["traffic_note"] = "profiles/%s_traffic.json: the PMC passes of this same profile run (tools/profile_round.sh); algorithmic = %d" % (
        tag, int(line["roofline"]["traffic_note"].rsplit("algorithmic = ", 1)[1]))
json.dump(line, open(os.path.join(dst, tag + "_bench.json"), "w"))
print("\n".join(out))
