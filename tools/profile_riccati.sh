# usage (GPU box, via gpurun): bash tools/profile_riccati.sh r4
# k_ilqr_backward_mfma inside the iLQR iteration of config 5 (4096 problems, hover problem, fp32 and fp64): kernel trace, then separate PMC
# passes for the matrix cores (instructions, busy cycles), LDS (bank conflicts) and HBM bytes.  Output: gpurun_out/prof_riccati_<tag>/
TAG=${1:-r4}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_riccati_$TAG
mkdir -p $O
for d in float32 float64; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${d}_trace -o t -- python $R/tools/prof_ilqr_device.py $d 4096 hover > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/${d}_mfma -o p -- python $R/tools/prof_ilqr_device.py $d 4096 hover > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/${d}_lds -o p -- python $R/tools/prof_ilqr_device.py $d 4096 hover > /dev/null 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${d}_fetch -o p -- python $R/tools/prof_ilqr_device.py $d 4096 hover > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${d}_write -o p -- python $R/tools/prof_ilqr_device.py $d 4096 hover > /dev/null 2>&1
done
python - <<PY
import csv, glob, collections, json
O = "$O"
out = {}
for d in ("float32", "float64"):
    rec = {}
    tr = glob.glob(O + "/%s_trace/**/t_kernel_trace.csv" % d, recursive=True)[0]
    v = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(tr)) if "k_ilqr_backward_mfma" in r["Kernel_Name"]]
    rec["launches"] = len(v); rec["avg_us"] = sum(v) / len(v)
    for grp in ("mfma", "lds", "fetch", "write"):
        f = glob.glob(O + "/%s_%s/**/p_counter_collection.csv" % (d, grp), recursive=True)[0]
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "k_ilqr_backward_mfma" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, x in acc.items():
            rec[k] = sum(x) / len(x)
    out[d] = rec
json.dump(out, open(O + "/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
