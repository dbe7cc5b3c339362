# usage (GPU box): bash tools/pmc_cmd.sh TAG KERNEL_SUBSTRING -- python script.py args...
# separate rocprofv3 --pmc passes (kernel-trace only), per-kernel means printed and written to gpurun_out/pmc_TAG.json
TAG=$1; KF=$2; shift 3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_$TAG
mkdir -p $O
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_WAVES SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM" \
           "SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "SQ_WAVES SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/g$i -o p -- "$@" > /dev/null 2>&1
done
python - <<PY
import csv, collections, json, glob
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$O/g*/p_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if '$KF' in r['Kernel_Name']:
            acc[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
out = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}
print(json.dumps(out, indent=1))
json.dump(out, open('$R/gpurun_out/pmc_$TAG.json', 'w'), indent=1)
PY
