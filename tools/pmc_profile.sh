cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for ppw in ${PPWS:-16 8}; do
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $R/gpurun_out/pmc1_$ppw -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --ppw $ppw > /dev/null 2>&1
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SMEM --kernel-trace --output-format csv -d $R/gpurun_out/pmc2_$ppw -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --ppw $ppw > /dev/null 2>&1
python - <<PY
import csv, collections
for d in ['pmc1_$ppw','pmc2_$ppw']:
    rows=list(csv.DictReader(open(f'$R/gpurun_out/{d}/p_counter_collection.csv')))
    acc=collections.defaultdict(list)
    for r in rows:
        if 'rollout' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    print(d, {k: round(sum(v)/len(v)/1e6,2) for k,v in acc.items()})
PY
done
