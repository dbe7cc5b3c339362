cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pm; mkdir -p $O
rocprofv3 -L > $O/counters.txt 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_SMEM --kernel-trace --output-format csv -d $O/a -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_WAIT_INST_ANY SQ_INSTS_VMEM SQ_IFETCH --kernel-trace --output-format csv -d $O/b -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
find $O -name "*counter_collection.csv"
