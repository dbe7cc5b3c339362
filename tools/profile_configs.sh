# usage (on the GPU box, via gpurun): bash tools/profile_configs.sh r4
# BASELINE configs 2, 3 and 5: kernel trace + stats, then separate PMC passes (HBM traffic, SQ occupancy / issue counters, fp64 / fp32
# instruction classes) for the kernels bench_configs.py names; tools/summarize_configs.py condenses the result into profiles/.
TAG=${1:-r4}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_cfg_$TAG
mkdir -p $O
run() {   # name, command...
  n=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${n}_trace -o t -- "$@" > $O/${n}.out 2>/dev/null
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${n}_fetch -o p -- "$@" > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${n}_write -o p -- "$@" > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/${n}_sq -o p -- "$@" > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 --kernel-trace --output-format csv -d $O/${n}_fl -o p -- "$@" > /dev/null 2>&1
}
run config2 python $R/tools/run_config_only.py acrobot 4
run config3 python $R/tools/run_config_only.py bundle 4
run config5_f32 python $R/tools/prof_ilqr_device.py float32 4096 config5
run config5_f64 python $R/tools/prof_ilqr_device.py float64 4096 config5
run config5_hover_f32 python $R/tools/prof_ilqr_device.py float32 4096 hover
python $R/bench_configs.py > $O/bench_configs.json 2> $O/bench_configs.err
ls $O
