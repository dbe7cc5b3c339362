cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_kernel_instantiations.py -x -q -m gpu > gpurun_out/t_inst.log 2>&1; echo "inst tests rc=$?" 
tail -15 gpurun_out/t_inst.log
timeout 600 python tools/diag_pp.py a > gpurun_out/diag_pp_a.log 2>&1; tail -60 gpurun_out/diag_pp_a.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_pp -o t -- python $R/tools/diag_pp.py prof > /dev/null 2>&1
find $R/gpurun_out/prof_pp -name "*kernel_stats.csv" | head -2 | xargs -I{} head -20 {}
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/t_gpu.log 2>&1; echo "gpu tests rc=$?"; tail -8 gpurun_out/t_gpu.log
python bench.py > gpurun_out/bench_a.json 2> gpurun_out/bench_a.err; cat gpurun_out/bench_a.json
