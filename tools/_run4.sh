cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_coop3.py -x -q -m gpu > gpurun_out/t_c3.log 2>&1; echo "coop3 tests rc=$?"; tail -4 gpurun_out/t_c3.log
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/t_gpu.log 2>&1; echo "gpu tests rc=$?"; tail -6 gpurun_out/t_gpu.log
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_bundle -o t -- python $GRAFT_REPO_ROOT/tools/run_bundle_only.py 20 > /dev/null 2>&1
head -8 $GRAFT_REPO_ROOT/gpurun_out/prof_bundle/t_kernel_stats.csv | cut -c1-200
