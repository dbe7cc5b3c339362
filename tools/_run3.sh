cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_coop3.py -x -q -m gpu > gpurun_out/t_c3.log 2>&1; echo "coop3 tests rc=$?"; tail -12 gpurun_out/t_c3.log
timeout 900 python -m pytest tests -x -q -m gpu -k "planar or bundle or iteration_statistics or golden" > gpurun_out/t_pp.log 2>&1; echo "pp tests rc=$?"; tail -5 gpurun_out/t_pp.log
timeout 600 python tools/diag_pp.py c > gpurun_out/diag_pp_c.log 2>&1; head -12 gpurun_out/diag_pp_c.log; grep -A3 "step_grad_B\|step_B" gpurun_out/diag_pp_c.log | grep -v "^--" | paste - - - - | awk '{print $1, $3}'
bash tools/pmc_cmd.sh bundle_c k_bundle -- python $GRAFT_REPO_ROOT/tools/run_bundle_only.py 4 | grep -E "k_bundle|INSTS_VALU|WAVE_CYCLES|WAIT_ANY|WAIT_INST_ANY|SQ_WAVES|ACTIVE_INST_VALU|BUSY_CYCLES"
