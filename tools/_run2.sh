cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -x -q -m gpu -k "planar or bundle or iteration_statistics or golden or sweep" > gpurun_out/t_pp.log 2>&1; echo "pp tests rc=$?"; tail -5 gpurun_out/t_pp.log
timeout 600 python tools/diag_pp.py b > gpurun_out/diag_pp_b.log 2>&1; head -12 gpurun_out/diag_pp_b.log; grep -A3 "step_grad_B65536\|step_B65536\|step_B1024\|step_B4096" gpurun_out/diag_pp_b.log
bash tools/pmc_cmd.sh bundle_b k_bundle -- python $GRAFT_REPO_ROOT/tools/run_bundle_only.py 4 | grep -E "INSTS_VALU|WAVE_CYCLES|WAIT_ANY|INSTS_LDS|ACTIVE_INST_VALU"
