cd $GRAFT_REPO_ROOT
python tools/bench_configs.py 2>/dev/null > gpurun_out/other_configs.json; tail -5 gpurun_out/other_configs.json
for c in bundle acrobot pp_step; do bash tools/pmc_cmd.sh cfg_$c k_ -- python $GRAFT_REPO_ROOT/tools/run_config_only.py $c 4 > /dev/null 2>&1; done
ls gpurun_out/pmc_cfg_*.json
