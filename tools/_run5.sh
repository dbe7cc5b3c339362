cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/t_gpu.log 2>&1; echo "gpu tests rc=$?"; tail -6 gpurun_out/t_gpu.log
python tools/sweep_hopper.py 2>&1 | grep -v amdgpu.ids
python bench.py --no-cpu-baseline 2>/dev/null | cut -c1-400
python bench.py --no-cpu-baseline --batch 8192 2>/dev/null | cut -c1-400
