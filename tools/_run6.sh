cd $GRAFT_REPO_ROOT
python tools/headline_hash.py gpurun_out/hash_after.json > /dev/null 2>&1
python - <<'PY'
import json
a=json.load(open('profiles/r3_hash_before_parallel_ls.json')); b=json.load(open('gpurun_out/hash_after.json'))
for k in a:
    same = a[k]==b[k]
    print(k, 'IDENTICAL' if same else 'DIFFERENT', '' if same else (a[k], b[k]))
PY
python bench.py --no-cpu-baseline 2>/dev/null | cut -c1-330
python bench.py --no-cpu-baseline --batch 8192 2>/dev/null | cut -c100-330
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/t_gpu.log 2>&1; echo "gpu tests rc=$?"; tail -4 gpurun_out/t_gpu.log
python tools/bench_configs.py 2>/dev/null | grep -A2 "acrobot_impact B=1024"
