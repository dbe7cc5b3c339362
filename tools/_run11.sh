cd $GRAFT_REPO_ROOT
python tools/headline_hash.py gpurun_out/hash_after3.json > /dev/null 2>&1
python - <<'PY'
import json
a=json.load(open('profiles/r3_hash_before_parallel_ls.json')); b=json.load(open('gpurun_out/hash_after3.json'))
print('all identical' if all(a[k]==b[k] for k in a) else [k for k in a if a[k]!=b[k]])
PY
python tools/acrobot_modes.py 2>&1 | grep -v amdgpu
python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
