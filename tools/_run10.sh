cd $GRAFT_REPO_ROOT
python tools/headline_hash.py gpurun_out/hash_rd.json variants/libod_rowdecoupled.so > /dev/null 2>&1
python - <<'PY'
import json
a=json.load(open('profiles/r3_hash_before_parallel_ls.json')); b=json.load(open('gpurun_out/hash_rd.json'))
for k in a:
    if 'hopper_rollout' in k: print(k, 'IDENTICAL' if a[k]==b[k] else 'DIFFERENT %s %s' % (a[k], b[k]))
PY
for B in 4096 1024 2048; do
python tools/time_rollout.py - 2 $B 100 2>&1 | grep -v amdgpu
python tools/time_rollout.py variants/libod_rowdecoupled.so 2 $B 100 2>&1 | grep -v amdgpu
done | tee gpurun_out/row_decoupling.txt
