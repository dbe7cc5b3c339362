cd $GRAFT_REPO_ROOT
python tools/headline_hash.py gpurun_out/hash_after2.json > /dev/null 2>&1
python - <<'PY'
import json
a=json.load(open('profiles/r3_hash_before_parallel_ls.json')); b=json.load(open('gpurun_out/hash_after2.json'))
for k in a:
    print(k, 'IDENTICAL' if a[k]==b[k] else 'DIFFERENT %s %s' % (a[k], b[k]))
PY
for i in 1 2 3; do python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'])"; done
python bench.py --no-cpu-baseline --batch 8192 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('8192:', d['ms_per_step'])"
