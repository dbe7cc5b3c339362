cd $GRAFT_REPO_ROOT
( time python bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err ) 2>&1 | grep real
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_full.json').read().strip().splitlines()[-1])
r=d["roofline"]
print("ms", d["ms_per_step"], "frac", r["frac"], "traffic", r["traffic"], "exec_frac", r["executed_frac"], "issue", r["valu_issue_util"], "floor", r["latency_floor_ms"])
print("dense", d["dense_fx_fu"]["ms_per_step"], "aux_roll", d.get("aux_large_batch_rollouts"))
print("aux_knots", d["aux_independent_knots"]["algorithmic_frac"], "cpu", d["cpu_baseline"]["value"])
PY
