# usage (on the GPU box, via gpurun): bash tools/round_records.sh r2
# every measurement record of a round except the rocprof passes (tools/profile_round.sh): bench repeats, batch sweep,
# the other BASELINE configs, cooperative vs lane-per-problem check.  Output: gpurun_out/records_<tag>/
TAG=${1:-r2}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/records_$TAG
mkdir -p $O
cd $R
for i in 1 2 3; do python bench.py --steps 50 --warmup 3 --no-cpu-baseline; done > $O/bench_repeats.jsonl 2> $O/err.log
for B in 64 256 1024 2048 4096 8192 16384 65536; do python bench.py --batch $B --steps 10 --warmup 2 --no-cpu-baseline; done > $O/batch_sweep.jsonl 2>> $O/err.log
python tools/bench_configs.py > $O/other_configs.json 2>> $O/err.log
python tools/coop_check.py > $O/coop_check.log 2>> $O/err.log
ls -la $O
