# usage (GPU box, via gpurun): bash tools/round6_gpu.sh   -- the validation sequence round 6 could not run (the GPU was closed to the builder):
# the -m gpu tier at seed offsets 0 and 82, smoke(), the bench line, the headline hash, both profile scripts.  Output: gpurun_out/r6a/
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6a
(time timeout 700 python -m pytest tests -q -m gpu --durations=25) > gpurun_out/r6a/tier_off0.txt 2>&1; tail -3 gpurun_out/r6a/tier_off0.txt
(OD_SEED_OFFSET=82 timeout 700 python -m pytest tests -q -m gpu) > gpurun_out/r6a/tier_off82.txt 2>&1; tail -3 gpurun_out/r6a/tier_off82.txt
(time python -c "import __graft_entry__ as g; g.smoke()") > gpurun_out/r6a/smoke.txt 2>&1; tail -2 gpurun_out/r6a/smoke.txt
(time python bench.py --gpus 1 --steps 20 --warmup 5) > gpurun_out/r6a/bench.txt 2>&1; tail -4 gpurun_out/r6a/bench.txt
python tools/headline_hash.py gpurun_out/r6a/hash.json > /dev/null 2> gpurun_out/r6a/hash.err
timeout 900 bash tools/profile_round.sh r6 > gpurun_out/r6a/profile_round.log 2>&1
timeout 1200 bash tools/profile_configs.sh r6 > gpurun_out/r6a/profile_configs.log 2>&1
