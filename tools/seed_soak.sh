# usage (GPU box): bash tools/seed_soak.sh -- the whole GPU tier under 12 seed offsets (OD_SEED_OFFSET), one summary line each
cd $GRAFT_REPO_ROOT
for k in 1 2 3 4 5 6 7 8 9 10 11 12; do
  OD_SEED_OFFSET=$k timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -1 | sed "s/^/seed offset $k: /"
done | tee gpurun_out/seed_soak_gpu.txt
