cd $GRAFT_REPO_ROOT
python tools/bench_ilqr.py 2>/dev/null > gpurun_out/ilqr.json; python -c "
import json; d=json.load(open('gpurun_out/ilqr.json'))
for k,v in d.items(): print(k, 'ms/it %.2f' % v['ms_per_iteration'], 'knot solves/s %.3g' % v['knot_solves_per_s'], 'J %.4g -> %.4g' % (v['J0_mean'], v['Jf_mean']))"
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_ilqr -o t -- python $GRAFT_REPO_ROOT/tools/bench_ilqr.py > /dev/null 2>&1
head -14 $GRAFT_REPO_ROOT/gpurun_out/prof_ilqr/t_kernel_stats.csv | cut -c1-150
