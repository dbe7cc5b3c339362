cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for ppw in 16 8 4; do
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/kt_$ppw -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --ppw $ppw > /dev/null 2>&1
python - <<PY
import csv,collections
kt=list(csv.DictReader(open("$R/gpurun_out/kt_$ppw/p_kernel_trace.csv")))
acc=collections.defaultdict(list)
for r in kt: acc[r['Kernel_Name'][:60]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6)
print("ppw $ppw", {k:[round(x,2) for x in v] for k,v in acc.items() if 'od::' in k})
PY
done
