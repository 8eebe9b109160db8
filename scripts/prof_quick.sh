cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_quick
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o q -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/trace.log 2>&1
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_quick/trace/q_kernel_stats.csv')))
for r in rows[:16]:
    print("%-60s calls %4s avg_us %9.1f pct %s"%(r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
