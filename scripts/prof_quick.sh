# quick per-kernel durations of the single-stream forward (run on the GPU box)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_quick
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/single -o s -- python bench.py --inflight 1 --steps 6 --warmup 2 --no-cpu-baseline "$@" > $OUT/single.log 2>&1
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_quick/single/s_kernel_stats.csv')))
for r in rows[:22]:
    print("%-70s calls %4s avg_us %9.1f pct %s" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
