# kernel durations of one-launch streaming hops (run on the GPU box)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_hop
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o s -- python bench.py --streaming --steps 300 --warmup 50 --no-cpu-baseline > $OUT/log.txt 2>&1
grep '"metric"' $OUT/log.txt | cut -c1-400
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_hop/t/s_kernel_stats.csv')))
for r in rows[:8]: print(r['Name'][:60], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
PY
