cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pt && mkdir -p gpurun_out/pt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pt -o t -- python bench.py --training --batch 16 --steps 2 --warmup 1 > gpurun_out/pt/log.txt 2>&1
python - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/pt/t_kernel_stats.csv")))
for r in rows[:6]:
    print(r["Name"][:60], r["Calls"], round(float(r["AverageNs"])/1e3,2), r["Percentage"])
PY
rm -f gpurun_out/pt/t_kernel_trace.csv
