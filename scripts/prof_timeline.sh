# kernel timeline of one forward alone (the strict schedule): start / end / duration of every kernel of the last forward, in order
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_timeline
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o s -- python bench.py --no-cpu-baseline --sequential --steps 4 --warmup 2 --no-phase-a > $OUT/log.txt 2>&1
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_timeline/t/s_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'features_kernel' in r['Kernel_Name']]
# a forward = 6 features launches (3 chunks x fb, sb): take the last complete forward
a=idx[-6]
t0=int(rows[a]['Start_Timestamp'])
last_end=0
for r in rows[a:]:
    s=(int(r['Start_Timestamp'])-t0)/1e3; e=(int(r['End_Timestamp'])-t0)/1e3
    print("%8.1f %8.1f dur %7.1f gap %6.1f  q%s  %s" % (s, e, e-s, s-last_end if last_end else 0, r.get('Queue_Id','?'), r['Kernel_Name'][:60]))
    last_end=max(last_end,e)
PY
