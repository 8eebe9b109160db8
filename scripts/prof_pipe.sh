cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_pipe
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o q -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --pipeline > $OUT/trace.log 2>&1
python - <<'PY'
import csv, collections
rows=list(csv.DictReader(open('gpurun_out/prof_pipe/trace/q_kernel_trace.csv')))
ours=[r for r in rows if any(k in r['Kernel_Name'] for k in ('gsn_scan','proj','features','deepfilter'))]
t0=min(int(r['Start_Timestamp']) for r in ours)
last=ours[-400:]
base=int(last[0]['Start_Timestamp'])
agg=collections.defaultdict(list)
for r in last:
    agg[r['Kernel_Name'][:40]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in agg.items(): print("%-42s n=%3d mean %8.1f us max %8.1f"%(k,len(v),sum(v)/len(v),max(v)))
print("span of last 400 kernels: %.2f ms; sum of durations %.2f ms"%((int(last[-1]['End_Timestamp'])-base)/1e6, sum(sum(v) for v in agg.values())/1e3))
for r in last[:60]:
    print("%8.1f %8.1f q=%s %s"%((int(r['Start_Timestamp'])-base)/1e3,(int(r['End_Timestamp'])-base)/1e3, r.get('Queue_Id','?'), r['Kernel_Name'][:50]))
PY
