# kernel trace of the strict forward (one forward at a time): the timeline of the last forward, per queue
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/trace_strict
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -o s -- python bench.py --no-cpu-baseline --no-streaming-leg --sequential --steps 6 --warmup 3 --no-phase-a $EXTRA > $OUT/log.txt 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/trace_strict/**/s_kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last forward: from the last FB features kernel that follows an error-word copy
idx = [i for i, r in enumerate(rows) if 'copyBuffer' in r['Kernel_Name'] or 'reduce_kernel' in r['Kernel_Name']]
sel = rows[-46:]
t0 = int(sel[0]['Start_Timestamp'])
for r in sel:
    print('%8.1f %8.1f q=%s %s grid=%s' % ((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r['Queue_Id'], r['Kernel_Name'][:44], r['Grid_Size_X']))
PY
rm -rf $OUT
