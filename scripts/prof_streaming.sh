# kernel timeline of one streaming hop (run on the GPU box)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_stream
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o s -- python bench.py --streaming --steps 300 --warmup 50 --no-cpu-baseline > $OUT/log.txt 2>&1
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_stream/t/s_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# take a window late in the run: find the last hist_shift occurrences
idx=[i for i,r in enumerate(rows) if 'hist_shift' in r['Kernel_Name'] or 'stream_hop' in r['Kernel_Name']]
a,b=idx[-3],idx[-2]
t0=int(rows[a]['Start_Timestamp'])
for r in rows[a:b+1]:
    s=int(r['Start_Timestamp'])-t0; e=int(r['End_Timestamp'])-t0
    print("%7.1f %7.1f dur %6.1f  %s" % (s/1e3, e/1e3, (e-s)/1e3, r['Kernel_Name'][:70]))
PY
