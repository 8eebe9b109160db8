# kernel trace of one training step (B = 64): how much of the step's wall time is GPU-busy, and where the gaps are
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/trace_training
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python bench.py --training --batch 64 --steps 2 --warmup 1 > $OUT/log.txt 2>&1
tail -n 1 $OUT/log.txt | cut -c1-300
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/trace_training/**/t_kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last step: between the last two seq_fwd FIRST launches ... simply take the last 45 % of the trace by time
t_end = max(int(r['End_Timestamp']) for r in rows)
fw = [i for i, r in enumerate(rows) if 'gsn_train_seq_fwd' in r['Kernel_Name']]
# a step has 4 fwd launches: the last step starts a little before the 4th-from-last fwd launch
i0 = fw[-4]
# walk back to the previous bwd launch's end
bw = [i for i, r in enumerate(rows[:i0]) if 'gsn_train_seq_bwd' in r['Kernel_Name']]
start = bw[-1] + 1 if bw else 0
sel = rows[start:]
t0 = int(sel[0]['Start_Timestamp'])
busy, last_end, gaps = 0, t0, []
ivs = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in sel)
cur_s, cur_e = ivs[0]
for s, e in ivs[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append((s - cur_e, cur_e - t0)); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = cur_e - t0
print('last step (after the previous backward): span %.1f ms, GPU busy %.1f ms, idle %.1f ms in %d gaps; %d kernels' % (span / 1e6, busy / 1e6, (span - busy) / 1e6, len(gaps), len(sel)))
gaps.sort(reverse=True)
print('largest gaps (us @ ms):', [(round(g / 1e3), round(at / 1e6, 1)) for g, at in gaps[:12]])
agg = {}
for r in sel:
    k = r['Kernel_Name'][:60]
    agg.setdefault(k, [0, 0]); agg[k][0] += 1; agg[k][1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print('%8.2f ms %5d x  %s' % (t / 1e6, n, k))
# timeline of the big launches
for r in sel:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
    if d > 0.4:
        print('  @%7.2f ms  %6.2f ms  %s' % ((int(r['Start_Timestamp']) - t0) / 1e6, d, r['Kernel_Name'][:50]))
PY
rm -rf $OUT
