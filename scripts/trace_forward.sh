# kernel timeline of one strict forward (B=64, T=1000): rocprofv3 --kernel-trace of bench.py --sequential, last forward printed
# usage (on the GPU box): bash scripts/trace_forward.sh <tag> [ENV=VALUE ...]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tag=$1; shift
OUT=gpurun_out/trace_$tag
rm -rf $OUT && mkdir -p $OUT
env "$@" rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python bench.py --no-cpu-baseline --sequential --steps 6 --warmup 3 --no-phase-a > $OUT/log.txt 2>&1
python - <<PY > $OUT/timeline.txt
import csv
rows=list(csv.DictReader(open('$OUT/t_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# the last forward starts at the last FillFunctor<float> pair before the last features_kernel run
idx=[i for i,r in enumerate(rows) if 'FillFunctor<float>' in r['Kernel_Name']]
start=idx[-2] if len(idx)>1 else 0
sel=rows[start:]
t0=int(sel[0]['Start_Timestamp'])
for r in sel:
    s=int(r['Start_Timestamp'])-t0; e=int(r['End_Timestamp'])-t0
    print(f"{s/1e3:9.1f} {e/1e3:9.1f} dur {(e-s)/1e3:8.1f} q{r['Queue_Id']:>3s} grid {r['Grid_Size_X']:>8s} {r['Kernel_Name'][:60]}")
PY
rm -f $OUT/t_kernel_trace.csv
tail -3 $OUT/log.txt | cut -c1-200
