# Round 5: per-step time of the scan kernels INSIDE the timed region, separated from the stagger of their workgroups' starts: kernel
# durations (rocprofv3 --kernel-trace) of the region at T = 500 / 1000 / 2000 frames -- duration(T) = a + b T: b = time per step in the
# region, a = prologue + the wait of the launch's last workgroup for a free compute unit.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/region_T_r05
rm -rf $OUT && mkdir -p $OUT
B="python bench.py --no-cpu-baseline --no-phase-a --no-streaming-leg"
for T in 500 1000 2000; do
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/T$T -o r -- $B --frames $T --steps 36 --warmup 12 > $OUT/T$T.log 2>&1
  python scripts/ledger_r05.py $OUT/T$T/r_kernel_trace.csv 36 > $OUT/ledger_T$T.json 2> $OUT/ledger_T$T.err
  rm -f $OUT/T$T/r_kernel_trace.csv
done
python - <<PY
import json
d={T: json.load(open('$OUT/ledger_T%d.json' % T)) for T in (500,1000,2000)}
print('ms per forward:', {T: d[T]['ms_per_forward'] for T in d})
names=[k['kernel'] for k in d[1000]['kernels']]
for n in names:
    v={T: next((k['mean_ms_in_region'] for k in d[T]['kernels'] if k['kernel']==n), None) for T in d}
    if None in v.values(): continue
    b=(v[2000]-v[500])/1500.0; a=v[1000]-b*1000
    print('%-44s T=500 %.4f  T=1000 %.4f  T=2000 %.4f ms   -> %.4f us per frame + %.4f ms' % (n[:44], v[500], v[1000], v[2000], 1e3*b, a))
PY
