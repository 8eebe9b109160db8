# Round 5, review item 1: what is the timed region short of?  (run on the GPU box through gpurun; every step under its own timeout)
#  (a) the shader clock idle / scans alone / the region: scripts/diag_region_r05.py (clock probe + HIP-event scan durations)
#  (b) the CU-time ledger: kernel traces of the region and of the same forwards one at a time -> scripts/ledger_r05.py
#  (c) SQ wait counters "in the region" cannot be taken: rocprofv3 serialises dispatches under --pmc; cycles per step = duration x clock
#      from (a) answers the same question (do the scans execute more cycles in the region, or the same cycles at a lower clock?)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/diag_r05
rm -rf $OUT && mkdir -p $OUT
timeout 420 python scripts/diag_region_r05.py > $OUT/clock_digest.txt 2> $OUT/clock.err; echo "diag rc=$?"
cat $OUT/clock_digest.txt
B="python bench.py --no-cpu-baseline --no-phase-a --no-streaming-leg"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/region -o r -- $B --steps 36 --warmup 12 > $OUT/region.log 2>&1
SFSN_OVERLAP_CHUNKS=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/alone -o a -- $B --sequential --rpw 8,16 --steps 6 --warmup 2 > $OUT/alone.log 2>&1
python scripts/ledger_r05.py $OUT/region/r_kernel_trace.csv 36 $OUT/alone/a_kernel_trace.csv > $OUT/ledger.json 2> $OUT/ledger.err
head -c 1200 $OUT/ledger.json
# the same region without the fp32 spike tensors (the mode the live recipe runs): value, strict forward
timeout 300 $B --no-layer-outputs --steps 36 --warmup 12 > $OUT/bench_region_nolayers.json 2>> $OUT/bench.err
timeout 300 $B --no-layer-outputs --sequential --steps 12 --warmup 3 > $OUT/bench_strict_nolayers.json 2>> $OUT/bench.err
timeout 300 $B --sequential --steps 12 --warmup 3 > $OUT/bench_strict.json 2>> $OUT/bench.err
timeout 300 $B --steps 36 --warmup 12 > $OUT/bench_region.json 2>> $OUT/bench.err
for f in region_nolayers strict_nolayers strict region; do python -c "
import json,sys
d=json.loads(open('$OUT/bench_$f.json').read().strip().splitlines()[-1])
print('$f: value %.2f M frames/s  ms/step %.3f' % (d['value']/1e6, d['ms_per_step']))
"; done
# keep what travels back small: the per-dispatch traces stay on the box (the stats and the ledger are what is kept)
rm -f $OUT/alone/a_kernel_trace.csv
gzip -f $OUT/region/r_kernel_trace.csv
