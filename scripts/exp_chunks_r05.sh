# Round 5: the strict forward against the chunking of the full-band / sub-band overlap (SFSN_OVERLAP_CHUNKS, SFSN_OVERLAP_FIRST,
# SFSN_OVERLAP_FRACS = explicit chunk lengths as fractions of T), B = 64, T = 1000
cd $GRAFT_REPO_ROOT
run() { timeout 120 python bench.py --no-cpu-baseline --no-phase-a --no-streaming-leg --sequential --steps 40 --warmup 6 $2 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$1: strict %.3f ms' % d['ms_per_step'])"; }
for f in ${FIRSTS:-240 300 320 340 360 400}; do SFSN_OVERLAP_CHUNKS=3 SFSN_OVERLAP_FIRST=$f run "chunks 3 first $f"; done
for fr in ${FRACS:-0.32,0.33,0.35 0.30,0.33,0.37 0.34,0.33,0.33 0.30,0.30,0.40 0.36,0.32,0.32 0.28,0.24,0.24,0.24}; do SFSN_OVERLAP_FRACS=$fr run "fracs $fr"; done
for f in 240 320; do SFSN_OVERLAP_CHUNKS=3 SFSN_OVERLAP_FIRST=$f run "no layer outputs, chunks 3 first $f" --no-layer-outputs; done
