# round 5, after lag 4 / PROJ split / IO-wave kernel in the chunks: chunk-length patterns of the strict forward once more (B = 64)
cd $GRAFT_REPO_ROOT
run() { timeout 120 python bench.py --no-cpu-baseline --no-phase-a --no-streaming-leg --sequential --steps 40 --warmup 6 $2 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$1: strict %.3f ms' % d['ms_per_step'])"; }
for i in 1 2 3; do
  run "default (.24 .38 .38)" ""
  for f in 0.30,0.40,0.30 0.34,0.36,0.30 0.32,0.33,0.35 0.36,0.36,0.28 0.28,0.40,0.32 0.26,0.32,0.24,0.18; do
    SFSN_OVERLAP_FRACS=$f run "chunks $f" ""
  done
done
