cd $GRAFT_REPO_ROOT
run() { timeout 120 python bench.py --no-cpu-baseline --no-phase-a --no-streaming-leg --sequential --steps 60 --warmup 6 $2 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$1: strict %.3f ms' % d['ms_per_step'])"; }
for i in 1 2 3 4; do
  SFSN_OVERLAP_FIRST=240 run "first 240"
  SFSN_OVERLAP_FIRST=280 run "first 280"
  SFSN_OVERLAP_FIRST=320 run "first 320"
  SFSN_OVERLAP_FRACS=0.32,0.33,0.35 run "fracs .32,.33,.35"
  SFSN_OVERLAP_FRACS=0.30,0.34,0.36 run "fracs .30,.34,.36"
done
