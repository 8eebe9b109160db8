# round 5: the full-band stream of the strict forward at high priority
cd $GRAFT_REPO_ROOT
run() { timeout 120 python bench.py --no-cpu-baseline --no-phase-a --no-streaming-leg --sequential --steps 40 --warmup 6 $2 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$1: strict %.3f ms' % d['ms_per_step'])"; }
for i in 1 2 3; do
  SFSN_FB_PRIORITY=0 run "equal priorities" ""
  SFSN_FB_PRIORITY=1 run "full-band stream high" ""
  SFSN_FB_PRIORITY=0 run "equal priorities, B=16" "--batch 16"
  SFSN_FB_PRIORITY=1 run "full-band stream high, B=16" "--batch 16"
done
EXTRA="" bash scripts/trace_strict_r05.sh | tail -n 34
