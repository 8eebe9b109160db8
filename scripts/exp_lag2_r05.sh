# round 5: with the shorter hand-off hysteresis (lag 4): the IO-wave full-band kernel beside the pair launch again, the chunk lengths again,
# and the timed region (twelve forwards in flight)
cd $GRAFT_REPO_ROOT
run() { timeout 120 python bench.py --no-cpu-baseline --no-phase-a --no-streaming-leg --sequential --steps 40 --warmup 6 $2 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$1: strict %.3f ms' % d['ms_per_step'])"; }
reg() { timeout 200 python bench.py --no-cpu-baseline --no-phase-a --no-streaming-leg --steps 60 --warmup 12 $2 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$1: region %.3f ms = %.2f M' % (d['ms_per_step'], d['value']/1e6))"; }
export SFSN_STACK_LAG=4
for i in 1 2; do
  run "lag 4" ""
  SFSN_STACK_FB3=1 run "lag 4, IO-wave full-band kernel beside the pair" ""
  SFSN_OVERLAP_FRACS=0.32,0.33,0.35 run "lag 4, chunks .32 .33 .35" ""
  SFSN_OVERLAP_FRACS=0.28,0.36,0.36 run "lag 4, chunks .28 .36 .36" ""
  SFSN_OVERLAP_FRACS=0.20,0.40,0.40 run "lag 4, chunks .20 .40 .40" ""
  SFSN_OVERLAP_FRACS=0.2,0.27,0.27,0.26 run "lag 4, four chunks .2 .27 .27 .26" ""
done
for i in 1 2; do
  SFSN_STACK_LAG=16 reg "lag 16" ""
  SFSN_STACK_LAG=4 reg "lag 4" ""
done
