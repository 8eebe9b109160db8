# round 5: the hand-off hysteresis of the stack launches (Engine.stack_lag: a consumer that has to wait lets its producer run this many
# frames ahead) against the strict forward: every chunk's launch ends lag + ring frames after its first layer does
cd $GRAFT_REPO_ROOT
run() { timeout 120 python bench.py --no-cpu-baseline --no-phase-a --no-streaming-leg --sequential --steps 40 --warmup 6 $2 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$1: strict %.3f ms' % d['ms_per_step'])"; }
for i in 1 2; do
  for lag in ${LAGS:-16 8 4 2 1}; do SFSN_STACK_LAG=$lag run "lag $lag" ""; done
done
