# round 5, end: the strict forward's tuned knobs once more on the final kernels (hand-off lag, PROJ column split, full-band rows)
cd $GRAFT_REPO_ROOT
run() { env $1 timeout 120 python bench.py --no-cpu-baseline --no-phase-a --no-streaming-leg --no-training-leg --sequential --steps 30 --warmup 6 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$1: strict %.3f ms' % d['ms_per_step'])"; }
for i in 1 2; do
  run "SFSN_STACK_LAG=4"
  run "SFSN_STACK_LAG=2"
  run "SFSN_STACK_LAG=1"
  run "SFSN_STACK_LAG=8"
  run "SFSN_PROJ_SPLIT=2"
  run "SFSN_PROJ_SPLIT=1"
  run "SFSN_OVERLAP_FRACS=0.28,0.36,0.36"
  run "SFSN_OVERLAP_FRACS=0.22,0.39,0.39"
done
