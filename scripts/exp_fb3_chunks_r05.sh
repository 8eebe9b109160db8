# round 5, after lag 4 + the PROJ column split: round 2's bodies (SFSN_STACK_FB3=0: the rule until then) against the IO-wave
# kernel (SFSN_STACK_FB3=1) in the chunks of the strict forward, over batch sizes / lengths / output modes
cd $GRAFT_REPO_ROOT
run() { timeout 120 python bench.py --no-cpu-baseline --no-phase-a --no-streaming-leg --sequential --steps 40 --warmup 6 $2 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$1: strict %.3f ms' % d['ms_per_step'])"; }
for i in 1 2; do
for a in "" "--no-layer-outputs" "--batch 4" "--batch 16" "--batch 32" "--frames 500" "--frames 2000"; do
  SFSN_STACK_FB3=0 run "[$a] round-2 bodies" "$a"
  SFSN_STACK_FB3=1 run "[$a] IO-wave kernel" "$a"
done; done
