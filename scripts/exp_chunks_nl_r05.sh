# round 5: chunk lengths of the strict forward in the mode without fp32 spike tensors (the pair launch is faster there: does the balance move?)
cd $GRAFT_REPO_ROOT
run() { timeout 120 python bench.py --no-cpu-baseline --no-phase-a --no-streaming-leg --no-training-leg --sequential --no-layer-outputs --steps 30 --warmup 6 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$1: strict %.3f ms' % d['ms_per_step'])"; }
for i in 1 2; do
  run "default (0.24 T first)"
  SFSN_OVERLAP_FRACS=0.20,0.40,0.40 run "fracs .20 .40 .40"
  SFSN_OVERLAP_FRACS=0.28,0.36,0.36 run "fracs .28 .36 .36"
  SFSN_OVERLAP_FRACS=0.24,0.40,0.36 run "fracs .24 .40 .36"
  SFSN_OVERLAP_FRACS=0.24,0.42,0.34 run "fracs .24 .42 .34"
  SFSN_OVERLAP_FRACS=0.20,0.28,0.28,0.24 run "fracs .20 .28 .28 .24"
done
