# round 5: the full-band model's time-parallel kernels of all chunks ahead of its stack launches (Engine.overlap_prep_ahead), with the
# IO-wave full-band kernel (ring of four frames, a loader wave) instead of round 2's two-slot bodies
cd $GRAFT_REPO_ROOT
run() { timeout 120 python bench.py --no-cpu-baseline --no-phase-a --no-streaming-leg --sequential --steps 40 --warmup 6 $2 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$1: strict %.3f ms' % d['ms_per_step'])"; }
for i in 1 2 3; do
  SFSN_PREP_AHEAD=0 run "prep in the chain" ""
  SFSN_PREP_AHEAD=1 run "prep ahead" ""
  SFSN_PREP_AHEAD=0 run "prep in the chain, no fp32 spike tensors" "--no-layer-outputs"
  SFSN_PREP_AHEAD=1 run "prep ahead, no fp32 spike tensors" "--no-layer-outputs"
done
