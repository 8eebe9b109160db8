# round 5: the full-band model announces chunk c behind its own features / input product of chunk c+1 (Engine.gate_behind_prep)
cd $GRAFT_REPO_ROOT
run() { timeout 120 python bench.py --no-cpu-baseline --no-phase-a --no-streaming-leg --sequential --steps 40 --warmup 6 $2 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$1: strict %.3f ms' % d['ms_per_step'])"; }
for i in 1 2 3; do
  for a in "" "--no-layer-outputs" "--batch 16"; do
  SFSN_GATE_BEHIND_PREP=0 run "[$a] announced behind the projection" "$a"
  SFSN_GATE_BEHIND_PREP=1 run "[$a] announced behind the prep of the next chunk" "$a"
  done
done

