# round 5: the new default chunk lengths of the overlapped schedule against round 3's (first = 0.24 T), interleaved; and, at T = 2000,
# the full-band stack's IO-wave kernel forced on beside the pair launch (SFSN_STACK_FB3=1) against round 2's bodies in the chunks (the rule at that time)
cd $GRAFT_REPO_ROOT
run() { timeout 120 python bench.py --no-cpu-baseline --no-phase-a --no-streaming-leg --sequential --steps 40 --warmup 6 $2 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$1: strict %.3f ms' % d['ms_per_step'])"; }
for i in 1 2 3; do
  run "B=64 new default" ""
  SFSN_OVERLAP_FRACS=0.24,0.38,0.38 run "B=64 round-3 chunks" ""
done
for b in 4 32; do for i in 1 2; do
  run "B=$b new default" "--batch $b"
  SFSN_OVERLAP_FRACS=0.24,0.38,0.38 run "B=$b round-3 chunks" "--batch $b"
done; done
for i in 1 2; do
  run "T=2000 new default" "--frames 2000"
  SFSN_STACK_FB3=1 run "T=2000 fb3 forced" "--frames 2000"
  SFSN_OVERLAP_FRACS=0.24,0.38,0.38 run "T=2000 round-3 chunks" "--frames 2000"
done
