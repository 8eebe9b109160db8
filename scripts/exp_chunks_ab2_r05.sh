cd $GRAFT_REPO_ROOT
run() { timeout 120 python bench.py --no-cpu-baseline --no-phase-a --no-streaming-leg --sequential --steps 40 --warmup 6 $2 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$1: strict %.3f ms' % d['ms_per_step'])"; }
for b in 4 16 32; do for i in 1 2; do
  SFSN_OVERLAP_FIRST=-1 run "B=$b default" "--batch $b"
  SFSN_OVERLAP_FRACS=0.32,0.33,0.35 run "B=$b fracs .32,.33,.35" "--batch $b"
  SFSN_OVERLAP_FRACS=0.36,0.32,0.32 run "B=$b fracs .36,.32,.32" "--batch $b"
done; done
for i in 1 2; do
  SFSN_OVERLAP_FIRST=-1 run "T=500 default" "--frames 500"
  SFSN_OVERLAP_FRACS=0.32,0.33,0.35 run "T=500 fracs .32,.33,.35" "--frames 500"
  SFSN_OVERLAP_FIRST=-1 run "T=2000 default" "--frames 2000"
  SFSN_OVERLAP_FRACS=0.32,0.33,0.35 run "T=2000 fracs .32,.33,.35" "--frames 2000"
done
