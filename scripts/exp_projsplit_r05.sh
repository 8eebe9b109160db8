# round 5: the narrow PROJ role split by columns over its padding workgroups (SFSN_PROJ_SPLIT caps the parts: 1 = round 2's role)
cd $GRAFT_REPO_ROOT
run() { timeout 120 python bench.py --no-cpu-baseline --no-phase-a --no-streaming-leg --sequential --steps 40 --warmup 6 $2 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$1: strict %.3f ms' % d['ms_per_step'])"; }
timeout 600 python -m pytest tests/test_stack_scan.py -x -q -m gpu 2>&1 | tail -n 2
for i in 1 2; do
  for sp in 1 2 4; do
    SFSN_PROJ_SPLIT=$sp run "PROJ parts <= $sp" ""
    SFSN_PROJ_SPLIT=$sp SFSN_STACK_FB3=1 run "PROJ parts <= $sp, IO-wave full-band kernel in the chunks" ""
  done
done
timeout 300 python scripts/exp_beside_r05.py 2>&1 | grep -v amdgpu.ids
