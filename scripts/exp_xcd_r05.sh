# round 5: the full-band launch in XCD-chained block order (plain-store hand-offs through the XCD's L2) against the contiguous order
cd $GRAFT_REPO_ROOT
run() { timeout 120 python bench.py --no-cpu-baseline --no-phase-a --no-streaming-leg --sequential --steps 40 --warmup 6 $2 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$1: strict %.3f ms' % d['ms_per_step'])"; }
timeout 900 python -m pytest tests/test_stack_scan.py tests/test_hip_parity.py -x -q -m gpu 2>&1 | tail -n 3
for i in 1 2; do
  SFSN_STACK_XCD=0 run "contiguous order, round-2 bodies in the chunks" ""
  SFSN_STACK_XCD=0 SFSN_STACK_FB3=1 run "contiguous order, IO-wave kernel in the chunks" ""
  SFSN_STACK_XCD=1 SFSN_STACK_FB3=1 run "XCD-chained order, IO-wave kernel in the chunks" ""
done
for x in 0 1; do echo "=== SFSN_STACK_XCD=$x"; SFSN_STACK_XCD=$x timeout 300 python scripts/exp_beside_r05.py 2>&1 | grep -v amdgpu.ids; done
