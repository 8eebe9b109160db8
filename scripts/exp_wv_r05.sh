cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_stack_scan.py tests/test_hip_parity.py -x -q -m gpu 2>&1 | tail -n 2
run() { timeout 120 python bench.py --no-cpu-baseline --no-phase-a --no-streaming-leg --sequential --steps 40 --warmup 6 $2 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$1: strict %.3f ms' % d['ms_per_step'])"; }
for i in 1 2; do run "strict" ""; run "strict, no fp32 spike tensors" "--no-layer-outputs"; done
timeout 300 python scripts/exp_beside_r05.py 2>&1 | grep -v amdgpu.ids
