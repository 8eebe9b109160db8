cd $GRAFT_REPO_ROOT
run() { timeout 120 python bench.py --no-cpu-baseline --no-phase-a --no-streaming-leg --sequential --steps 40 --warmup 6 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$1: strict %.3f ms' % d['ms_per_step'])"; }
for i in 1 2; do
unset SFSN_LIB_PATH SFSN_STACK_FB_V2; run default
SFSN_STACK_FB_V2=1 run V2
SFSN_FB_STACK_ROWS=8 run default_rows8
for v in ls4 ls4dg4 dg4pf3; do export SFSN_LIB_PATH=$GRAFT_REPO_ROOT/spiking_fullsubnet_amd/csrc_$v/libsfsn_hip.so; run $v; SFSN_FB_STACK_ROWS=8 run ${v}_rows8; done
done
