cd $GRAFT_REPO_ROOT
for inf in 8 10 12 16 20; do for st in "20 5" "60 12"; do set -- $st
  GPU_MAX_HW_QUEUES=32 timeout 120 python bench.py --no-cpu-baseline --no-phase-a --no-streaming-leg --inflight $inf --steps $1 --warmup $2 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('inflight $inf steps $1 warmup $2: value %.2f M frames/s  ms/step %.3f' % (d['value']/1e6, d['ms_per_step']))"
done; done
