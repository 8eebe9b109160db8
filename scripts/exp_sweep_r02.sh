# round 2: timed-region sweep with the stack launch in play: stack policy x forwards in flight x overlap chunks (run on the GPU box)
run() { python bench.py --no-cpu-baseline --no-phase-a --steps 60 --warmup 12 "$@" 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$*'.ljust(44), 'value %.2fM ms %.3f' % (d['value']/1e6, d['ms_per_step']))
"; }
for st in auto 0; do for inf in 8 12 16; do run --stack $st --inflight $inf; done; done
run --stack auto --inflight 12 --rpw 8,16
run --stack auto --inflight 24
SFSN_OVERLAP_CHUNKS=0 run --stack auto --inflight 12
for r in 8 16; do SFSN_FB_STACK_ROWS=$r run --stack auto --inflight 12; SFSN_FB_STACK_ROWS=$r run --stack auto --inflight 16; done
