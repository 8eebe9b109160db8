# phase-B sweep: forwards in flight x rows per scan workgroup (run on the GPU box)
for inf in 4 6 8 10; do for rpw in 4,16 4,8 8,16 16,16; do
  python bench.py --no-cpu-baseline --inflight $inf --rpw $rpw --steps 60 --warmup 12 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('inflight $inf rpw $rpw value %.2fM ms %.3f' % (d['value']/1e6, d['ms_per_step']))
"
done; done
