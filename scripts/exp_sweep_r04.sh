# round 4: the timed region (forwards in flight x rows per scan workgroup) with the pair launch in the pool: sub-band geometry 8
# = all sub-band layers of a lane's forward in ONE 208-workgroup launch (FUSEDX3 / FUSED3 roles), 16 = round 2's fused 16-row
# kernels per layer.  Per line: frames/s, ms per step, and the scan kernels' rows x steps per CU-us alone on the chip.
for rpw in 8,16 8,8 4,8 16,16; do for inf in 2 3 4 6 8 12; do
  python bench.py --no-cpu-baseline --no-phase-a --no-streaming-leg --inflight $inf --rpw $rpw --steps 36 --warmup 12 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('rpw $rpw inflight %2d: value %.2f M frames/s  ms/step %.3f' % ($inf, d['value']/1e6, d['ms_per_step']))
"
done; done
