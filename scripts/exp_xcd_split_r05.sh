# round 5: the strict forward with the full-band stream confined to the first XCD(s) by a CU mask and the sub-band stream to the others
# (SFSN_OV_XCD_SPLIT = number of XCDs for the full-band stream; SFSN_FB_STACK_ROWS = rows per workgroup of the full-band stack)
cd $GRAFT_REPO_ROOT
run() { timeout 120 python bench.py --no-cpu-baseline --no-phase-a --no-streaming-leg --sequential --steps 30 --warmup 6 $2 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$1: strict %.3f ms' % d['ms_per_step'])"; }
for i in 1 2; do
  run "no masks, 4 rows" ""
  SFSN_OV_XCD_SPLIT=1 SFSN_FB_STACK_ROWS=8 run "fb on 1 XCD at 8 rows" ""
  SFSN_FB_STACK_ROWS=8 run "no masks, 8 rows" ""
  SFSN_OV_XCD_SPLIT=2 SFSN_FB_STACK_ROWS=4 run "fb on 2 XCDs at 4 rows (sub-band on 192 CUs)" ""
  SFSN_OV_XCD_SPLIT=1 SFSN_FB_STACK_ROWS=8 run "fb on 1 XCD at 8 rows, no fp32 spike tensors" "--no-layer-outputs"
  run "no masks, no fp32 spike tensors" "--no-layer-outputs"
done
