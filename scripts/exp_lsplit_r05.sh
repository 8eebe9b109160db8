# Round 5: the loader / storer split of a frame's fp32 spike stores in the 8-row IO-wave roles (SFSN_S3_LSPLIT: scan3 roles,
# SFSN_S3X_LSPLIT: the FUSEDX3 role; csrc/sfsn_scan3_dev.h).  Per pair of values: the strict forward (ms per forward) and the pair
# launch alone as one whole-sequence launch (HIP events), B = 64, T = 1000.
cd $GRAFT_REPO_ROOT
for v in ${LSPLITS:-0,0 2,2 2,0 2,1 3,0 3,1 7,0 7,1 7,2 1,1}; do
  a=${v%,*}; b=${v#*,}
  SFSN_S3_LSPLIT=$a SFSN_S3X_LSPLIT=$b timeout 240 python bench.py --no-cpu-baseline --no-streaming-leg --no-training-leg --sequential --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
s=d['config']['single_stream']; r=(d['roofline'] or {}).get('sub_band_scan_single_forward') or {}
print('SFSN_S3_LSPLIT=$a SFSN_S3X_LSPLIT=$b: strict forward %.3f ms (phase S) / %.3f ms (timed region, sequential); pair launch %.4f ms = %.3f us per frame, frac %.4f' % (s['ms_per_step'], d['ms_per_step'], r.get('launch_ms', 0), r.get('per_step_us', 0), r.get('frac', 0)))
"
done
