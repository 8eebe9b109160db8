cd $GRAFT_REPO_ROOT
run() { env $1 timeout 200 python bench.py --no-cpu-baseline --no-streaming-leg --no-training-leg $2 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); s=d['config'].get('single_stream') or {}; nl=(d['config'].get('no_layer_outputs') or {})
print('$1 $2: value %.2f M, ms %.3f, strict %s, nl strict %s, nl region %s' % (d['value']/1e6, d['ms_per_step'], s.get('ms_per_step'), (nl.get('single_stream') or {}).get('ms_per_step'), ((nl.get('timed_region') or {}).get('value'))))"; }
for i in 1 2; do
  run "SFSN_S3_LSPLIT=2 SFSN_S3X_LSPLIT=1" ""
  run "SFSN_S3_LSPLIT=0 SFSN_S3X_LSPLIT=0" ""
done
for b in 16 32; do
  run "SFSN_S3_LSPLIT=2 SFSN_S3X_LSPLIT=1" "--sequential --batch $b --steps 20 --warmup 4 --no-phase-a"
  run "SFSN_S3_LSPLIT=0 SFSN_S3X_LSPLIT=0" "--sequential --batch $b --steps 20 --warmup 4 --no-phase-a"
done
