#!/bin/bash
# strict forward with the round-6 schedule switches of the overlapped forward (prep_next, post_stream), interleaved
run() { n=$1; shift
  env "$@" python bench.py --steps 24 --no-cpu-baseline --no-training-leg --no-streaming-leg --no-w16-leg 2>/dev/null | python -c "
import json,sys; l=json.loads(sys.stdin.readline()); c=l['config']; print('$n', 'strict', c['single_stream']['ms_per_step'], 'lean strict', c['no_layer_outputs']['single_stream']['ms_per_step'], 'value', l['value'])"
}
for i in 1 2 3; do
  run base X=1
  run prep_next SFSN_OV_PREP_NEXT=1
  run post_stream SFSN_OV_POST_STREAM=1
  run both SFSN_OV_PREP_NEXT=1 SFSN_OV_POST_STREAM=1
done
