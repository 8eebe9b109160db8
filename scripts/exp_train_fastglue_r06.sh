#!/bin/bash
# training step with / without the round-6 glue (SFSN_TRAIN_FAST_GLUE), interleaved
for i in 1 2; do for v in 1 0; do
  SFSN_TRAIN_FAST_GLUE=$v python bench.py --training --batch 64 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('FAST_GLUE=$v B=64', d['value'], 'graph', d['config']['hip_graph_replay'].get('ms_per_step'))"
done; done
SFSN_TRAIN_FAST_GLUE=1 python bench.py --training --batch 16 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('FAST_GLUE=1 B=16', d['value'], 'graph', d['config']['hip_graph_replay'].get('ms_per_step'))"
