#!/bin/bash
# the driver's command (--steps 20 --warmup 5) against the number of forwards in flight
for i in 1 2 3; do
for n in 12 8 6 24 16; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --inflight $n --no-cpu-baseline --no-training-leg --no-streaming-leg --no-w16-leg 2>/dev/null | python -c "
import json,sys; l=json.loads(sys.stdin.readline()); c=l['config']; print('inflight $n steps', l['steps'], 'value', l['value'], 'lean', c['no_layer_outputs']['timed_region']['value'])"
done; done
