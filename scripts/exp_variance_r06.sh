#!/bin/bash
# run-to-run spread of the driver's command (timed region only)
for i in $(seq 1 ${1:-20}); do
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-phase-a --no-cpu-baseline --no-training-leg --no-streaming-leg --no-w16-leg 2>/dev/null | python -c "
import json,sys; l=json.loads(sys.stdin.readline()); print('run $i value', l['value'], 'ms', l['ms_per_step'])"
done
