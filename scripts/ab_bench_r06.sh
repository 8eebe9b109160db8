#!/bin/bash
# Round 6: interleaved A/B of bench.py's default command with an environment switch set / unset.
#   scripts/ab_bench_r06.sh NAME=VALUE [rounds] [tag]    e.g. SFSN_FUSED_V2=1 2 f3
sw="$1"; rounds="${2:-2}"; tag="${3:-ab}"
mkdir -p gpurun_out/r06ab
for i in $(seq 1 $rounds); do
  for on in 1 0; do
    if [ $on = 1 ]; then env "$sw" python bench.py --steps 60 --no-cpu-baseline --no-training-leg --no-streaming-leg --no-w16-leg > gpurun_out/r06ab/${tag}_${on}_$i.json 2>/dev/null
    else python bench.py --steps 60 --no-cpu-baseline --no-training-leg --no-streaming-leg --no-w16-leg > gpurun_out/r06ab/${tag}_${on}_$i.json 2>/dev/null; fi
    python - <<PY
import json
l = json.load(open("gpurun_out/r06ab/${tag}_${on}_$i.json")); c = l["config"]; n = c["no_layer_outputs"]
print("$sw", "SET" if $on else "unset", "run $i: value", l["value"], "strict", c["single_stream"]["ms_per_step"], "| lean region", n["timed_region"]["value"], "lean strict", n["single_stream"]["ms_per_step"])
PY
  done
done
