#!/bin/bash
# Round 6: interleaved A/B of a library variant (scripts/build_exp_lib.sh NAME "...") against the product's build:
#   strict forward (scripts/exp_projdf_r06.py's strict figure via bench.py single_stream) and the timed region.
#   scripts/ab_lib_r06.sh NAME [rounds]
name="$1"; rounds="${2:-2}"
mkdir -p gpurun_out/r06ab
for i in $(seq 1 $rounds); do
  for on in 1 0; do
    if [ $on = 1 ]; then export SFSN_LIB_PATH=$PWD/spiking_fullsubnet_amd/csrc_$name/libsfsn_hip.so; else unset SFSN_LIB_PATH; fi
    python bench.py --steps ${STEPS:-60} --no-cpu-baseline --no-training-leg --no-streaming-leg --no-w16-leg > gpurun_out/r06ab/lib_${name}_${on}_$i.json 2>/dev/null
    python - <<PY
import json
l = json.load(open("gpurun_out/r06ab/lib_${name}_${on}_$i.json")); c = l["config"]; n = c["no_layer_outputs"]; r = l["roofline"]
print("lib", "$name" if $on else "product", "run $i: value", l["value"], "strict", c["single_stream"]["ms_per_step"], "| pair launch ms", r["sub_band_scan_single_forward"]["launch_ms"], "fb stack ms", r["full_band_stack"]["launch_ms"], "| lean region", n["timed_region"]["value"], "lean strict", n["single_stream"]["ms_per_step"])
PY
  done
done
