#!/bin/bash
# strict forward: prep-ahead (full-band features + input products of all chunks up front) with deeper full-band rings (exp libs dg6 / dg9)
run() { # name, env...
  n=$1; shift
  env "$@" python bench.py --steps 24 --no-cpu-baseline --no-training-leg --no-streaming-leg --no-w16-leg 2>/dev/null | python -c "
import json,sys; l=json.loads(sys.stdin.readline()); c=l['config']; print('$n', 'strict', c['single_stream']['ms_per_step'], 'lean strict', c['no_layer_outputs']['single_stream']['ms_per_step'], 'value', l['value'])"
}
for i in 1 2; do
  run base X=1
  run prep SFSN_PREP_AHEAD=1
  run dg6 SFSN_LIB_PATH=$PWD/spiking_fullsubnet_amd/csrc_dg6/libsfsn_hip.so
  run dg6+prep SFSN_LIB_PATH=$PWD/spiking_fullsubnet_amd/csrc_dg6/libsfsn_hip.so SFSN_PREP_AHEAD=1
  run dg9 SFSN_LIB_PATH=$PWD/spiking_fullsubnet_amd/csrc_dg9/libsfsn_hip.so
  run dg9+prep SFSN_LIB_PATH=$PWD/spiking_fullsubnet_amd/csrc_dg9/libsfsn_hip.so SFSN_PREP_AHEAD=1
done
