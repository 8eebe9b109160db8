#!/bin/bash
# training step at B = 64 against the row-block target of a layer call (SFSN_TRAIN_WGS: needs the -DSFSN_EXPERIMENTS build)
mkdir -p gpurun_out
export SFSN_LIB_PATH=$PWD/spiking_fullsubnet_amd/csrc_exp/libsfsn_hip.so
for w in 160 80 40 20; do
  SFSN_TRAIN_WGS=$w python bench.py --training --batch 64 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('WGS=$w', d['value'], d['roofline']['forward'], d['roofline']['backward'])"
done
