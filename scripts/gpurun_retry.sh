#!/bin/bash
# usage: scripts/gpurun_retry.sh TIMEOUT 'command' -- retries while the pod's GPU slots are busy (nothing is charged for those)
T=$1; shift
for i in $(seq 1 20); do
  out=$(/usr/local/graft/bin/gpurun --timeout "$T" -- "$@" 2>&1)
  if echo "$out" | grep -q "status=transient"; then sleep 60; continue; fi
  echo "$out"; exit 0
done
echo "$out"; exit 3
