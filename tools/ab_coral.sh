#!/bin/bash
# A/B of coral_kernel builds in ONE gpurun call: tbv_slam_public_amd/libcfear_hip_<tag>.so for every tag given, alternating;
# prints the verification step time (coral_kernel is ~40 % of it).  The current libcfear_hip.so is restored afterwards.
L=tbv_slam_public_amd
cp $L/libcfear_hip.so /tmp/keep.so
for rep in 1 2; do
  for t in "$@"; do
    cp $L/libcfear_hip_$t.so $L/libcfear_hip.so
    python bench.py --workload verify --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$t', 'ms/step', round(d['ms_per_step'],4), 'value', round(d['value']))"
  done
done
cp /tmp/keep.so $L/libcfear_hip.so
