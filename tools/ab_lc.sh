#!/bin/bash
# A/B of builds on the loop-closure and verification workloads: tbv_slam_public_amd/variants/*.so against the current library
L=tbv_slam_public_amd
cp $L/libcfear_hip.so /tmp/cur.so
run() { for w in loopclosure verify; do python bench.py --no-cpu-baseline --no-extras --workload $w 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-10s %-12s' % ('$1', '$w'), 'value', round(d['value']), 'ms/step', round(d['ms_per_step'],4))"; done; }
for rep in $(seq ${REPS:-2}); do
  cp /tmp/cur.so $L/libcfear_hip.so; run cur
  for v in $L/variants/*.so; do cp $v $L/libcfear_hip.so; run $(basename $v .so); done
done
cp /tmp/cur.so $L/libcfear_hip.so
