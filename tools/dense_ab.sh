#!/bin/bash
# dense-scene A/B in ONE gpurun call: tbv_slam_public_amd/libcfear_hip_base.so vs the current libcfear_hip.so, alternating
L=tbv_slam_public_amd
cp $L/libcfear_hip.so /tmp/new.so
run() { python bench.py --dense --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$1', 'dense value', round(d['value']), 'ms/batch', round(d['ms_per_frame_batch'],4), {k: round(v['ms_per_frame_batch'],4) for k,v in d['kernel_breakdown'].items()}, 'failed', d.get('failed_registrations'))"; }
for rep in 1 2; do
  cp $L/libcfear_hip_base.so $L/libcfear_hip.so; run base
  cp /tmp/new.so $L/libcfear_hip.so; run new
done
