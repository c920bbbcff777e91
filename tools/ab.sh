#!/bin/bash
# A/B of two builds in ONE gpurun call (boxes differ by ~5 %): tbv_slam_public_amd/libcfear_hip_base.so vs the current
# libcfear_hip.so, alternating.  Prints the kernel breakdown of each run.
L=tbv_slam_public_amd
cp $L/libcfear_hip.so /tmp/new.so
run() {
  python bench.py --no-cpu-baseline --no-extras ${BENCH_ARGS:-} 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$1', 'value', round(d['value']), 'ms/batch', round(d['ms_per_frame_batch'],4), {k: round(v['ms_per_frame_batch'],4) for k,v in d['kernel_breakdown'].items()})"
}
for rep in 1 2; do
  cp $L/libcfear_hip_base.so $L/libcfear_hip.so; run base
  cp /tmp/new.so $L/libcfear_hip.so; run new
done
