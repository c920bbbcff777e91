#!/bin/bash
# A/B/C... of several builds in ONE gpurun call (boxes differ by ~5 %): every tbv_slam_public_amd/variants/*.so against the
# current libcfear_hip.so ("cur"), alternating, REPS rounds.  Prints value, ms per frame batch and the kernel breakdown.
#   BENCH_ARGS="--dense --steps 3 --frames-per-step 8" / "--bins-major ..." select the scene population (default: the headline).
#   build a variant:  make -C tbv_slam_public_amd/csrc EXTRA=-DMT_SOMETHING && cp tbv_slam_public_amd/libcfear_hip.so tbv_slam_public_amd/variants/something.so
L=tbv_slam_public_amd
cp $L/libcfear_hip.so /tmp/cur.so
run() {
  python bench.py --no-cpu-baseline --no-extras ${BENCH_ARGS:---steps 8} 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-14s' % '$1', 'value', round(d['value']), 'ms/batch', round(d['ms_per_frame_batch'],4), {k: round(v['ms_per_frame_batch'],4) for k,v in d['kernel_breakdown'].items() if v['ms_per_frame_batch'] > 0.02})"
}
for rep in $(seq ${REPS:-2}); do
  cp /tmp/cur.so $L/libcfear_hip.so; run cur
  for v in $L/variants/*.so; do cp $v $L/libcfear_hip.so; run $(basename $v .so); done
done
cp /tmp/cur.so $L/libcfear_hip.so
