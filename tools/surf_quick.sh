#!/bin/bash
# quick look at the surface kernels: bench kernel breakdown with the release build, then the phase split of
# surface_sort_kernel with the -DCFEAR_SURF_TIMING build (the timing build is NOT kept: the release .so is restored)
python bench.py --no-cpu-baseline --no-extras ${BENCH_ARGS:-} 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('value', round(d['value']), 'ms/batch', round(d['ms_per_frame_batch'],4), {k: round(v['ms_per_frame_batch'],4) for k,v in d['kernel_breakdown'].items()}, 'frac', round(d['roofline']['frac'],3))"
cd tbv_slam_public_amd/csrc; cp ../libcfear_hip.so /tmp/keep.so; touch surface.hip; make EXTRA=-DCFEAR_SURF_TIMING 2>&1 | grep -E "error"; cd ../..
python bench.py --no-cpu-baseline --no-extras --no-profile --steps 3 ${BENCH_ARGS:-} 2>&1 | grep -B2 "surface_sort phases" | tail -3
cp /tmp/keep.so tbv_slam_public_amd/libcfear_hip.so
