#!/bin/bash
# phase split of matcher_kernel (workgroup 0) with the -DCFEAR_REG_TIMING build; the release .so is restored afterwards
cd tbv_slam_public_amd/csrc; cp ../libcfear_hip.so /tmp/keep.so; touch matcher.hip; make EXTRA=-DCFEAR_REG_TIMING 2>&1 | grep -E "error"; cd ../..
timeout 240 python bench.py --no-cpu-baseline --no-extras --no-profile --steps 1 --warmup 1 --frames-per-step 12 --streams ${STREAMS:-4096} ${BENCH_ARGS:-} 2>&1 | grep "matcher cycles" | tail -24
cp /tmp/keep.so tbv_slam_public_amd/libcfear_hip.so; touch tbv_slam_public_amd/csrc/matcher.hip
