#!/bin/bash
# host timeline of cfear_verify_loop_candidates (the -DCFEAR_VERIFY_TIMING build; the release .so is restored afterwards)
cd tbv_slam_public_amd/csrc; cp ../libcfear_hip.so /tmp/keep.so; touch verify.hip; make EXTRA=-DCFEAR_VERIFY_TIMING 2>&1 | grep -E "error"; cd ../..
python bench.py --no-cpu-baseline --no-extras --workload verify --steps 4 2>&1 | grep "verify us" | tail -3
cp /tmp/keep.so tbv_slam_public_amd/libcfear_hip.so; touch tbv_slam_public_amd/csrc/verify.hip
