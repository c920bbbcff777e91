#!/bin/bash
# phase split of coral_kernel (a few jobs) with the -DCFEAR_CORAL_TIMING build; the release .so is restored afterwards
cd tbv_slam_public_amd/csrc; cp ../libcfear_hip.so /tmp/keep.so; touch coral.hip; make EXTRA=-DCFEAR_CORAL_TIMING 2>&1 | grep -E "error"; cd ../..
python bench.py --workload verify --steps 1 --warmup 1 2>&1 | grep "coral job" | tail -8
cp /tmp/keep.so tbv_slam_public_amd/libcfear_hip.so
