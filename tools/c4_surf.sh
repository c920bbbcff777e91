#!/bin/bash
# config4 alone with the release build, then surface_sort_kernel's phase split with the -DCFEAR_SURF_TIMING build (restored after)
python tools/c4_quick.py ${STREAMS:-512} 2>&1 | tail -1
cd tbv_slam_public_amd/csrc; cp ../libcfear_hip.so /tmp/keep.so; touch surface.hip; make EXTRA=-DCFEAR_SURF_TIMING 2>&1 | grep -E "error"; cd ../..
python tools/c4_quick.py ${STREAMS:-512} 44 2>&1 | grep -A3 "surface routes" | tail -5
cp /tmp/keep.so tbv_slam_public_amd/libcfear_hip.so; touch tbv_slam_public_amd/csrc/surface.hip
