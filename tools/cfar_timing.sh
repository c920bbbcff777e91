#!/bin/bash
# phase split of cacfar_rows_kernel (a few rows) with the -DCFEAR_CFAR_TIMING build; the release .so is restored afterwards
cd tbv_slam_public_amd/csrc; cp ../libcfear_hip.so /tmp/keep.so; touch filter.hip; make EXTRA=-DCFEAR_CFAR_TIMING 2>&1 | grep -E "error"; cd ../..
python tools/cfar_quick.py ${1:-512} 2>&1 | grep "cfar row" | tail -12
cp /tmp/keep.so tbv_slam_public_amd/libcfear_hip.so
