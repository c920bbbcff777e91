#!/bin/bash
# A/B of two builds on the CA-CFAR pipeline in ONE gpurun call (boxes differ by a few per cent):
# tbv_slam_public_amd/libcfear_hip_base.so vs the current libcfear_hip.so, alternating; prints cacfar_rows per launch.
L=tbv_slam_public_amd
cp $L/libcfear_hip.so /tmp/new.so
for rep in 1 2 3; do
  cp $L/libcfear_hip_base.so $L/libcfear_hip.so; echo -n "base "; python tools/cfar_events.py 512 2>&1 | grep "mode 2"
  cp /tmp/new.so $L/libcfear_hip.so; echo -n "new  "; python tools/cfar_events.py 512 2>&1 | grep "mode 2"
done
