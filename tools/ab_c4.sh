#!/bin/bash
# A/B of builds on config4 alone (tools/c4_quick.py): tbv_slam_public_amd/variants/*.so against the current library
L=tbv_slam_public_amd
cp $L/libcfear_hip.so /tmp/cur.so
for rep in $(seq ${REPS:-2}); do
  cp /tmp/cur.so $L/libcfear_hip.so; echo -n "cur   "; python tools/c4_quick.py ${STREAMS:-512} 2>&1 | tail -1
  for v in $L/variants/*.so; do cp $v $L/libcfear_hip.so; echo -n "$(basename $v .so)  "; python tools/c4_quick.py ${STREAMS:-512} 2>&1 | tail -1; done
done
cp /tmp/cur.so $L/libcfear_hip.so
