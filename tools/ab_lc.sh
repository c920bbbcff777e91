#!/bin/bash
# A/B of register_kernel's users in ONE gpurun call (base .so vs current): loop-closure candidates (compact geometry), the
# 4096-stream headline forced onto register_kernel (CFEAR_NO_REG3=1) and one sequence alone (8-wavefront form)
L=tbv_slam_public_amd
cp $L/libcfear_hip.so /tmp/new.so
lc() { python bench.py --workload loopclosure --steps 30 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1 loopclosure ms/step', round(d['ms_per_step'],4), 'value', round(d['value']))"; }
hd() { CFEAR_NO_REG3=1 python bench.py --no-cpu-baseline --no-extras --steps 8 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1 headline on register_kernel: register', round(d['kernel_breakdown']['register']['ms_per_frame_batch'],4))"; }
for rep in 1 2; do
  cp $L/libcfear_hip_base.so $L/libcfear_hip.so; lc base; hd base; python tools/single_stream.py 2>&1 | tail -2 | head -1 | cut -c1-160
  cp /tmp/new.so $L/libcfear_hip.so; lc new; hd new; python tools/single_stream.py 2>&1 | tail -2 | head -1 | cut -c1-160
done
