#!/bin/bash
# A/B over the three scan populations in ONE gpurun call (base .so vs current): street scenes (headline), dense rows,
# CA-CFAR clouds on [bins][azimuths] input (config4)
L=tbv_slam_public_amd
cp $L/libcfear_hip.so /tmp/new.so
hd() { python bench.py --no-cpu-baseline --no-extras --steps 8 $2 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1 $2', 'value', round(d['value']), 'ms/batch', round(d['ms_per_frame_batch'],4), {k: round(v['ms_per_frame_batch'],4) for k,v in d['kernel_breakdown'].items() if k.startswith('surface') or k.startswith('register')})"; }
c4() { BENCH_SKIP=dense,single,host,mulran python bench.py --no-cpu-baseline --steps 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())['config4_cacfar_kvarntorp']; print('$1 config4', 'value', round(d['value']), 'ms/batch', round(d['ms_per_frame_batch'],4), {k: round(v,4) for k,v in d['kernel_breakdown'].items() if v})"; }
for rep in 1 2; do
  for t in base new; do
    if [ $t = base ]; then cp $L/libcfear_hip_base.so $L/libcfear_hip.so; else cp /tmp/new.so $L/libcfear_hip.so; fi
    hd $t; hd $t --dense; c4 $t
  done
done
cp /tmp/new.so $L/libcfear_hip.so
