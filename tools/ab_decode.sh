#!/bin/bash
# the fused decode in the pipeline: A/B of the bins-major bench on sparse and dense scenes (context option FUSED_DECODE = 0 = rotation
# kernel + row sweep always; default = fused decode on radar-like sweeps, the two-kernel route once the decode reports dense ones)
for DENSE in "" "--dense --streams 1024 --sequences 64"; do
for v in 1 0; do
  if [ $v = 1 ]; then export BENCH_FUSED_DECODE=0; else unset BENCH_FUSED_DECODE; fi
  python bench.py ${BENCH_FUSED_DECODE:+--ctx-option FUSED_DECODE=$BENCH_FUSED_DECODE} --no-cpu-baseline --no-extras --bins-major --steps 3 --frames-per-step 8 --streams 2048 $DENSE 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('dense=[$DENSE] two_kernel_route=$v', round(d['value']), round(d['ms_per_frame_batch'],3), {k: round(v['ms_per_frame_batch'],3) for k,v in d.get('kernel_breakdown',{}).items()})"
done
done
