#!/bin/bash
# the fused decode in the pipeline: parity tests, then A/B of the bins-major bench (CFEAR_NO_FUSED_DECODE=1 = rotation kernel + row sweep)
python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for v in 1 0; do
  if [ $v = 1 ]; then export CFEAR_NO_FUSED_DECODE=1; else unset CFEAR_NO_FUSED_DECODE; fi
  python bench.py --no-cpu-baseline --no-extras --bins-major --streams 2048 --steps 3 --frames-per-step 8 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('two_kernel_route=$v', round(d['value']), round(d['ms_per_frame_batch'],3), {k: round(v['ms_per_frame_batch'],3) for k,v in d.get('kernel_breakdown',{}).items()}, d['roofline']['kernel'], round(d['roofline']['frac'],3))"
done
