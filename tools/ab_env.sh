#!/bin/bash
# A/B of one build under two environments in ONE gpurun call (boxes differ by ~5 %): `VAR=1` (base) vs unset (new), alternating.
#   VAR=CFEAR_NO_REG3 bash tools/ab_env.sh        BENCH_ARGS="--streams 2048" adds bench arguments
VAR=${VAR:-CFEAR_UNUSED}
run() {
  python bench.py --no-cpu-baseline --no-extras ${BENCH_ARGS:-} 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$1', 'value', round(d['value']), 'ms/batch', round(d['ms_per_frame_batch'],4), {k: round(v['ms_per_frame_batch'],4) for k,v in d['kernel_breakdown'].items()})"
}
for rep in 1 2; do
  env $VAR=1 bash -c "$(declare -f run); run base"
  run new
done
