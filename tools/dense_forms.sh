#!/bin/bash
# dense-row scenes: the matcher forms a frame batch of S streams can run on (context options), value and kernel breakdown
run() { python bench.py --no-cpu-baseline --no-extras --dense --streams ${S:-4096} --sequences 64 --steps 3 --frames-per-step 8 "$@" 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$*', 'value', round(d['value']), 'ms/batch', round(d['ms_per_frame_batch'],4), {k: round(v['ms_per_frame_batch'],4) for k,v in d['kernel_breakdown'].items() if v['ms_per_frame_batch'] > 0.02}, 'failed', d.get('failed_registrations'))"; }
run
run --ctx-option MATCHER_WAVES=16 --ctx-option MATCHER_LDS_KB=160
run --ctx-option MATCHER_WAVES=8 --ctx-option MATCHER_LDS_KB=160
run --ctx-option MATCHER_WAVES=8 --ctx-option MATCHER_LDS_KB=80
