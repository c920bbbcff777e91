#!/bin/bash
# What do the hipEvents inside the timed region cost?  Alternates the default command (profile mode 2: events around the
# polar sweep only) with --no-profile (no events at all) on one box and prints both rates.
for rep in 1 2 3; do
  for flag in "" "--no-profile"; do
    python bench.py --no-cpu-baseline --no-extras $flag 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('events: %-12s value %.0f registrations/s  ms_per_frame_batch %.4f' % ('none' if '$flag' else 'sweep only', d['value'], d['ms_per_frame_batch']))"
  done
done
