for S in 96 100 255 256 300 511 1000 2047; do
python bench.py --no-cpu-baseline --no-extras --bins-major --streams $S --sequences 32 --steps 2 --frames-per-step 4 --warmup 1 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('streams', $S, round(d['value']), 'failed', d['failed_registrations'], {k: round(v['ms_per_frame_batch'],3) for k,v in d.get('kernel_breakdown',{}).items() if 'kstrong' in k or 'rotate' in k})"
done
