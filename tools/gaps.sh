#!/bin/bash
# where the GPU idles inside a frame batch: average gap in front of every kernel of this library (kernel trace time stamps)
set -u
ROOT=$(pwd); OUT=/tmp/gaps; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
rocprofv3 --output-format csv --kernel-trace --memory-copy-trace -d $OUT -o g -- python $ROOT/bench.py --no-cpu-baseline --no-extras --no-profile --steps 3 --warmup 1 > /dev/null 2> $OUT/err.txt
cd $ROOT
python - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/gaps/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows]
mc = glob.glob('/tmp/gaps/**/*memory_copy_trace.csv', recursive=True)
copies = []
if mc:
    for r in csv.DictReader(open(mc[0])):
        copies.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY ' + r.get('Direction', '')))
allv = sorted(ev + copies)
ours = ('kstrongest_rows', 'surface_prep', 'surface_sort', 'surface_points', 'surface_finish', 'matcher_kernel')
# last quarter of the run = steady state
k0 = [i for i, e in enumerate(allv) if 'matcher_kernel' in e[2]]
start = k0[len(k0) // 2]
gap = collections.defaultdict(list); dur = collections.defaultdict(list)
prev_end = None
for s, e, n in allv[start:]:
    key = next((o for o in ours if o in n), n[:24])
    if prev_end is not None:
        gap[key].append(max(0, s - prev_end))
    dur[key].append(e - s)
    prev_end = max(prev_end or 0, e)
tot_gap = 0
for k in gap:
    g = sum(gap[k]) / len(gap[k]); tot_gap += sum(gap[k])
    print("%-28s n=%5d  avg gap before %8.1f us   avg duration %8.1f us" % (k, len(gap[k]), g / 1e3, sum(dur[k]) / len(dur[k]) / 1e3))
nreg = len(gap.get('matcher_kernel', [1]))
print("idle per frame batch: %.1f us" % (tot_gap / nreg / 1e3))
PY
rm -rf $OUT
