#!/bin/bash
# rocprofv3 passes behind the numbers in bench.py / DESIGN.md.  Run on the GPU box from the repo root:
#   bash tools/profile.sh r01
# Writes raw output under gpurun_out/prof_<tag>/ and the judged summaries under gpurun_out/profiles_<tag>/
# (copy those into profiles/).  Counters are collected in their own passes (no trace domains mixed in).
set -u
TAG=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
SUM=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT" "$SUM"
export TMPDIR=/tmp
make -C oracle -s
# 1. kernel trace + stats of the bench command
( cd /tmp && rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/bench" -o bench -- python "$ROOT/bench.py" --no-cpu-baseline --steps 10 > "$SUM/bench_stdout.json" 2> "$OUT/bench.err" )
find "$OUT/bench" -name "*kernel_stats.csv" -exec cp {} "$SUM/bench_kernel_stats.csv" \;
# 1b. the neighbouring workloads: candidate verification (register + coral + cost-only) and the bins-major
#     input layout (adds rotate_ccw_rows_kernel)
( cd /tmp && rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/verify" -o v -- python "$ROOT/bench.py" --workload verify --steps 5 --warmup 2 > "$SUM/verify_stdout.json" 2> "$OUT/verify.err" )
find "$OUT/verify" -name "*kernel_stats.csv" -exec cp {} "$SUM/verify_kernel_stats.csv" \;
( cd /tmp && rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/binsmajor" -o b -- python "$ROOT/bench.py" --no-cpu-baseline --bins-major --steps 5 > "$SUM/bins_major_stdout.json" 2> "$OUT/binsmajor.err" )
find "$OUT/binsmajor" -name "*kernel_stats.csv" -exec cp {} "$SUM/bins_major_kernel_stats.csv" \;
# 2. polar sweep alone: kernel trace, then PMC passes (FETCH_SIZE and WRITE_SIZE do not fit one pass)
for DATA in scene uniform; do
  ( cd /tmp && rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/filter_$DATA" -o f -- python "$ROOT/tools/bench_filter.py" --data $DATA > "$SUM/filter_${DATA}_stdout.txt" 2> "$OUT/filter_$DATA.err" )
  find "$OUT/filter_$DATA" -name "*kernel_stats.csv" -exec cp {} "$SUM/filter_${DATA}_kernel_stats.csv" \;
done
( cd /tmp && rocprofv3 --output-format csv --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o p -- python "$ROOT/tools/bench_filter.py" --iters 5 > /dev/null 2> "$OUT/pmc_fetch.err" )
( cd /tmp && rocprofv3 --output-format csv --pmc WRITE_SIZE -d "$OUT/pmc_write" -o p -- python "$ROOT/tools/bench_filter.py" --iters 5 > /dev/null 2> "$OUT/pmc_write.err" )
( cd /tmp && rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d "$OUT/pmc_sq" -o p -- python "$ROOT/tools/bench_filter.py" --iters 5 > /dev/null 2> "$OUT/pmc_sq.err" )
# whole pipeline under the SQ counters, SMALL run (counter collection serialises every dispatch)
( cd /tmp && timeout 300 rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d "$OUT/pmc_pipe" -o p -- python "$ROOT/bench.py" --no-cpu-baseline --steps 3 --warmup 2 --streams 256 > /dev/null 2> "$OUT/pmc_pipe.err" )
python "$ROOT/tools/summarize_pmc.py" "$OUT" > "$SUM/pmc_summary.txt" 2>&1
ls -la "$SUM"
