#!/bin/bash
# rocprofv3 passes behind the numbers in bench.py / DESIGN.md.  Run on the GPU box from the repo root:
#   bash tools/profile.sh r05
# Raw output goes to /tmp (a kernel trace of torch's input generation is hundreds of MB); the judged summaries land under
# gpurun_out/profiles_<tag>/ (copy them into profiles/<tag>/).  Counters are collected in their own passes (no trace
# domains mixed in).
set -u
TAG=${1:-r05}
ROOT=$(pwd)
OUT=/tmp/prof_$TAG
SUM=$ROOT/gpurun_out/profiles_$TAG
rm -rf "$OUT" "$SUM"; mkdir -p "$OUT" "$SUM"
export TMPDIR=/tmp
make -C oracle -s
keep_ours() {   # kernel_stats.csv of a run -> only this library's kernels (torch's input generation dominates the raw file)
  python - "$1" "$2" <<'PY'
import csv, glob, sys
src = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
rows = list(csv.reader(open(src[0]))) if src else []
ours = ("kstrongest_rows_kernel", "kstrong_cloud_kernel", "kstrong_image", "kstrong_extract", "kstrong_select", "kstrongest_cols", "cacfar_", "surface_", "matcher_kernel", "assoc_kernel", "eval_kernel",
        "expand_candidates", "coral_kernel", "sc_descriptor", "sc_distance", "rotate_ccw", "compensate", "legacy_", "scan_grid", "cells_to_slab",
        "slab_to_cells", "closest_idx")
keep = [rows[0]] + [r for r in rows[1:] if any(k in r[0] for k in ours)] if rows else []
csv.writer(open(sys.argv[2], "w")).writerows(keep)
PY
}
# 1. kernel trace + stats of the default bench command (headline launches + the extra passes)
( cd /tmp && rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/bench" -o bench -- python "$ROOT/bench.py" --no-cpu-baseline --no-extras > "$SUM/bench_stdout.json" 2> "$OUT/bench.err" )
keep_ours "$OUT/bench" "$SUM/bench_kernel_stats.csv"; rm -rf "$OUT/bench"
( cd /tmp && rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/dense" -o d -- python "$ROOT/bench.py" --no-cpu-baseline --no-extras --dense --streams 1024 --sequences 64 --steps 4 --frames-per-step 8 > "$SUM/bench_dense_stdout.json" 2> "$OUT/dense.err" )
keep_ours "$OUT/dense" "$SUM/bench_dense_kernel_stats.csv"; rm -rf "$OUT/dense"
# 1b. the neighbouring workloads
( cd /tmp && rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/verify" -o v -- python "$ROOT/bench.py" --workload verify --steps 5 --warmup 2 > "$SUM/verify_stdout.json" 2> "$OUT/verify.err" )
keep_ours "$OUT/verify" "$SUM/verify_kernel_stats.csv"; rm -rf "$OUT/verify"
( cd /tmp && rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/lc" -o v -- python "$ROOT/bench.py" --workload loopclosure --steps 20 --warmup 3 > "$SUM/loopclosure_stdout.json" 2> "$OUT/lc.err" )
keep_ours "$OUT/lc" "$SUM/loopclosure_kernel_stats.csv"; rm -rf "$OUT/lc"
( cd /tmp && rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/binsmajor" -o b -- python "$ROOT/bench.py" --no-cpu-baseline --no-extras --bins-major --steps 3 --frames-per-step 8 > "$SUM/bins_major_stdout.json" 2> "$OUT/binsmajor.err" )
keep_ours "$OUT/binsmajor" "$SUM/bins_major_kernel_stats.csv"; rm -rf "$OUT/binsmajor"
# 1c. [bins][azimuths] sweeps: the fused decode (kstrong_image), its fall-backs and the two-kernel route on 2048 images
for MODE in fused two-pass tile lists; do
  FLAG=""; [ $MODE != fused ] && FLAG="--$MODE"
  ( cd /tmp && rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/decode_$MODE" -o d -- python "$ROOT/tools/decode_bench.py" 2048 $FLAG >> "$SUM/decode_stdout.txt" 2> "$OUT/decode_$MODE.err" )
  keep_ours "$OUT/decode_$MODE" "$SUM/decode_${MODE}_kernel_stats.csv"; rm -rf "$OUT/decode_$MODE"
done
( cd /tmp && timeout 900 rocprofv3 --output-format csv --pmc FETCH_SIZE -d "$OUT/pmc_decode_fetch" -o p -- python "$ROOT/tools/decode_bench.py" 512 --iters 2 > /dev/null 2> "$OUT/pmc_decode_fetch.err" )
( cd /tmp && timeout 900 rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d "$OUT/pmc_decode_sq" -o p -- python "$ROOT/tools/decode_bench.py" 512 --iters 2 > /dev/null 2> "$OUT/pmc_decode_sq.err" )
# 2. polar sweep alone: kernel trace on the three data sets, then PMC passes (FETCH_SIZE and WRITE_SIZE do not fit one pass)
for DATA in scene dense uniform; do
  ( cd /tmp && rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/filter_$DATA" -o f -- python "$ROOT/tools/bench_filter.py" --data $DATA > "$SUM/filter_${DATA}_stdout.txt" 2> "$OUT/filter_$DATA.err" )
  keep_ours "$OUT/filter_$DATA" "$SUM/filter_${DATA}_kernel_stats.csv"; rm -rf "$OUT/filter_$DATA"
done
# HBM traffic of the sweep AS IT RUNS IN THE HEADLINE PIPELINE (fused: 4-byte keys per kept bin, no cloud): the default bench
# command's configuration at a small batch (counter collection serialises every dispatch), kernel-filtered afterwards
PIPE_ARGS="--no-cpu-baseline --no-extras --steps 1 --warmup 1 --frames-per-step 4 --streams 512 --sequences 32"   # (512 streams: the matcher's 8-wavefront form; the pmc_pipe pass below runs 4096 streams: the regular form, four per CU)
( cd /tmp && timeout 900 rocprofv3 --output-format csv --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o p -- python "$ROOT/bench.py" $PIPE_ARGS > /dev/null 2> "$OUT/pmc_fetch.err" )
( cd /tmp && timeout 900 rocprofv3 --output-format csv --pmc WRITE_SIZE -d "$OUT/pmc_write" -o p -- python "$ROOT/bench.py" $PIPE_ARGS > /dev/null 2> "$OUT/pmc_write.err" )
export CFEAR_PMC_IMAGES=512
# CA-CFAR pipeline (BASELINE configs[4]): kernel trace + SQ counters of tools/cfar_events.py (512 Kvarntorp sweeps per launch)
( cd /tmp && rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/cacfar" -o c -- python "$ROOT/tools/cfar_events.py" 512 > "$SUM/cacfar_stdout.txt" 2> "$OUT/cacfar.err" )
keep_ours "$OUT/cacfar" "$SUM/cacfar_kernel_stats.csv"; rm -rf "$OUT/cacfar"
( cd /tmp && timeout 900 rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d "$OUT/pmc_cacfar" -o p -- python "$ROOT/tools/cfar_events.py" 512 > /dev/null 2> "$OUT/pmc_cacfar.err" )
( cd /tmp && timeout 900 rocprofv3 --output-format csv --pmc FETCH_SIZE -d "$OUT/pmc_cacfar_fetch" -o p -- python "$ROOT/tools/cfar_events.py" 512 > /dev/null 2> "$OUT/pmc_cacfar_fetch.err" )
# ... and on [range bins][azimuths] sweeps (the layout the Kvarntorp / Volvo / MulRan drivers deliver): cacfar_cols_kernel
( cd /tmp && rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/cacfar_bm" -o c -- python "$ROOT/tools/cfar_events.py" 512 --bins-major > "$SUM/cacfar_bins_major_stdout.txt" 2> "$OUT/cacfar_bm.err" )
keep_ours "$OUT/cacfar_bm" "$SUM/cacfar_bins_major_kernel_stats.csv"; rm -rf "$OUT/cacfar_bm"
( cd /tmp && timeout 900 rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d "$OUT/pmc_cacfar_cols" -o p -- python "$ROOT/tools/cfar_events.py" 512 --bins-major > /dev/null 2> "$OUT/pmc_cacfar_cols.err" )
( cd /tmp && timeout 900 rocprofv3 --output-format csv --pmc FETCH_SIZE -d "$OUT/pmc_cacfar_cols_fetch" -o p -- python "$ROOT/tools/cfar_events.py" 512 --bins-major > /dev/null 2> "$OUT/pmc_cacfar_cols_fetch.err" )
( cd /tmp && rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d "$OUT/pmc_sq" -o p -- python "$ROOT/tools/bench_filter.py" --iters 5 > /dev/null 2> "$OUT/pmc_sq.err" )
( cd /tmp && rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d "$OUT/pmc_sq_dense" -o p -- python "$ROOT/tools/bench_filter.py" --iters 5 --data dense > /dev/null 2> "$OUT/pmc_sq_dense.err" )
# whole pipeline under the SQ counters, SMALL run (counter collection serialises every dispatch)
( cd /tmp && timeout 600 rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d "$OUT/pmc_pipe" -o p -- python "$ROOT/bench.py" --no-cpu-baseline --no-extras --steps 1 --warmup 1 --frames-per-step 2 --streams 4096 --sequences 64 > /dev/null 2> "$OUT/pmc_pipe.err" )
python "$ROOT/tools/summarize_pmc.py" "$OUT" "$SUM/pmc_traffic.json" > "$SUM/pmc_summary.txt" 2>&1
rm -rf "$OUT"
ls -la "$SUM"
