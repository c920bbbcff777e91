#!/bin/bash
# one SQ counter pass over a small bench run; prints per-kernel totals per dispatch for this library's kernels
set -u
ROOT=$(pwd); OUT=/tmp/pq; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-extras --no-profile --steps 1 --warmup 1 --frames-per-step 2 --streams ${STREAMS:-512} --sequences 32 ${BENCH_ARGS:-}"
cd /tmp
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $OUT/pmc_sq -o p -- python $ROOT/bench.py $ARGS > /dev/null 2> $OUT/pmc.err
cd $ROOT
python tools/summarize_pmc.py $OUT 2>&1 | grep -E "surface_|matcher_kernel|kstrongest_rows" | grep -E "INSTS_VALU|INSTS_LDS|INSTS_SALU|WAVE_CYCLES|WAIT_INST_ANY|SQ_WAVES|BUSY_CYCLES"
rm -rf $OUT
