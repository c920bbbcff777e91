#!/bin/bash
# quick PMC look at the per-frame kernels (small batch); raw rocprofv3 output is deleted, only the summary is kept
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pc; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-extras --steps 1 --warmup 1 --frames-per-step 2 --streams ${STREAMS:-512} --sequences ${SEQS:-64}"   # (streams / sequences <= the ring's 64 frames, or bench.py refuses)
cd /tmp
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $OUT/pmc_sq -o p -- python $ROOT/bench.py $ARGS > /dev/null 2> $OUT/pmc.err
rocprofv3 --output-format csv --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS -d $OUT/pmc_mem -o p -- python $ROOT/bench.py $ARGS > /dev/null 2> $OUT/pmc2.err
rocprofv3 --output-format csv --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_tc -o p -- python $ROOT/bench.py $ARGS > /dev/null 2> $OUT/pmc3.err
rocprofv3 --output-format csv --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum SQ_INST_CYCLES_VALU SQ_INST_CYCLES_SALU -d $OUT/pmc_tlb -o p -- python $ROOT/bench.py $ARGS > /dev/null 2> $OUT/pmc4.err
cd $ROOT
python tools/summarize_pmc.py $OUT > $OUT/pmc_summary.txt 2>&1
find $OUT -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
grep -E "surface_|register|kstrongest|==" $OUT/pmc_summary.txt | head -120
tail -3 $OUT/pmc4.err
