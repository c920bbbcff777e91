#!/bin/bash
# SQ counters + HBM fetch of the CA-CFAR kernels (512 Kvarntorp sweeps per dispatch): instruction mix and issue utilisation
#   bash tools/cfar_pmc.sh 512 [--bins-major]
set -u
ROOT=$(pwd); OUT=/tmp/cpmc; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $OUT/pmc_sq -o p -- python $ROOT/tools/cfar_events.py ${1:-512} ${2:-} > /dev/null 2> $OUT/pmc.err
rocprofv3 --output-format csv --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA -d $OUT/pmc_mem -o p -- python $ROOT/tools/cfar_events.py ${1:-512} ${2:-} > /dev/null 2> $OUT/pmc2.err
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -o p -- python $ROOT/tools/cfar_events.py ${1:-512} ${2:-} > /dev/null 2> $OUT/pmc3.err
rocprofv3 --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum -d $OUT/pmc_tc -o p -- python $ROOT/tools/cfar_events.py ${1:-512} ${2:-} > /dev/null 2> $OUT/pmc4.err
cd $ROOT
python tools/summarize_pmc.py $OUT 2>&1 | grep -E "cacfar_|==" 
tail -2 $OUT/pmc2.err
rm -rf $OUT
