#!/bin/bash
# counters of the fused decode kernel (tools/decode_bench.py, 512 images); separate passes, no trace domains
export TMPDIR=/tmp
ROOT=$(pwd)
mkdir -p gpurun_out/decode
rm -f gpurun_out/decode/pmc.txt
run() {  # name counters...
  name=$1; shift
  ( cd /tmp && timeout 600 rocprofv3 --output-format csv --pmc "$@" -d /tmp/dpmc_$name -o p -- python "$ROOT/tools/decode_bench.py" 512 --iters 2 $FLAG > /tmp/dpmc_$name.log 2>&1 )
  python - /tmp/dpmc_$name <<'PY' | tee -a gpurun_out/decode/pmc.txt
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r["Kernel_Name"] for k in ("kstrongest", "rotate_ccw")):
            a = acc[r["Kernel_Name"][:48]][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, d in acc.items():
    print(k, {c: round(v / n, 1) for c, (n, v) in d.items()})
PY
}
FLAG="$1"
run sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum
run ta TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT
for f in /tmp/dpmc_*.log; do tail -n 3 "$f"; done | grep -i -E "error|invalid" | head
