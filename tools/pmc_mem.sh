#!/bin/bash
# usage: tools/pmc_mem.sh <kernel-substring> <python script> [args]   memory-path counters (runs on the GPU box)
export TMPDIR=/tmp
K=$1; shift
ROOT=$(pwd)
cd /tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_BUSY_avr" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_TAG_STALL_sum"; do
  n=$(echo $set | cut -c1-14 | tr " " "_")
  rm -rf /tmp/pmc_$n
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_$n -- python3 $ROOT/"$@" > /tmp/pmc_$n.log 2>&1 || { tail -3 /tmp/pmc_$n.log; }
  f=$(find /tmp/pmc_$n -name "*counter_collection.csv" | head -1)
  [ -z "$f" ] && { echo "no output for: $set"; tail -2 /tmp/pmc_$n.log; continue; }
  python3 - "$f" "$K" <<PY
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
acc=collections.defaultdict(list)
for r in rows:
    if sys.argv[2] in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items(): print(f"{k:34s} {sum(v)/len(v):16.0f}")
PY
done
