#!/bin/bash
# usage: tools/pmc_any.sh <kernel-substring> <python script> [args]   (runs on the GPU box, from the repo root)
export TMPDIR=/tmp
K=$1; shift
ROOT=$(pwd)
cd /tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_MISC" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM" "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $set | cut -c1-14 | tr " " "_")
  rm -rf /tmp/pmc_$n
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_$n -- python3 $ROOT/"$@" > /tmp/pmc_$n.log 2>&1 || { tail -3 /tmp/pmc_$n.log; }
  f=$(find /tmp/pmc_$n -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$K" <<PY
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
acc=collections.defaultdict(list)
for r in rows:
    if sys.argv[2] in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items(): print(f"{k:28s} {sum(v)/len(v):16.0f}")
PY
done
