#!/bin/bash
# usage: tools/pmc.sh <kernel-substring> <parts-bitmask>   (runs on the GPU box)
export TMPDIR=/tmp
K=$1; P=$2
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_MISC" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM"; do
  n=$(echo $set | cut -c1-14 | tr " " "_")
  rm -rf /tmp/pmc_$n
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_$n -- python tools/attn_bench.py --B 16 --reps 1 --parts $P > /dev/null 2>&1
  f=$(find /tmp/pmc_$n -name "*counter_collection.csv" | head -1)
  python - "$f" "$K" <<PY
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
acc=collections.defaultdict(list)
for r in rows:
    if sys.argv[2] in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items(): print(f"{k:28s} {sum(v)/len(v):16.0f}")
PY
done
