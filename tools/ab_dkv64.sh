#!/bin/bash
# Same-box A/B of dK/dV builds (GPU box): bash tools/ab_dkv64.sh "product k64stplain ..." [B=64] [rounds=2]; each process runs 3 timed rounds, the last is quoted
LIBS=${1:?names}; B=${2:-64}; R=${3:-2}
for r in $(seq $R); do
  for n in $LIBS; do
    if [ $n = product ]; then L=musicgeneration_amd/libmgx.so; else L=musicgeneration_amd/libmgx_$n.so; fi
    MGX_LIB_PATH=$L timeout -k 10 100 python3 tools/attn_bench.py --B $B --parts 8 --reps 10 --rounds 3 2>&1 | grep "^dkv " | tail -1 | sed "s/^/$n   /"
  done
done
