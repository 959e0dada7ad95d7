#!/bin/bash
# What a dK/dV step is made of: builds of the kernel with one part compiled out (timing only, the results are wrong).
#   here (no GPU):  bash tools/peel_dkv.sh build      -> musicgeneration_amd/libmgx_dkvpeel<mask>.so for every mask below
#   GPU box:        bash tools/peel_dkv.sh run [B]     -> one line per build: dkv ms (tools/attn_bench.py --parts 8)
# masks (MGX_DKV_PEEL, rel_attn_bwd.hip): 1 no E-fragment loads in the sweep | 2 no dS stores | 4 no skew (ds_bpermute; until the bpermute version: the LDS band round trip) |
#   8 no exponentials | 16 the q / dO tile prefetch always re-reads tile 0 (L2-resident) | 32 lse / delta are constants (no LDS reads of the statistics)
MASKS=${MASKS:-"0 1 2 4 8 16 32 63"}
if [ "$1" = build ]; then
  for m in $MASKS; do python3 -m musicgeneration_amd._build --variant dkvpeel$m -DMGX_DKV_PEEL=$m | tail -1; done
else
  B=${2:-32}
  for m in $MASKS; do
    printf "peel %2d  " $m
    MGX_LIB_PATH=musicgeneration_amd/libmgx_dkvpeel$m.so timeout -k 10 200 python3 tools/attn_bench.py --B $B --parts 8 --reps 10 --rounds 2 2>&1 | grep dkv | tail -1
  done
fi
