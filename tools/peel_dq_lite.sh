#!/bin/bash
# What a dq_lite step is made of: builds of the kernel with one part compiled out (timing only, the results are wrong).
#   here (no GPU):  bash tools/peel_dq_lite.sh build      -> musicgeneration_amd/libmgx_dqlpeel<mask>.so for every mask below
#   GPU box:        bash tools/peel_dq_lite.sh run [B]     -> one line per build: dq_lite ms (tools/attn_bench.py --parts 4)
# masks (MGX_DQL_PEEL, rel_attn_bwd.hip): 1 dS^T patch stores | 2 three quarters of the band stores | 4 half of dS K |
#   8 half of dS_rel ErT | 16 K / ErT ring refills (the step then only streams dS) | 31 everything
MASKS="0 1 2 4 8 16 31"
if [ "$1" = build ]; then
  for m in $MASKS; do python3 -m musicgeneration_amd._build --variant dqlpeel$m -DMGX_DQL_PEEL=$m | tail -1; done
else
  B=${2:-32}
  for m in $MASKS; do
    printf "peel %2d  " $m
    MGX_LIB_PATH=musicgeneration_amd/libmgx_dqlpeel$m.so timeout -k 10 200 python3 tools/attn_bench.py --B $B --parts 4 --reps 10 --rounds 2 2>&1 | grep dq_lite | tail -1
  done
fi
