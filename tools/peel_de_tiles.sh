#!/bin/bash
# What a de_tiles step is made of: builds of the kernel with one part compiled out (timing only, the results are wrong).
#   here (no GPU):  bash tools/peel_de_tiles.sh build   -> musicgeneration_amd/libmgx_detpeel<mask>.so
#   GPU box:        bash tools/peel_de_tiles.sh run [B]  -> one line per build: de_tiles ms (tools/attn_bench.py --parts 16)
# masks (MGX_DET_PEEL, rel_attn_bwd.hip): 1 a quarter of the scatter stores | 2 no products | 4 q re-read from row block 0 | 7 all
MASKS="0 1 2 4 7"
if [ "$1" = build ]; then
  for m in $MASKS; do python3 -m musicgeneration_amd._build --variant detpeel$m -DMGX_DET_PEEL=$m | tail -1; done
else
  B=${2:-32}
  for m in $MASKS; do
    printf "peel %2d  " $m
    MGX_LIB_PATH=musicgeneration_amd/libmgx_detpeel$m.so timeout -k 10 200 python3 tools/attn_bench.py --B $B --parts 16 --reps 10 --rounds 1 2>&1 | grep de_tiles | tail -1
  done
fi
