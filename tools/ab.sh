#!/bin/bash
# Same-box A/B of library builds (GPU box, repo root): the attention micro-benchmark once per library and round, the libraries
# interleaved so that clock / thermal drift of the box hits all of them alike.
#   bash tools/ab.sh "base new" [parts=8] [B=64] [rounds=3] [grep-pattern]
#     names -> musicgeneration_amd/libmgx_<name>.so ("product" = libmgx.so); parts = tools/attn_bench.py --parts
LIBS=${1:?names}; PARTS=${2:-8}; B=${3:-64}; R=${4:-3}; PAT=${5:-.}
for r in $(seq $R); do
  for n in $LIBS; do
    if [ $n = product ]; then L=musicgeneration_amd/libmgx.so; else L=musicgeneration_amd/libmgx_$n.so; fi
    MGX_LIB_PATH=$L timeout -k 10 200 python3 tools/attn_bench.py --B $B --parts $PARTS --reps 10 --rounds 1 2>&1 | grep -v amdgpu.ids | grep -E "$PAT" | sed "s/^/$n  /"
  done
done
