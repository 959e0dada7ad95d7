#!/bin/bash
# Same-box A/B of library builds on the GEMM micro-benchmark (GPU box, repo root), libraries interleaved:
#   bash tools/ab_gemm.sh "base new" [rounds=3] [M=131072]      names -> musicgeneration_amd/libmgx_<name>.so ("product" = libmgx.so)
LIBS=${1:?names}; R=${2:-3}; M=${3:-131072}
for r in $(seq $R); do
  for n in $LIBS; do
    if [ $n = product ]; then L=musicgeneration_amd/libmgx.so; else L=musicgeneration_amd/libmgx_$n.so; fi
    MGX_LIB_PATH=$L GEMM_M=$M timeout -k 10 200 python3 tools/gemm_bench.py 2>&1 | grep -E "per-step|grouped" | sed "s/^/$n  /"
  done
done
