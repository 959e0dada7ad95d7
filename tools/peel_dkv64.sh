#!/bin/bash
# Timing-only "peel" builds of the generated dK/dV loop (gen_dkv_asm.py PEEL bits; results are wrong), same box, interleaved with the product:
#   here:     for b in 1 2 4 8 16 32 64 128 7; do MGX_DKV64_PEEL=$b python -m musicgeneration_amd._build --variant k64peel$b -DMGX_DKV64_PEEL; done
#   GPU box:  bash tools/peel_dkv64.sh "1 2 4 8 16 32 64 128 7" [B=64] [rounds=2]
BITS=${1:?bits}; B=${2:-64}; R=${3:-2}
for r in $(seq $R); do
  MGX_LIB_PATH=musicgeneration_amd/libmgx.so timeout -k 10 100 python3 tools/attn_bench.py --B $B --parts 8 --reps 10 --rounds 1 2>&1 | grep "^dkv " | sed "s/^/product   /"
  for b in $BITS; do
    MGX_LIB_PATH=musicgeneration_amd/libmgx_k64peel$b.so timeout -k 10 100 python3 tools/attn_bench.py --B $B --parts 8 --reps 10 --rounds 1 2>&1 | grep "^dkv " | sed "s/^/peel $b   /"
  done
done
