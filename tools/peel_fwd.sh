#!/bin/bash
# What a forward step is made of: builds of rel_attn_fwd_kernel with one part compiled out (timing only, the results are wrong).
#   here (no GPU):  bash tools/peel_fwd.sh build   -> musicgeneration_amd/libmgx_fwdpeel<mask>.so
#   GPU box:        bash tools/peel_fwd.sh run [B]  -> one line per build: fwd32 ms (tools/attn_bench.py --parts 1)
# masks (MGX_FWD_PEEL, rel_attn_fwd.hip): 1 no E-fragment loads in the main loop | 2 no band round trip | 4 no exponentials |
#   8 K / V prefetch re-reads tile 0 | 15 all | 16 no parity XOR on the 16 band-store addresses | 32 no row-sum adds (16 v_add_f32):
#   64 no lazy-softmax redo branch (main loop and general body) | 128 no general steps after a main loop (the diagonal 128 x 128 block)
MASKS=${MASKS:-"0 1 2 4 8 15 16 32 64 128"}
if [ "$1" = build ]; then
  for m in $MASKS; do python3 -m musicgeneration_amd._build --variant fwdpeel$m -DMGX_FWD_PEEL=$m | tail -1; done
else
  B=${2:-32}
  for m in $MASKS; do
    printf "peel %2d  " $m
    MGX_LIB_PATH=musicgeneration_amd/libmgx_fwdpeel$m.so timeout -k 10 200 python3 tools/attn_bench.py --B $B --parts 1 --reps 10 --rounds 2 2>&1 | grep fwd32 | tail -1
  done
fi
