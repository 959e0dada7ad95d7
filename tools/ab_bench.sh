#!/bin/bash
# Same-box A/B of library builds on the whole training step (GPU box, repo root): bench.py once per library and round, interleaved.
#   bash tools/ab_bench.sh "base product" [rounds=2] [extra bench.py args]
LIBS=${1:?names}; shift; R=${1:-2}; [ $# -gt 0 ] && shift
for r in $(seq $R); do
  for n in $LIBS; do
    if [ $n = product ]; then L=musicgeneration_amd/libmgx.so; else L=musicgeneration_amd/libmgx_$n.so; fi
    MGX_LIB_PATH=$PWD/$L timeout -k 10 300 python3 bench.py --no-cfg4 --no-cpu-baseline --no-decode --no-kernel-timing "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', round(d['ms_per_step'],3), 'ms', round(d['value']/1e6,3), 'M events/s')"
  done
done
