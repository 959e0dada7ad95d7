#!/bin/bash
# One GPU call that regenerates the round's profile artefacts under gpurun_out/prof/ (copy the ones to keep into profiles/):
#   kernel trace of the bench step, PMC utilisation of the attention kernels, PMC traffic, the bench line, the decode trace,
#   A/B timings of every attention-kernel variant in the tree, in-kernel stamps of the 64-row kernels.
# usage (GPU box, repo root): bash tools/prof_round.sh <tag> [batch]      e.g. r03 64   (batch: default = the bench default, 64)
set -o pipefail
export TMPDIR=/tmp
TAG=${1:-rXX}
B=${2:-64}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
echo "== bench line"; timeout -k 10 900 python3 bench.py --batch $B --steps 20 --warmup 5 2> $OUT/${TAG}_bench.err > $OUT/${TAG}_bench_line.json || { tail -5 $OUT/${TAG}_bench.err; exit 1; }
cut -c1-400 $OUT/${TAG}_bench_line.json
echo "== kernel trace of the step"
rm -rf /tmp/kt && (cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python3 $ROOT/bench.py --batch $B --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-decode > /tmp/kt.log 2>&1) || { tail -5 /tmp/kt.log; exit 1; }
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_bench_cfg2_b${B}_kernel_stats.csv
python3 tools/stats_summary.py $OUT/${TAG}_bench_cfg2_b${B}_kernel_stats.csv 7 > /tmp/kt_summary.txt; head -14 /tmp/kt_summary.txt
echo "== PMC utilisation of the attention kernels"; bash tools/pmc_attn.sh $B $OUT/${TAG}_pmc_attn_b${B}.json > /tmp/pmc_attn.txt || { tail -5 /tmp/pmc_attn.txt; exit 1; }; tail -8 /tmp/pmc_attn.txt
echo "== PMC traffic"; BATCH=$B bash tools/traffic.sh $OUT/${TAG}_traffic_cfg2_b${B}.json > /tmp/traffic.txt || { tail -5 /tmp/traffic.txt; exit 1; }; head -12 /tmp/traffic.txt
echo "== attention variants A/B"; timeout -k 10 300 python3 tools/attn_bench.py --B $B --parts 125 --reps 10 --rounds 2 2>&1 | grep -v amdgpu.ids | tee $OUT/${TAG}_attn_variants.txt
for v in 81000 160000; do echo "MGX_FWD_LDS=$v"; MGX_FWD_LDS=$v timeout -k 10 300 python3 tools/attn_bench.py --B $B --parts 1 --reps 10 --rounds 2 2>&1 | grep fwd32; done | tee -a $OUT/${TAG}_attn_variants.txt
if [ -f musicgeneration_amd/libmgx_stamp.so ]; then
  echo "== stamps"; for k in fwd; do MGX_LIB_PATH=musicgeneration_amd/libmgx_stamp.so timeout -k 10 300 python3 tools/attn64_stamp.py --kernel $k 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_stamps_${k}64.txt; tail -3 $OUT/${TAG}_stamps_${k}64.txt; done
fi
echo "== decode trace"
rm -rf /tmp/dt && (cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dt -- python3 $ROOT/tools/decode_bench.py --L 8192 > /tmp/dt.log 2>&1) || { tail -5 /tmp/dt.log; exit 1; }
cp $(find /tmp/dt -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_decode_cfg5_L8192_kernel_stats.csv
tail -1 /tmp/dt.log
