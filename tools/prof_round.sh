#!/bin/bash
# One GPU call that regenerates the round's profile artefacts under gpurun_out/prof/ (copy the ones to keep into profiles/):
#   kernel trace of the bench step, PMC utilisation of the attention kernels, PMC traffic, the bench line, the decode trace,
#   timings of every attention kernel alone, the cfg4 kernel trace.
# usage (GPU box, repo root): bash tools/prof_round.sh <tag> [batch]      e.g. r03 64   (batch: default = the bench default, 128 since round 6)
set -o pipefail
export TMPDIR=/tmp
TAG=${1:-rXX}
B=${2:-128}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof
export TMPD=/tmp/mgx_prof_$$; mkdir -p $TMPD
mkdir -p $OUT
echo "== bench line"; timeout -k 10 900 python3 bench.py --batch $B --steps 20 --warmup 5 2> $OUT/${TAG}_bench.err > $OUT/${TAG}_bench_line.json || { tail -5 $OUT/${TAG}_bench.err; exit 1; }
cut -c1-400 $OUT/${TAG}_bench_line.json
echo "== kernel trace of the step"
rm -rf $TMPD/kt && (cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $TMPD/kt -- python3 $ROOT/bench.py --batch $B --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-decode --no-cfg4 > $TMPD/kt.log 2>&1) || { tail -5 $TMPD/kt.log; exit 1; }
cp $(find $TMPD/kt -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_bench_cfg2_b${B}_kernel_stats.csv
python3 tools/stats_summary.py $OUT/${TAG}_bench_cfg2_b${B}_kernel_stats.csv 7 > $TMPD/kt_summary.txt; head -14 $TMPD/kt_summary.txt
echo "== PMC utilisation of the attention kernels"; bash tools/pmc_attn.sh $B $OUT/${TAG}_pmc_attn_b${B}.json > $TMPD/pmc_attn.txt || { tail -5 $TMPD/pmc_attn.txt; exit 1; }; tail -8 $TMPD/pmc_attn.txt
echo "== PMC traffic"; BATCH=$B bash tools/traffic.sh $OUT/${TAG}_traffic_cfg2_b${B}.json > $TMPD/traffic.txt || { tail -5 $TMPD/traffic.txt; exit 1; }; head -12 $TMPD/traffic.txt
echo "== attention variants A/B"; timeout -k 10 300 python3 tools/attn_bench.py --B $B --parts 125 --reps 10 --rounds 2 2>&1 | grep -v amdgpu.ids | tee $OUT/${TAG}_attn_variants.txt
echo "== cfg4 (BASELINE configs[3] at its single-GPU share) kernel trace"
rm -rf $TMPD/k4 && (cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $TMPD/k4 -- python3 $ROOT/bench.py --workload cfg4 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-decode > $TMPD/k4.log 2>&1) || { tail -5 $TMPD/k4.log; exit 1; }
cp $(find $TMPD/k4 -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_bench_cfg4_b4_kernel_stats.csv
python3 tools/stats_summary.py $OUT/${TAG}_bench_cfg4_b4_kernel_stats.csv 7 | head -12
echo "== decode trace"
rm -rf $TMPD/dt && (cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $TMPD/dt -- python3 $ROOT/tools/decode_bench.py --L 8192 > $TMPD/dt.log 2>&1) || { tail -5 $TMPD/dt.log; exit 1; }
cp $(find $TMPD/dt -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_decode_cfg5_L8192_kernel_stats.csv
tail -1 $TMPD/dt.log
