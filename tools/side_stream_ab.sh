#!/bin/bash
# Round 6: the two-stream backward against one stream, same box.  GPU box, repo root:
#   bash tools/side_stream_ab.sh            -> step times of the splits + kernel traces and tools/side_overlap.py tables under gpurun_out/
# (under rocprofv3 a process that created CU-masked streams dies with SIGSEGV in __cxa_finalize AFTER its trace is written: the
#  return code of those runs is ignored here)
export TMPDIR=/tmp
ROOT=$(pwd)
B=${1:-64}
one() { timeout -k 10 200 python3 bench.py --batch $B --no-cfg4 --no-cpu-baseline --no-decode --no-kernel-timing $1 2>gpurun_out/side_ab.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$1]', round(d['ms_per_step'],3), 'ms', round(d['value']/1e6,3), 'M events/s')" || tail -5 gpurun_out/side_ab.err; }
mkdir -p gpurun_out
for a in "" "--side-cus 64" "--side-cus 96" "--side-cus 128" "--side-cus 32" "--side-cus 64 --side-work de" "--side-cus 32 --side-work de" "--side-cus 64 --side-shared" ""; do one "$a"; done 2>&1 | tee gpurun_out/side_ab.txt
for v in one side64 side64de; do
  case $v in one) A="";; side64) A="--side-cus 64";; side64de) A="--side-cus 64 --side-work de";; esac
  rm -rf /tmp/kt_$v
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$v -- python3 $ROOT/bench.py --batch $B --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-decode --no-cfg4 $A > $ROOT/gpurun_out/side_kt_$v.log 2>&1)
done
python3 tools/side_overlap.py /tmp/kt_one /tmp/kt_side64 | tee gpurun_out/side64_overlap.txt
python3 tools/side_overlap.py /tmp/kt_one /tmp/kt_side64de | tee gpurun_out/side64de_overlap.txt
