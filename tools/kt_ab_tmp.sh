export TMPDIR=/tmp
ROOT=$(pwd)
for k in 0 1; do
rm -rf /tmp/kt$k && (cd /tmp && MGX_RING4=$k MGX_LIB_PATH=$ROOT/musicgeneration_amd/libmgx_ringab.so timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt$k -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-decode --no-cfg4 > /tmp/kt$k.log 2>&1) || { tail -5 /tmp/kt$k.log; exit 1; }
echo "MGX_RING4=$k"; python3 tools/kt_by_call.py $(find /tmp/kt$k -name "*kernel_trace.csv" | head -1) 7
done
