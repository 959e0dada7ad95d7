#!/bin/bash
# One rocprofv3 --pmc pass over a python script, averaged per launch for the kernels whose name contains a substring
# (GPU box, repo root):   bash tools/pmc_one.sh <kernel-substring> "<COUNTER ...>" tools/attn_bench.py --B 64 --parts 16 --reps 2
# The library under test is chosen by MGX_LIB_PATH in the caller's environment; the program after `--` is python3 itself.
export TMPDIR=/tmp
K=$1; SET=$2; shift 2
ROOT=$(pwd)
D=/tmp/mgx_pmc_$$
rm -rf $D; mkdir -p $D
(cd /tmp && timeout -k 10 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $D -- python3 $ROOT/"$@" > $D/log 2>&1) || { tail -3 $D/log; exit 1; }
python3 - "$D" "$K" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k:28s} {sum(v) / len(v):16.0f}   ({len(v)} launches)")
PY
rm -rf $D
