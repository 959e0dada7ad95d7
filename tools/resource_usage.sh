#!/bin/bash
# Registers / scratch / occupancy of every kernel in the product library (runs here: hipcc cross-compiles), one line per kernel:
#   bash tools/resource_usage.sh > profiles/rNN_kernel_resource_usage.txt
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Imusicgeneration_amd/csrc"
echo "# hipcc $FLAGS -Rpass-analysis=kernel-resource-usage -c <file> (the flags of musicgeneration_amd/_build.py)"
for f in musicgeneration_amd/csrc/*.hip; do
  echo "## $(basename $f)"
  /opt/rocm/bin/hipcc $FLAGS -Rpass-analysis=kernel-resource-usage -c $f -o /tmp/ru_$$.o 2>&1 | python3 -c '
import sys, re
cur = {}
for line in sys.stdin:
    m = re.search(r"remark: +(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)", line)
    if not m: continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = {"Name": v}
    else:
        cur[k] = v
        if k.startswith("LDS Size"):
            print("\t".join(f"{a}: {b}" for a, b in cur.items()))
'
done
rm -f /tmp/ru_$$.o
