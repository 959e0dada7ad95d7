#!/bin/bash
# HBM traffic per kernel launch from the PMC counters (runs on the GPU box, from the repo root).
# Two separate passes (FETCH_SIZE and WRITE_SIZE do not fit one pass), kernel-trace only.
# bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024   -- gfx950 tallies 128-B read requests at 64 B (MI355X_MICROARCH.md, HBM)
export TMPDIR=/tmp
OUT=${1:-gpurun_out/traffic.json}
export BATCH=${BATCH:-64}
ROOT=$(pwd)
export TMPD=/tmp/mgx_traffic_$$; mkdir -p $TMPD
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $TMPD/pmc_$c
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $TMPD/pmc_$c -- \
      python3 $ROOT/bench.py --batch $BATCH --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-decode --no-cfg4 > $TMPD/pmc_$c.log 2>&1 || { tail -5 $TMPD/pmc_$c.log; exit 1; }
done
cd $ROOT
python3 - "$OUT" <<'PY'
import csv, glob, json, os, sys, collections
acc = {c: collections.defaultdict(list) for c in ("FETCH_SIZE", "WRITE_SIZE")}
for c in acc:
    for f in glob.glob(os.environ["TMPD"] + f"/pmc_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                acc[c][r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
out = {}
for k in sorted(set(acc["FETCH_SIZE"]) | set(acc["WRITE_SIZE"])):
    f = acc["FETCH_SIZE"].get(k, [0.0]); w = acc["WRITE_SIZE"].get(k, [0.0])
    fe, wr = sum(f) / len(f), sum(w) / len(w)
    out[k] = {"launches": len(f), "fetch_kb_raw": fe, "write_kb": wr, "hbm_bytes_per_launch": (2 * fe + wr) * 1024}
json.dump({"command": f"bench.py --batch {os.environ['BATCH']} --steps 2 --warmup 1 (cfg2)", "formula": "(2*FETCH_SIZE + WRITE_SIZE)*1024",
           "kernels": out}, open(sys.argv[1], "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:25]:
    print(f"{k[:60]:60s} n={v['launches']:4d} {v['hbm_bytes_per_launch']/1e6:10.2f} MB/launch")
PY
