#!/bin/bash
# usage: tools/pmc_attn.sh [batch]   (GPU box, repo root) -- separate rocprofv3 --pmc passes over tools/attn_bench.py,
# averaged per attention kernel; prints MFMA-busy and wave-state fractions per kernel.
export TMPDIR=/tmp
B=${1:-16}
ROOT=$(pwd)
OUT=${2:-gpurun_out/pmc_attn.json}
PARTS=${PARTS:-65}     # attn_bench.py parts: forward + the whole backward
TAG=pmcattn_$$
mkdir -p gpurun_out
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_MISC" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES"; do
  n=$(echo $set | cut -c1-14 | tr " " "_")
  rm -rf /tmp/${TAG}_$n
  (cd /tmp && timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/${TAG}_$n -- python3 $ROOT/tools/attn_bench.py --B $B --reps 1 --parts $PARTS $ATTN_ARGS > $ROOT/gpurun_out/pmc_attn_pass_$n.log 2>&1) \
    || { echo "pmc pass $n failed (rc $?):"; tail -5 gpurun_out/pmc_attn_pass_$n.log; exit 1; }
done
python3 - "$OUT" "$B" "$TAG" "$PARTS" <<'PY'
import csv, glob, json, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
files = glob.glob(f"/tmp/{sys.argv[3]}_*/**/*counter_collection.csv", recursive=True)
assert len(files) == 3, f"expected 3 counter files of this run, found {files}"
for f in files:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "rel_attn" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, c in acc.items():
    v = {n: sum(x) / len(x) for n, x in c.items()}
    need = ("GRBM_GUI_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_LDS_IDX_ACTIVE")
    missing = [n for n in need if n not in v]
    assert not missing, f"{k}: counters missing from the passes: {missing}"
    cyc = v["GRBM_GUI_ACTIVE"] / 8
    v["kernel_cycles"] = cyc
    v["mfma_busy_frac"] = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / cyc          # 1024 SIMDs
    v["waves_per_simd"] = v.get("SQ_WAVE_CYCLES", 0) * 4 / 1024 / cyc                # SQ_WAVE_CYCLES counts quad-cycles
    for n in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY"):
        v[n + "_frac"] = v.get(n, 0) / max(v.get("SQ_WAVE_CYCLES", 1), 1)
    v["lds_active_frac"] = v.get("SQ_LDS_IDX_ACTIVE", 0) / 256 / cyc                  # per CU
    out[k] = v
    print(f"{k[:44]:44s} cycles {cyc:9.0f}  MFMA busy {100*v['mfma_busy_frac']:5.1f} %  waves/SIMD {v['waves_per_simd']:.2f}  "
          f"active {100*v['SQ_ACTIVE_INST_ANY_frac']:.0f} % inst-stall {100*v['SQ_WAIT_INST_ANY_frac']:.0f} % wait {100*v['SQ_WAIT_ANY_frac']:.0f} %  "
          f"VALU {v.get('SQ_INSTS_VALU',0)/1e6:.1f} M LDS {v.get('SQ_INSTS_LDS',0)/1e6:.1f} M MFMA {v.get('SQ_INSTS_MFMA',0)/1e6:.2f} M")
json.dump({"command": f"tools/pmc_attn.sh {sys.argv[2]} (attn_bench.py --B {sys.argv[2]} --reps 1 --parts {sys.argv[4]}, three --pmc passes)", "kernels": out},
          open(sys.argv[1], "w"), indent=1)
PY
