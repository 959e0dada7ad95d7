#!/bin/bash
# usage: tools/dw4_clock.sh "<variant names>"   (GPU box, repo root): kernel duration and GRBM cycles of linear_dw_ring4_kernel per library variant -> shader clock
export TMPDIR=/tmp
ROOT=$(pwd)
for v in $1; do
  rm -rf /tmp/dwclk_$v
  (cd /tmp && MGX_LIB_PATH=$ROOT/musicgeneration_amd/libmgx_$v.so MGX_DW_RING4=1 timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d /tmp/dwclk_$v -- python3 $ROOT/tools/check_dw4.py 131072 > /tmp/dwclk_$v.log 2>&1) || { tail -3 /tmp/dwclk_$v.log; exit 1; }
  python3 - $v <<'PY'
import csv, glob, sys, collections
v = sys.argv[1]
cnt = collections.defaultdict(list)
for f in glob.glob(f"/tmp/dwclk_{v}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "dw_ring4" in r["Kernel_Name"] and int(r["Grid_Size"]) == 240 * 256:
            cnt[r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = []
for f in glob.glob(f"/tmp/dwclk_{v}/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "dw_ring4" in r["Kernel_Name"] and int(r.get("Grid_Size") or r.get("Grid_Size_X") or 0) == 240 * 256:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
m = {k: sum(x) / len(x) for k, x in cnt.items()}
cyc = m["GRBM_GUI_ACTIVE"] / 8
d = sum(dur) / len(dur)
wc = max(m["SQ_WAVE_CYCLES"], 1)
print(f"{v:14s} {d:7.1f} us  {cyc:9.0f} cycles  {cyc / d / 1e3:5.2f} GHz  MFMA busy {100 * m['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / cyc:5.1f} %  "
      f"active {100 * m['SQ_ACTIVE_INST_ANY'] / wc:4.0f} %  inst-stall {100 * m['SQ_WAIT_INST_ANY'] / wc:4.0f} %  wait {100 * m['SQ_WAIT_ANY'] / wc:4.0f} %  "
      f"LDS active {100 * m['SQ_LDS_IDX_ACTIVE'] / 256 / cyc:5.1f} %  bank conflicts {100 * m['SQ_LDS_BANK_CONFLICT'] / 256 / cyc:5.1f} %")
PY
done
