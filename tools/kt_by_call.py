"""Per call-site durations of the GEMM kernels inside the training step, from a rocprofv3 --kernel-trace CSV: the launches of one step
come in a fixed order, so the k-th launch of a kernel within a step is always the same call site.   python tools/kt_by_call.py trace.csv [steps]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 7
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
by = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"].split("(")[0]
    if "linear_" in n or "dw_fixup" in n:
        by[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, d in sorted(by.items()):
    per = len(d) // steps
    if per == 0:
        continue
    d = d[len(d) - per * (steps - 2):]           # the last steps - 2 steps (warm)
    k = len(d) // per
    site = [sum(d[i + per * j] for j in range(k)) / k for i in range(per)]
    print(f"{n[:60]:60s} {per:3d} calls/step, sum {sum(site) / 1e3:6.3f} ms:", " ".join(f"{x:.0f}" for x in site))
