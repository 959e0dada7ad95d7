"""How much of a kernel trace runs concurrently: python overlap_from_trace.py <dir with *kernel_trace.csv>
Prints total kernel time, wall span covered by >=1 kernel, and the share of that span with >=2 kernels in flight."""
import csv, glob, sys
fn = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
ev = []
for r in csv.DictReader(open(fn)):
    ev.append((int(r["Start_Timestamp"]), 1)); ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
n = 0; last = ev[0][0]; busy1 = busy2 = 0
for t, d in ev:
    if n >= 1: busy1 += t - last
    if n >= 2: busy2 += t - last
    n += d; last = t
tot = sum(1 for _, d in ev if d == 1)
print(f"{fn}: {tot} kernels, span with >=1 kernel {busy1/1e6:.2f} ms, with >=2 kernels {busy2/1e6:.2f} ms ({100*busy2/max(busy1,1):.1f} %)")
