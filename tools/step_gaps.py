"""GPU idle time inside a training step, from a rocprofv3 kernel trace:
    (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-decode --no-cfg4)
    python3 tools/step_gaps.py /tmp/kt [steps=5]
Takes the last `steps` adam_kernel launches as step boundaries and prints, per step, the wall time between boundaries, the sum of
kernel durations and the idle time (gaps between consecutive kernels), with the largest gaps and the kernels they follow."""
import csv, glob, sys
d = sys.argv[1]; steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(({"n": r["Kernel_Name"], "s": int(r["Start_Timestamp"]), "e": int(r["End_Timestamp"])} for r in csv.DictReader(open(f))), key=lambda r: r["s"])
ad = [i for i, r in enumerate(rows) if r["n"].startswith("adam_kernel")]
ad = ad[-(steps + 1):]
for a, b in zip(ad[:-1], ad[1:]):
    seg = rows[a + 1:b + 1]
    wall = seg[-1]["e"] - rows[a]["e"]
    busy = sum(r["e"] - r["s"] for r in seg)
    gaps = sorted(((seg[i]["s"] - max(rows[a]["e"] if i == 0 else seg[i - 1]["e"], 0), seg[i - 1]["n"] if i else "adam", seg[i]["n"]) for i in range(len(seg))), reverse=True)
    idle = sum(max(g[0], 0) for g in gaps)
    print(f"step: wall {wall/1e6:7.3f} ms  kernels {busy/1e6:7.3f} ms  idle {idle/1e6:6.3f} ms in {len(seg)} launches; largest gaps (us): " +
          ", ".join(f"{g[0]/1e3:.1f} after {g[1][:28]}" for g in gaps[:4]))
