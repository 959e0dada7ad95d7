"""Per-kernel statistics from a rocprofv3 rocpd (SQLite) result file -- for runs made without --output-format csv.
usage: python tools/rocpd_stats.py results.db [--seq]     (--seq: dispatches in launch order instead of the summary)"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
names = {r[0]: r[1] for r in db.execute(f"select id, kernel_name from {ks}")}
rows = list(db.execute(f"select kernel_id, start, end from {kd} order by start"))
if "--seq" in sys.argv:
    for k, s, e in rows: print(f"{(e - s) / 1e3:10.1f} us  {names[k][:90]}")
    sys.exit(0)
agg = collections.defaultdict(list)
for k, s, e in rows: agg[names[k]].append(e - s)
tot = sum(sum(v) for v in agg.values())
for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"{n[:80]:80s} {len(v):5d} calls  avg {sum(v) / len(v) / 1e3:9.1f} us  min {min(v) / 1e3:9.1f}  {100 * sum(v) / tot:5.1f}%")
