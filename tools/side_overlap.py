"""Two-stream backward (bench.py --side-cus N): what each kernel costs beside the other stream, from two rocprofv3 kernel traces
of the same bench command -- one stream, and with the side stream.
    python3 tools/side_overlap.py <dir of the one-stream trace> <dir of the two-stream trace> [steps=5]
Per kernel: launches per step and mean duration in both runs; for the two-stream run the queue each kernel ran on, the share of
every side-queue kernel's duration during which a main-queue kernel was running too (true overlap in time, not just two queues),
and the step's wall time against the sum of its kernels."""
import collections
import csv
import glob
import sys


def load(d):
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = [{"n": r["Kernel_Name"].split("(")[0].replace("void ", ""), "s": int(r["Start_Timestamp"]), "e": int(r["End_Timestamp"]),
             "q": r.get("Queue_Id", "0")} for r in csv.DictReader(open(f))]
    rows.sort(key=lambda r: r["s"])
    return rows


def steps_of(rows, steps):
    ad = [i for i, r in enumerate(rows) if r["n"].startswith("adam_kernel")][-(steps + 1):]
    return [rows[a + 1:b + 1] for a, b in zip(ad[:-1], ad[1:])], [(rows[a]["e"], rows[b]["e"]) for a, b in zip(ad[:-1], ad[1:])]


def main():
    one, two = load(sys.argv[1]), load(sys.argv[2])
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    s1, w1 = steps_of(one, steps)
    s2, w2 = steps_of(two, steps)
    mainq = collections.Counter(r["q"] for seg in s2 for r in seg).most_common(1)[0][0]
    def table(segs):
        t = collections.defaultdict(list)
        for seg in segs:
            for r in seg:
                t[r["n"]].append((r["e"] - r["s"]) / 1e3)
        return t
    t1, t2 = table(s1), table(s2)
    # overlap of side-queue kernels with main-queue kernels
    ov = collections.defaultdict(lambda: [0.0, 0.0])
    for seg in s2:
        mains = [(r["s"], r["e"]) for r in seg if r["q"] == mainq]
        for r in seg:
            if r["q"] == mainq:
                continue
            o = sum(max(0, min(r["e"], e) - max(r["s"], s)) for s, e in mains)
            ov[r["n"]][0] += o / 1e3
            ov[r["n"]][1] += (r["e"] - r["s"]) / 1e3
    print(f"{'kernel':58s} {'n/step':>6s} {'one stream us':>13s} {'two streams us':>14s} {'ratio':>6s}  queue / overlapped with main-queue kernels")
    tot1 = tot2 = 0.0
    for n in sorted(t1, key=lambda n: -sum(t1[n])):
        a = sum(t1[n]) / len(t1[n])
        per = len(t1[n]) / len(s1)
        b = sum(t2[n]) / len(t2[n]) if n in t2 else float("nan")
        tot1 += sum(t1[n]) / len(s1)
        tot2 += sum(t2.get(n, [])) / len(s2)
        side = f"side, {100 * ov[n][0] / ov[n][1]:.0f} % of its time overlapped" if n in ov and ov[n][1] > 0 else "main"
        if per * a < 20:
            continue
        print(f"{n[:58]:58s} {per:6.1f} {a:13.1f} {b:14.1f} {b / a:6.2f}  {side}")
    wall1 = sum(b - a for a, b in w1) / len(w1) / 1e6
    wall2 = sum(b - a for a, b in w2) / len(w2) / 1e6
    print(f"step wall time: one stream {wall1:.3f} ms (sum of kernels {tot1 / 1e3:.3f}), two streams {wall2:.3f} ms (sum of kernels {tot2 / 1e3:.3f})")


if __name__ == "__main__":
    main()
