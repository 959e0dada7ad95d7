import csv, sys, signal
signal.signal(signal.SIGPIPE, signal.SIG_DFL)      # `| head` closes the pipe: end quietly
rows = list(csv.DictReader(open(sys.argv[1])))
nstep = int(sys.argv[2]) if len(sys.argv) > 2 else 7
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total GPU ms/step", round(tot / nstep / 1e6, 3))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 24]:
    print("%-58s calls/step %5.1f avg %8.1f us %5.1f%% per-step %6.3f ms" % (
        r["Name"][:58], int(r["Calls"]) / nstep, float(r["AverageNs"]) / 1e3, float(r["Percentage"]),
        float(r["TotalDurationNs"]) / nstep / 1e6))
