import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
n = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:18]:
    print("%-62s calls %5d  per-fwd ms %7.3f  avg_us %8.1f  pct %5.1f" % (r["Name"][:62], int(r["Calls"]), float(r["TotalDurationNs"]) / n / 1e6, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
print("total per fwd ms %.3f" % (tot / n / 1e6))
