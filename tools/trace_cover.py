"""GPU timeline coverage from a rocprofv3 kernel trace CSV: union of kernel intervals / wall, per replayed step"""
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
# last 60% of the trace = steady-state replays
t0 = rows[int(len(rows) * 0.4)][0]
rows = [r for r in rows if r[0] >= t0]
wall = rows[-1][1] - rows[0][0]
busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
for s, e, _ in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(e - s for s, e, _ in rows)
print(f"kernels {len(rows)} wall {wall/1e6:.3f} ms  union-busy {busy/1e6:.3f} ms ({busy/wall:.3f})  sum of durations {tot/1e6:.3f} ms (avg concurrency {tot/busy:.2f})")
