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
# per-kernel average duration inside the sliced / graph run (compare with the one-stream profile: a kernel that shares the
# chip with another slice's kernel runs longer than alone)
import re
from collections import defaultdict
agg = defaultdict(lambda: [0, 0])
for s, e, n in rows:
    k = re.sub(r"\(.*", "", n).replace("void ", "")[:48]
    agg[k][0] += e - s; agg[k][1] += 1
for k, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:10]:
    print(f"  {k:50s} calls {c:5d} avg {t / c / 1e3:8.1f} us  total {t / 1e6:7.3f} ms")
