#!/bin/bash
# On the GPU box (via gpurun): bash tools/prof_model.sh <tag> <bench.py model args...> — kernel-trace stats of a single-stream, no-graph run
tag=$1; shift
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$tag
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 --streams 1 --graph 0 --reps 1 --min-seconds 0 --box-probe 0 --measure-traffic 0 "$@" > $OUT/bench.json 2> $OUT/trace.err
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:18]:
    print(f"{r['Name'][:90]:90s} calls {int(r['Calls']):5d} total_us {float(r['TotalDurationNs'])/1e3:10.1f} avg_us {float(r['AverageNs'])/1e3:9.2f} pct {float(r['Percentage']):6.2f}")
PY
find $OUT -name "*kernel_trace.csv" -size +3M -delete
