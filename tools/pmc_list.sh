#!/bin/bash
# usage (GPU box): bash tools/pmc_list.sh <tag> "<counters...>" <command...>  — one rocprofv3 --pmc pass, per-kernel means
tag=$1; ctrs=$2; shift; shift
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmcl_$tag
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $OUT/a -o p -- "$@" > /dev/null 2> $OUT/a.err
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("$OUT/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in acc:
    print(k)
    for c in sorted(acc[k]): print(f"   {c:28s} {acc[k][c]/cnt[k][c]:16.1f} per dispatch  (n={cnt[k][c]})")
PY
tail -3 $OUT/a.err
