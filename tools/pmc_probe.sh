#!/bin/bash
# usage (GPU box): bash tools/pmc_probe.sh <tag> <command...>   — SQ / memory counter passes of one command, summarised per kernel
tag=$1; shift
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_$tag
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -o p -- "$@" > /dev/null 2> $OUT/a.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d $OUT/b -o p -- "$@" > /dev/null 2> $OUT/b.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/c -o p -- "$@" > /dev/null 2> $OUT/c.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/d -o p -- "$@" > /dev/null 2> $OUT/d.err
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
