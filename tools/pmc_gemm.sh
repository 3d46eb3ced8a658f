#!/bin/bash
# On the GPU box: bash tools/pmc_gemm.sh <tag> <shape> [extra env]  — PMC passes over tools/gemm3_check.py --time-only
tag=$1; shape=$2
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r2/pmc_$tag
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/gemm3_check.py --time-only --shapes $shape --reps 5"
rocprofv3 -L 2>/dev/null | grep -o "SQC\?_[A-Z_0-9]*" | sort -u > $OUT/counters.txt
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -o p -- $CMD > /dev/null 2> $OUT/a.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_IFETCH SQ_WAIT_INST_LDS SQ_INSTS_BRANCH --output-format csv -d $OUT/b -o p -- $CMD > /dev/null 2> $OUT/b.err
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/c -o p -- $CMD > /dev/null 2> $OUT/c.err
python - <<PY
import csv, glob, collections
for sub in "abc":
    fs = glob.glob("$OUT/%s/**/*counter_collection.csv" % sub, recursive=True)
    if not fs:
        print(sub, "no csv"); print(open("$OUT/%s.err" % sub).read()[-600:]); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        if "gemm" not in k: continue
        print(k)
        for c, v in d.items(): print("   %-28s mean %.4g  (n=%d)" % (c, sum(v)/len(v), len(v)))
PY
