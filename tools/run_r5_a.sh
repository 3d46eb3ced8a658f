#!/bin/bash
# Round 5, first GPU call: baseline bench of the round-4 library, kernel-trace artefacts of the three other single-GPU
# configurations, the L2 footprint sweep, and the round-2 library under the round-4 stress tools (build/r2tree).
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5a
mkdir -p $O
cd $R
python bench.py --no-cpu-baseline > $O/bench_deit_small.json 2> $O/bench_deit_small.err; tail -1 $O/bench_deit_small.json | cut -c1-160
./tools/ubench/l2_sweep > $O/l2_sweep.txt 2>&1; cat $O/l2_sweep.txt
for m in swin_tiny deit_base vit_base_384; do
  bash tools/prof_model.sh r05_$m --model $m > $O/prof_$m.txt 2>&1; tail -20 $O/prof_$m.txt
done
# round-2 library (commit a3a0367, packed fp32 ON, no layernorm_reg_kernel) under the stress tools of rounds 3/4
cd $R/build/r2tree
(SWIN_ONLY=1 timeout 600 python tools/op_stress.py 40 8 > $O/r2_op_stress.txt 2>&1); tail -40 $O/r2_op_stress.txt
(timeout 600 python tools/swin_stress.py 60 1,2,4,8 > $O/r2_swin_stress.txt 2>&1); cat $O/r2_swin_stress.txt
(STRESS_OPS=1 timeout 300 python tools/swin_stress.py 20 4,8 > $O/r2_swin_stress_ops.txt 2>&1); cat $O/r2_swin_stress_ops.txt
