#!/bin/bash
# Round 5 final artefacts (GPU box): profile of the default configuration, default bench + the other single-GPU configurations +
# one-stream line, kernel stats of Swin-T, all with the final library
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5f; mkdir -p $O; cd $R
bash tools/prof.sh r05f > $O/prof.log 2>&1; tail -5 $O/prof.log
cp gpurun_out/prof_r05f/pmc_traffic.json profiles/pmc_traffic.json 2>/dev/null    # bench.py's roofline.traffic reads the committed copy
python bench.py > $O/bench_deit_small.json 2> $O/bench_deit_small.err; tail -1 $O/bench_deit_small.json | cut -c1-200
for m in deit_tiny deit_base swin_tiny vit_base_384; do python bench.py --model $m --no-cpu-baseline > $O/bench_$m.json 2>/dev/null; tail -1 $O/bench_$m.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$m', d['ms_per_step'], d['value'], d['roofline']['frac'])"; done
python bench.py --no-cpu-baseline --streams 1 --graph 0 > $O/bench_deit_small_1stream.json 2>/dev/null; tail -1 $O/bench_deit_small_1stream.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('1stream', d['ms_per_step'], d['value'])"
bash tools/prof_model.sh r05f_swin_tiny --model swin_tiny > $O/prof_swin_tiny.txt 2>&1; head -14 $O/prof_swin_tiny.txt
cp gpurun_out/prof_r05f/pmc_traffic.json $O/pmc_traffic.json
