#!/bin/bash
# Round 5 final artefacts (GPU box): default bench + the other single-GPU configurations + one-stream line, with the final library
set -x
O=gpurun_out/r5f; mkdir -p $O
python bench.py > $O/bench_deit_small.json 2> $O/bench_deit_small.err; tail -1 $O/bench_deit_small.json | cut -c1-200
for m in deit_tiny deit_base swin_tiny vit_base_384; do python bench.py --model $m --no-cpu-baseline > $O/bench_$m.json 2>/dev/null; tail -1 $O/bench_$m.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$m', d['ms_per_step'], d['value'], d['roofline']['frac'])"; done
python bench.py --no-cpu-baseline --streams 1 --graph 0 > $O/bench_deit_small_1stream.json 2>/dev/null; tail -1 $O/bench_deit_small_1stream.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('1stream', d['ms_per_step'], d['value'])"
