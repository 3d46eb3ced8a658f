#!/bin/bash
# GPU box: the four launch modes of the DeiT-S headline (slices x hipGraph) interleaved on one box.  usage: tools/mode_sweep.sh [rounds] [bench args]
n=${1:-2}; shift
for i in $(seq $n); do
  for m in "2 1" "2 0" "1 1" "1 0" "3 0"; do
    set -- $m
    python bench.py --no-cpu-baseline --measure-traffic 0 --profile-steps 0 --min-seconds 1 --streams $1 --graph $2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('streams $1 graph $2:', d['ms_per_step'], d['value'])"
  done
done
