#!/bin/bash
# GPU box: the tree's library against several other builds, per model.  usage: tools/ab_multi.sh "<model> [bench args]" lib1.so lib2.so ...
margs=$1; shift
for i in 1 2; do
  python bench.py --no-cpu-baseline --measure-traffic 0 --model $margs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('tree', '$margs', d['ms_per_step'], d['value'])"
  for l in "$@"; do IVIT_LIB=$l python bench.py --no-cpu-baseline --measure-traffic 0 --model $margs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$l', '$margs', d['ms_per_step'], d['value'])"; done
done
