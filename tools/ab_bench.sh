#!/bin/bash
# GPU box: interleaved A/B of the default bench — this tree's library against another build (IVIT_LIB).  usage: tools/ab_bench.sh <other.so> [rounds] [bench args...]
other=$1; n=${2:-3}; shift; shift
for i in $(seq $n); do
  python bench.py --no-cpu-baseline --measure-traffic 0 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('tree ', d['ms_per_step'], d['value'])"
  IVIT_LIB=$other python bench.py --no-cpu-baseline --measure-traffic 0 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('other', d['ms_per_step'], d['value'])"
done
