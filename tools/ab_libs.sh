#!/bin/bash
# GPU box: interleaved bench of this tree's library against several other builds.  usage: tools/ab_libs.sh <rounds> <lib.so>... [-- bench args]
n=$1; shift
libs=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do libs+=("$1"); shift; done; [ "$1" == "--" ] && shift
run() { python bench.py --no-cpu-baseline --measure-traffic 0 --profile-steps 0 --box-probe 0 --min-seconds 1 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['value'], d['config'].get('streams_per_gpu'), d.get('bit_exact_vs_reference_golden'))"; }
for i in $(seq $n); do
  tag=tree; run "$@"
  for l in "${libs[@]}"; do tag=$(basename $l); IVIT_LIB=$l run "$@"; done
done
