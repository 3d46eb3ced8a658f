#!/bin/bash
# Round 5: the role-split fused Mlp in the library — GPU parity suite, then A/B of the default bench against the same
# sources built with -DIVIT_OPT_MLP_RS=0 (build/ab/libivit_hip_nors.so), interleaved, three rounds.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5b; mkdir -p $O; cd $R
python -m pytest tests -x -q -m gpu > $O/gputests.txt 2>&1; tail -3 $O/gputests.txt
for i in 1 2 3; do
  python bench.py --no-cpu-baseline > $O/bench_rs_$i.json 2>$O/bench_rs_$i.err; tail -1 $O/bench_rs_$i.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rs  ', d['ms_per_step'], d['value'])"
  IVIT_LIB=$R/build/ab/libivit_hip_nors.so python bench.py --no-cpu-baseline > $O/bench_nors_$i.json 2>$O/bench_nors_$i.err; tail -1 $O/bench_nors_$i.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('nors', d['ms_per_step'], d['value'])"
done
python bench.py --no-cpu-baseline --streams 1 --graph 0 > $O/bench_rs_1s.json 2>/dev/null; tail -1 $O/bench_rs_1s.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rs 1stream  ', d['ms_per_step'], d['value'])"
IVIT_LIB=$R/build/ab/libivit_hip_nors.so python bench.py --no-cpu-baseline --streams 1 --graph 0 > $O/bench_nors_1s.json 2>/dev/null; tail -1 $O/bench_nors_1s.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('nors 1stream', d['ms_per_step'], d['value'])"
python bench.py --model swin_tiny --no-cpu-baseline > $O/bench_swin_rs.json 2>/dev/null; tail -1 $O/bench_swin_rs.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('swin rs  ', d['ms_per_step'], d['value'])"
IVIT_LIB=$R/build/ab/libivit_hip_nors.so python bench.py --model swin_tiny --no-cpu-baseline > $O/bench_swin_nors.json 2>/dev/null; tail -1 $O/bench_swin_nors.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('swin nors', d['ms_per_step'], d['value'])"
