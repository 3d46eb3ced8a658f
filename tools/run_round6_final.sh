#!/bin/bash
# Round 6 final artefacts (GPU box, ONE box for all of it): profile of the DeiT-S headline (kernel-trace stats + PMC passes +
# sliced-graph trace), the default bench line, the four launch modes interleaved, the other single-GPU configurations, Swin-T
# kernel stats.  usage: bash tools/run_round6_final.sh
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6f; mkdir -p $O; cd $R
bash tools/prof.sh r06f > $O/prof.log 2>&1; tail -5 $O/prof.log
cp gpurun_out/prof_r06f/pmc_traffic.json $O/pmc_traffic.json 2>/dev/null
python bench.py > $O/bench_deit_small.json 2> $O/bench_deit_small.err; tail -1 $O/bench_deit_small.json | cut -c1-200
bash tools/mode_sweep.sh 2 > $O/mode_sweep.txt 2>&1; cat $O/mode_sweep.txt
for m in deit_tiny deit_base swin_tiny vit_base_384; do python bench.py --model $m --no-cpu-baseline --measure-traffic 0 > $O/bench_$m.json 2>/dev/null; tail -1 $O/bench_$m.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$m', d['ms_per_step'], d['value'], d['roofline']['frac'], d['config']['launch_mode_trials'])"; done
python tools/attn_probe.py > $O/attn_probe.txt 2>&1; grep -v amdgpu $O/attn_probe.txt | cut -c1-150
python tools/qkv_bench.py > $O/qkv_bench.txt 2>&1; grep -v amdgpu $O/qkv_bench.txt
python tools/gemm_ws_probe.py > $O/gemm_ws_probe.txt 2>&1; grep -v amdgpu $O/gemm_ws_probe.txt
tools/ubench/mfma_valu_split > $O/mfma_valu_split.txt 2>&1; cat $O/mfma_valu_split.txt
tools/ubench/valu_rates > $O/valu_rates.txt 2>&1
for m in swin_tiny deit_base vit_base_384; do bash tools/prof_model.sh r06f_$m --model $m > $O/prof_$m.txt 2>&1; head -12 $O/prof_$m.txt | cut -c1-150; done
