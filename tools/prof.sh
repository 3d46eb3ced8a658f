#!/bin/bash
# On the GPU box (via gpurun): bash tools/prof.sh <tag>
# kernel-trace/stats pass + separate PMC passes (never combined with other trace domains).
tag=$1
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$tag
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# one stream, no graph, whole batch per launch: the launches bench.py's roofline block times
CMD="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 --streams 1 --graph 0 --reps 1 --min-seconds 0 --box-probe 0 --measure-traffic 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $CMD > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -o p -- $CMD > /dev/null 2> $OUT/pmc_sq.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_I8 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq2 -o p -- $CMD > /dev/null 2> $OUT/pmc_sq2.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- $CMD > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o p -- $CMD > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_tcc -o p -- $CMD > /dev/null 2> $OUT/pmc_tcc.err
find $OUT -name "*.csv" | head -30
python $R/tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
# the default (2 slices, hipGraph) run: timeline coverage and per-kernel durations when slices share the chip
rocprofv3 --kernel-trace --output-format csv -d $OUT/graph_trace -o g -- python $R/bench.py --steps 12 --warmup 2 --no-cpu-baseline --profile-steps 0 --reps 1 --min-seconds 0 --streams 2 --graph 1 --box-probe 0 --measure-traffic 0 > $OUT/bench_graph_trace.json 2> $OUT/graph_trace.err
python $R/tools/trace_cover.py $OUT/graph_trace > $OUT/sliced_graph_trace.txt 2>&1
cat $OUT/sliced_graph_trace.txt
# keep only small artefacts
find $OUT -name "*kernel_trace.csv" -size +3M -delete
