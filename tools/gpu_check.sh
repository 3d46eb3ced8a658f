#!/bin/bash
# usage (on the GPU box via gpurun): bash tools/gpu_check.sh <tag> [pytest -k expr]
tag=$1
python -m pytest tests -m gpu -x -q ${2:+-k "$2"} 2>&1 | tail -12 > gpurun_out/test_$tag.log
cat gpurun_out/test_$tag.log
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_$tag.json 2>gpurun_out/bench_$tag.err || tail -5 gpurun_out/bench_$tag.err
python - <<PY
import json
d = json.load(open("gpurun_out/bench_$tag.json"))
print("img/s", d["value"], "ms", d["ms_per_step"], "exact", d["bit_exact_vs_reference_golden"], "gemm TOPS", d["roofline"]["achieved"], "model frac", d["model_roofline_frac"])
for k, v in d["kernel_breakdown_ms"].items(): print(f"  {k:36s} {v['ms_per_step']:8.4f} ms  x{v['launches']}")
PY
