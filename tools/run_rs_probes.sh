#!/bin/bash
# GPU box: run every tools/ubench/mlp_rs_probe_<tag> (timings + timeline; the correctness sweep only for `base` and `prio`)
mkdir -p gpurun_out/rs
for b in tools/ubench/mlp_rs_probe_*; do
  tag=${b##*mlp_rs_probe_}
  n=0; case $tag in base*|prio*|opt*|no*|neither) n=9;; esac
  timeout 120 $b $n > gpurun_out/rs/$tag.txt 2>&1
  echo "== $tag rc=$?"; grep -E "differ|M 50432|M 25216" gpurun_out/rs/$tag.txt | grep -v " 0 of" | head -8
  grep -A8 "^unit 1" gpurun_out/rs/$tag.txt | cut -c1-100
done
