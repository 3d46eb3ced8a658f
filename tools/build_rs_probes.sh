#!/bin/bash
# builds the role-split Mlp probe in its ablation / option variants: tools/ubench/mlp_rs_probe_<tag>
cd "$(dirname "$0")/.."
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-pass-failed -DRS_TRACE=1"
build() { tag=$1; shift; hipcc $F "$@" tools/ubench/mlp_rs_probe.hip -o tools/ubench/mlp_rs_probe_$tag 2>&1 | grep -E "error" ; }
for spec in "$@"; do tag=${spec%%:*}; defs=${spec#*:}; build $tag $(echo $defs | tr ',' ' ') & done
wait
ls tools/ubench/mlp_rs_probe_*
