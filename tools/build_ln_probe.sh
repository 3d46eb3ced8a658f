#!/bin/bash
# Builds the probe libraries of the S = 1 register LayerNorm study (profiles/README.md round 4) into build/lnprobe/:
#   v0 as emitted (packed fp32), v1 s_nop 7 around the DPP groups, v2 ds_bpermute instead of DPP, v3 constants from global memory,
#   v4 built without packed fp32.  Then: gpurun -- tools/ln_s1_probe.sh / tools/ln_s1_stress_r04.sh
set -e
cd "$(dirname "$0")/../i-vit_amd/csrc"
mkdir -p ../../build/lnprobe
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-pass-failed -fPIC -shared -DIVIT_PROBE_LN192_S1=1"
for v in 0 1 2 3; do /opt/rocm/bin/hipcc $F -DLNR_S1_VARIANT=$v ivit_hip.hip -o ../../build/lnprobe/libivit_s1v$v.so & done
/opt/rocm/bin/hipcc $F -Xclang -target-feature -Xclang -packed-fp32-ops ivit_hip.hip -o ../../build/lnprobe/libivit_s1v4.so &
wait
ls -la ../../build/lnprobe
