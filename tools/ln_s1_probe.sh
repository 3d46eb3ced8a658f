#!/bin/bash
# Round-4 probe of the S = 1 register LayerNorm (see profiles/README.md): the mixed-operator stress with C = 192 LayerNorm as
# the victim, once per probe library built with -DIVIT_PROBE_LN192_S1=1 [-DLNR_S1_VARIANT=n] into build/lnprobe/.
REP=${1:-40}
for v in 0 1 2 3 4; do
  lib=build/lnprobe/libivit_s1v$v.so
  [ -f $lib ] || continue
  echo "=== variant $v"
  IVIT_LIB=$PWD/$lib MIXED_ONLY=1 SWIN_ONLY=1 MIX_FILTERS="${FILTERS:-ivit_linear_i8_requant;ivit_linear_i8_requant_residual;attention;mlp}" timeout 900 python tools/op_stress.py $REP 8 2>&1 | grep -v "^$" | tail -8
done
