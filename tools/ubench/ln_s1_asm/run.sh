#!/bin/bash
# GPU part of the bisection: every code object of build/ln_s1_asm beside the MFMA-only aggressor (nt_like<25>) of ln_s1_standalone
mkdir -p gpurun_out/haz
for co in ${@:-none all only_add only_addmod only_mul only_muls only_fmas only_fma keep_add keep_addmod keep_mul keep_muls keep_fmas keep_fma}; do
    echo "== $co"
    VICTIM_CO=build/ln_s1_asm/$co.co timeout 120 tools/ubench/ln_s1_standalone_pk ${ROUNDS:-20} 8 ${AGGR:-225} 2>&1 | grep -v "^  round\|^victim from"
done
