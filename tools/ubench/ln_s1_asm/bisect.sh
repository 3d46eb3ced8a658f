#!/bin/bash
# Assembly-level bisection of the S = 1 LayerNorm's packed-fp32 sensitivity (CPU part: builds the code objects; the GPU part is
# tools/ubench/ln_s1_asm/run.sh).  Every code object is the compiler's own assembly of layernorm_reg_kernel<192, 1> with some
# classes of v_pk_*_f32 rewritten into two scalar instructions by unpack_pk.py — nothing else changes, not even the schedule.
set -e
cd "$(dirname "$0")"
LL=/opt/rocm/lib/llvm/bin
OUT=../../../build/ln_s1_asm
mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-pass-failed -Wno-unused-command-line-argument -DIVIT_PROBE_LN192_S1=1 --cuda-device-only -S victim_only.hip -o $OUT/victim.s
K=layernorm_reg_kernelILi192ELi1
build() {   # name classes [range]
    python3 unpack_pk.py $OUT/victim.s $OUT/$1.s $K "$2" $3 | sed "s/^/$1: /"
    $LL/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $OUT/$1.s -o $OUT/$1.o
    $LL/ld.lld -shared $OUT/$1.o -o $OUT/$1.co
    rm -f $OUT/$1.o
}
build none none
build all all
for c in add addmod mul muls fmas fma; do build only_$c $c; done
build keep_add addmod,mul,muls,fmas,fma
build keep_addmod add,mul,muls,fmas,fma
build keep_mul add,addmod,muls,fmas,fma
build keep_muls add,addmod,mul,fmas,fma
build keep_fmas add,addmod,mul,muls,fma
build keep_fma add,addmod,mul,muls,fmas
build keep_addneg add,addsel,mul,muls,fmas,fma
build keep_addsel add,addneg,mul,muls,fmas,fma
for extra in "$@"; do build $(echo "$extra" | tr ':,' '__') ${extra%%@*} ${extra#*@}; done
ls $OUT/*.co | wc -l
