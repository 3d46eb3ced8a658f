// The victim of ln_s1_standalone.hip alone, for the assembly-level bisection (tools/ubench/ln_s1_asm/bisect.sh): compiled to
// gfx950 assembly, classes of v_pk_*_f32 are rewritten into their two scalar instructions by unpack_pk.py, the result is assembled
// into a code object and ln_s1_standalone loads it (VICTIM_CO=...) in place of its compiled-in victim.
#include "../../../i-vit_amd/csrc/ivit_device.h"
#include "../../../i-vit_amd/csrc/ivit_elementwise.h"
#include "../../../i-vit_amd/csrc/ivit_layernorm.h"
template __global__ void layernorm_reg_kernel<192, 1>(const int16_t *, long long, long long, float, const float *, const float *, const ivit_dyadic *, int8_t *);
