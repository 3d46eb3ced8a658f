// ivit_device.h — device-side building blocks shared by the gfx950 kernels.
//
// Numerics contract (DESIGN.md §"numerics contract"):
//   * contractions are exact int32 (MFMA i8);
//   * dyadic requantisation is the reference's fp64 sequence
//       rne( (double(z) * m) * 2^-e )            (quant_utils.py:229-230)
//   * Shiftmax / ShiftGELU / I-LayerNorm interiors are binary32 sequences whose
//     individual roundings matter: this translation unit is built with
//     -ffp-contract=off, IEEE division (hipcc default) and no fast-math.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/ivit.h"

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef short v8s __attribute__((ext_vector_type(8)));

#define IVIT_WAVE 64

// ---- dyadic requant -------------------------------------------------------
__device__ __forceinline__ double rq_f64(double z, double m, double r) {
    return __builtin_rint((z * m) * r);
}
// output.type(torch.float) then clamp (quant_utils.py:247-251); lo/hi are small
// powers of two so clamping in double before the cast is equivalent.
__device__ __forceinline__ int clamp_bits(double v, int bits) {
    const double hi = (double)((1ll << (bits - 1)) - 1), lo = -hi - 1.0;
    v = v < lo ? lo : (v > hi ? hi : v);
    return (int)v;
}
template <int BITS>
__device__ __forceinline__ int clamp_b(double v) {
    constexpr double hi = (double)((1ll << (BITS - 1)) - 1), lo = -hi - 1.0;
    v = v < lo ? lo : (v > hi ? hi : v);
    return (int)v;
}

// rne to int32, SATURATING: v_rndne_f64 + v_cvt_i32_f64.  The C++ cast of an out-of-range double is undefined behaviour; the
// requant paths below rely on the instruction's saturation (|z * c| may exceed 2^31 before the clamp), so the instruction is
// named instead of implied.  tests/test_gpu_parity.py::test_linear_requant_saturating_multipliers pins it.
__device__ __forceinline__ int rint_sat_i32(double t) {
    int v;
    asm("v_cvt_i32_f64 %0, %1" : "=v"(v) : "v"(__builtin_rint(t)));
    return v;
}

// lean form of the same requant: c = m*2^-e is exact in fp64 (m integer <= 2^31, power-of-two
// scaling), so fl64(z*c) == fl64(z*m)*2^-e: one v_mul_f64; v_cvt_i32_f64 saturates and the
// clamp is an integer med3.  Bit-identical to rq_f64 + clamp_b.
__device__ __forceinline__ int rq_c(double z, double c, int lo, int hi) {
    int v = rint_sat_i32(z * c);
    return min(max(v, lo), hi);
}

// Fast exact form: t = fma(double(z), c, 1.5*2^52) rounds z*c to the nearest integer
// (ties-to-even) inside the FMA and leaves it, two's complement, in the low dword of t.
// Identical to rq_c whenever (1) z*m is exact in fp64, i.e. |z| < 2^22 (m <= 2^31), so the
// reference's first rounding fl64(z*m) is a no-op, and (2) |z*c| < 2^31 (|c| < 2^9 given (1)),
// so the low dword does not wrap where v_cvt_i32_f64 would saturate.  Callers check both.
#define RQ_FAST_ZLIM (1 << 22)
#define RQ_FAST_CLIM 512.0
__device__ __forceinline__ int rq_fast(int z, double c) {
    return __double2loint(__builtin_fma((double)z, c, 6755399441055744.0));
}

// ---- correctly-rounded fp32 division by a loop-invariant divisor --------------
// Measured on MI355X (tools/ubench/valu_rates.hip): `a / b` (v_div_scale, v_rcp, 5 fma,
// v_div_fmas, v_div_fixup) costs ~16x a v_mul_f32.  Every divisor on this path is a
// per-layer or per-channel constant, so the reciprocal refinement is hoisted and the
// quotient is produced by the same FMA tail as the compiler's IEEE expansion
// (q1 = fma(fma(-d,q0,n), y, q0); q = fma(fma(-d,q1,n), y, q1)) — correctly rounded for
// normal-range operands (no div_scale/div_fixup cases occur here: |n|,|d| in ~[1e-12,1e12]).
// Pinned bit-for-bit against `/` on the GPU in tests/test_gpu_parity.py::test_lean_division.
struct RcpC { float d, y; };
__device__ __forceinline__ RcpC rcp_prepare(float d) {
    float r = __builtin_amdgcn_rcpf(d);
    float e = __builtin_fmaf(-d, r, 1.0f);
    RcpC c;
    c.d = d;
    c.y = __builtin_fmaf(e, r, r);
    return c;
}
__device__ __forceinline__ float lean_div(float n, RcpC c) {
    float q0 = n * c.y;
    float r0 = __builtin_fmaf(-c.d, q0, n);
    float q1 = __builtin_fmaf(r0, c.y, q0);
    float r1 = __builtin_fmaf(-c.d, q1, n);
    return __builtin_fmaf(r1, c.y, q1);
}
// fl(fl(Q*s)/s) with the hoisted reciprocal
__device__ __forceinline__ float requotient_c(float q, RcpC s) { return lean_div(q * s.d, s); }
// shift_exp with hoisted reciprocal of x0
__device__ __forceinline__ float shift_exp_c(float x, RcpC x0, float nx0, int n) {
    float t = x + floorf(x * 0.5f);
    t = t - floorf(x * 0.0625f);
    t = fmaxf(t, nx0);
    float q = floorf(lean_div(t, x0));
    float r = t - x0.d * q;
    float e = r * 0.5f - x0.d;
    e = floorf(ldexpf(e, n - (int)q));
    return fmaxf(e, 0.0f);
}

// shift_exp on a precomputed x (same value sequence as shift_exp_c; the two products
// x0*q and r*0.5 are exact, so fma(-x0,q,t) == fl(t - fl(x0*q)) and fma(r,0.5,-x0) ==
// fl(fl(r/2) - x0) bit for bit)
__device__ __forceinline__ float shift_exp_f(float x, RcpC x0, float nx0, int n) {
    float t = x + floorf(x * 0.5f);
    t = t - floorf(x * 0.0625f);
    t = fmaxf(t, nx0);
    float q = floorf(lean_div(t, x0));
    float r = __builtin_fmaf(-x0.d, q, t);
    float e = __builtin_fmaf(r, 0.5f, -x0.d);
    e = floorf(ldexpf(e, n - (int)q));
    return fmaxf(e, 0.0f);
}

// Shiftmax only (x <= 0): the reference's final clamp(min=0) can never bind — t <= 0 and x0 <= -1 give
// q >= 0, r = t - x0*q in (x0, 0] up to an ulp, so e = r/2 - x0 >= -x0/2 - ulp > 0 and floor(e * 2^(n-q)) >= 0.
__device__ __forceinline__ float shift_exp_nonpos(float x, RcpC x0, float nx0, int n) {
    float t = x + floorf(x * 0.5f);
    t = t - floorf(x * 0.0625f);
    t = fmaxf(t, nx0);
    float q = floorf(lean_div(t, x0));
    float r = __builtin_fmaf(-x0.d, q, t);
    float e = __builtin_fmaf(r, 0.5f, -x0.d);
    return floorf(ldexpf(e, n - (int)q));
}

// ---- fp32-faithful pieces ---------------------------------------------------
// value a consumer of (Q, s) sees: fl(fl(Q*s)/s)   (quant_modules.py:204-206 then
// :94/:359/:426/:484).  Not always equal to Q.
__device__ __forceinline__ float requotient(float q, float s) {
    float X = q * s;
    return X / s;
}

// int_exp_shift of Shiftmax/ShiftGELU (quant_modules.py:410-423, 469-481).
// x: fp32 "integer" (<= 0 except the exp(-max) call of ShiftGELU);
// x0 = floor(-1/s); nx0 = fl(n*x0).
__device__ __forceinline__ float shift_exp(float x, float x0, float nx0, int n) {
    float t = x + floorf(x * 0.5f);       // x/2: exact scaling
    t = t - floorf(x * 0.0625f);          // x/2**4
    t = fmaxf(t, nx0);
    float q = floorf(t / x0);
    float r = t - x0 * q;
    float e = r * 0.5f - x0;
    e = floorf(ldexpf(e, n - (int)q));    // e * 2**(n-q): exact scaling (or inf)
    return fmaxf(e, 0.0f);
}

// factor = floor((2**31-1)/S): python-int / tensor = reciprocal * fl32(2^31-1)=2^31
__device__ __forceinline__ float recip_factor(float S) {
    S = fminf(S, 2147483648.0f);          // clamp_max_(2**31-1) in fp32
    return floorf((1.0f / S) * 2147483648.0f);
}

// ---- torch-CPU summation order over a contiguous fp32 row -------------------
// ATen SumKernel (vectorized_inner_sum -> row_sum -> multi_row_sum): 8-lane
// vectors, 4 interleaved accumulators, 4-level cascade with 16-step levels.
// 32 consecutive GPU lanes own the 32 (ilp k, vector lane l) accumulators:
// sub = k*8 + l  reads element 32*i + sub.  `elem(idx)` returns the row's value.
// Every one of the 32 lanes returns the full sum.  `sub` = lane index in [0,32).
template <typename F>
__device__ __forceinline__ float torch_order_sum32(int n, int sub, F elem) {
    const int nvec = n >> 3;
    const int size = nvec >> 2;           // steps of 32 elements
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int i = 0;
    // level_power = max(4, CeilLog2(size)/4) = 4 for every size < 2^20
    for (; i + 16 <= size;) {
        for (int j = 0; j < 16; ++j, ++i) a0 += elem(32 * i + sub);
        a1 += a0; a0 = 0.f;
        if ((i & 0xF0) != 0) continue;
        a2 += a1; a1 = 0.f;
        if ((i & 0xF00) != 0) continue;
        a3 += a2; a2 = 0.f;
    }
    for (; i < size; ++i) a0 += elem(32 * i + sub);
    a0 += a1; a0 += a2; a0 += a3;
    // leftover whole vectors go to accumulator k = 0
    for (int v = size * 4; v < nvec; ++v) {
        float x = (sub < 8) ? elem(v * 8 + sub) : 0.f;
        a0 += x;                            // lanes >= 8 add 0 to an unused value... see below
    }
    // NOTE: for sub >= 8 the "+= 0.f" above must not perturb a0: x + 0.0f == x for all
    // finite x (and -0.0f + 0.0f = 0.0f compares equal; sums here are never -0).
    // combine the 4 ilp accumulators in order k = 1,2,3 (vector adds)
    float p = a0;
    p += __shfl(a0, (sub & 7) + 8, 32);
    p += __shfl(a0, (sub & 7) + 16, 32);
    p += __shfl(a0, (sub & 7) + 24, 32);
    // p is valid where sub < 8 (lane l's partial); scalar tail first, then lanes 0..7
    float fin = 0.f;
    for (int k = nvec * 8; k < n; ++k) fin += elem(k);
    if (nvec > 0) {
#pragma unroll
        for (int l = 0; l < 8; ++l) fin += __shfl(p, l, 32);
    }
    return fin;
}

// Same summation order with 16 lanes per row (sub in [0,16)): lane `sub` owns the two
// accumulators sub and sub+16.  Every one of the 16 lanes returns the full sum.
template <typename F>
__device__ __forceinline__ float torch_order_sum16(int n, int sub, F elem) {
    const int nvec = n >> 3, size = nvec >> 2;
    float a0[2] = {0.f, 0.f}, a1[2] = {0.f, 0.f}, a2[2] = {0.f, 0.f}, a3[2] = {0.f, 0.f};
    int i = 0;
    for (; i + 16 <= size;) {
        for (int j = 0; j < 16; ++j, ++i) {
            a0[0] += elem(32 * i + sub);
            a0[1] += elem(32 * i + sub + 16);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) { a1[s] += a0[s]; a0[s] = 0.f; }
        if ((i & 0xF0) != 0) continue;
#pragma unroll
        for (int s = 0; s < 2; ++s) { a2[s] += a1[s]; a1[s] = 0.f; }
        if ((i & 0xF00) != 0) continue;
#pragma unroll
        for (int s = 0; s < 2; ++s) { a3[s] += a2[s]; a2[s] = 0.f; }
    }
    for (; i < size; ++i) {
        a0[0] += elem(32 * i + sub);
        a0[1] += elem(32 * i + sub + 16);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) { a0[s] += a1[s]; a0[s] += a2[s]; a0[s] += a3[s]; }
    for (int v = size * 4; v < nvec; ++v) a0[0] += (sub < 8) ? elem(v * 8 + sub) : 0.f;
    const float o0 = __shfl(a0[0], (sub & 7) + 8, 16), o1 = __shfl(a0[1], (sub & 7) + 8, 16);
    const float p = ((a0[0] + o0) + a0[1]) + o1;       // valid on lanes sub < 8
    float fin = 0.f;
    for (int k = nvec * 8; k < n; ++k) fin += elem(k);
    if (nvec > 0) {
#pragma unroll
        for (int l = 0; l < 8; ++l) fin += __shfl(p, l, 16);
    }
    return fin;
}
