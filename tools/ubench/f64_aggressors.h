// One-instruction-class aggressor kernels shared by pk_f64_hazard.hip (synthetic victim) and ln_s1_standalone.hip (the real
// LayerNorm victim): every wave issues ONE kind of instruction in a tight loop, 16 per iteration.
#pragma once
#include <hip/hip_runtime.h>
#define F64_AGGR_KINDS 14
static const char *const f64_aggr_names[F64_AGGR_KINDS] = {"none", "v_rndne_f64", "v_cvt_i32_f64", "v_mul_f64", "v_add_f64", "v_fma_f64", "v_min/max_f64",
    "v_cmp_lt_f64 + v_cndmask", "v_cvt_f64_i32", "v_floor/trunc_f64", "v_rndne/floor_f32", "v_cvt_f64_f32 / v_cvt_f32_f64", "v_rcp_f64", "v_fract / v_ldexp_f64"};
#define REP8(X) X X X X X X X X
template <int KIND>
__global__ __launch_bounds__(256) void aggressor(double *out, int n, double seed) {
    double a = seed + threadIdx.x * 0.37, b = 1.0000001 + threadIdx.x * 1e-9, c = a * 0.5, d = 3.7 + threadIdx.x;
    float fa = (float)a, fb = (float)b;
    int ia = threadIdx.x * 7 + 1, ib = 0;
    for (int it = 0; it < n; ++it) {
        if (KIND == 1) asm volatile(REP8("v_rndne_f64 %0, %1\n\tv_rndne_f64 %2, %3\n\t") : "=&v"(c), "+v"(a), "=&v"(d), "+v"(b));
        if (KIND == 2) asm volatile(REP8("v_cvt_i32_f64 %0, %1\n\tv_cvt_i32_f64 %0, %2\n\t") : "=&v"(ib) : "v"(a), "v"(b));
        if (KIND == 3) asm volatile(REP8("v_mul_f64 %0, %1, %2\n\tv_mul_f64 %3, %1, %2\n\t") : "=&v"(c) : "v"(a), "v"(b), "v"(d));
        if (KIND == 4) asm volatile(REP8("v_add_f64 %0, %1, %2\n\tv_add_f64 %3, %1, %2\n\t") : "=&v"(c) : "v"(a), "v"(b), "v"(d));
        if (KIND == 5) asm volatile(REP8("v_fma_f64 %0, %1, %2, %0\n\tv_fma_f64 %3, %1, %2, %3\n\t") : "+v"(c) : "v"(a), "v"(b), "v"(d));
        if (KIND == 6) asm volatile(REP8("v_min_f64 %0, %1, %2\n\tv_max_f64 %3, %1, %2\n\t") : "=&v"(c) : "v"(a), "v"(b), "v"(d));
        if (KIND == 7) asm volatile(REP8("v_cmp_lt_f64 vcc, %1, %2\n\tv_cndmask_b32 %0, %3, %4, vcc\n\t") : "=&v"(ib) : "v"(a), "v"(b), "v"(ia), "v"(ib) : "vcc");
        if (KIND == 8) asm volatile(REP8("v_cvt_f64_i32 %0, %1\n\tv_cvt_f64_i32 %2, %1\n\t") : "=&v"(c) : "v"(ia), "v"(d));
        if (KIND == 9) asm volatile(REP8("v_floor_f64 %0, %1\n\tv_trunc_f64 %2, %1\n\t") : "=&v"(c) : "v"(a), "v"(d));
        if (KIND == 10) asm volatile(REP8("v_rndne_f32 %0, %1\n\tv_floor_f32 %0, %2\n\t") : "=&v"(fa) : "v"(fa), "v"(fb));
        if (KIND == 11) asm volatile(REP8("v_cvt_f64_f32 %0, %1\n\tv_cvt_f32_f64 %2, %0\n\t") : "=&v"(c) : "v"(fa), "v"(fb));
        if (KIND == 12) asm volatile(REP8("v_rcp_f64 %0, %1\n\tv_rcp_f64 %2, %1\n\t") : "=&v"(c) : "v"(a), "v"(d));
        if (KIND == 13) asm volatile(REP8("v_fract_f64 %0, %1\n\tv_ldexp_f64 %2, %1, 3\n\t") : "=&v"(c) : "v"(a), "v"(d));
    }
    out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + fa + fb + ia + ib;
}


static inline void launch_f64_aggressor(int kind, double *sink, unsigned grid, int n, hipStream_t s) {
    switch (kind) {
#define F64_CASE(K) case K: aggressor<K><<<grid, 256, 0, s>>>(sink, n, 12345.678); break;
        F64_CASE(1) F64_CASE(2) F64_CASE(3) F64_CASE(4) F64_CASE(5) F64_CASE(6) F64_CASE(7) F64_CASE(8) F64_CASE(9) F64_CASE(10) F64_CASE(11) F64_CASE(12) F64_CASE(13)
#undef F64_CASE
        default: break;
    }
}
