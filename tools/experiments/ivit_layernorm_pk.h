// ivit_layernorm_pk.h — EXPERIMENT (round 6), not part of the library: the register-resident I-LayerNorm + requant
// (i-vit_amd/csrc/ivit_layernorm.h) with its element-wise fp32 arithmetic written as hand-packed VOP3P instructions in PLAIN form —
// v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 on element pairs, negation by neg_lo / neg_hi, NO op_sel (the form round 5 found faulty
// beside MFMA-issuing waves: profiles/r05_hazard/README.md).  Same operation sequence as LnGroup::run, every product and sum rounded
// separately, so the output is byte-identical (tools/ubench/ln_pk_probe.hip: 0 differing bytes at C = 384 / 768 / 192).
//
// It was INTEGRATED for a day (a second translation unit built with packed fp32 on and -fno-slp-vectorize, an ISA test that allowed
// packed fp32 inside this kernel only and never with op_sel, 60 000 launches beside the K = 48 MFMA aggressors + tools/op_stress.py +
// tools/swin_stress.py: 0 differences) and then taken out again, on these measurements (profiles/README.md, round 6):
//   * a packed instruction issues in 5.3 cycles against 3.1-4.1 for the single forms: 1.55x per FMA, ~1.2x per add / mul, not 2x;
//   * stand-alone 19.6 -> 18.5 us at 50 432 x 384, 36.6 -> 34.9 at 50 432 x 768, SLOWER at C = 192 (register budget);
//   * whole model, same box, interleaved: DeiT-S one eager stream -0.5 % (3.000 / 2.990 / 2.995 against 3.015 / 3.010 / 3.001 ms),
//     DeiT-S 2 slices + graph +1.0 % (3.055 against 3.030 / 3.018), DeiT-B sliced +0.75 %, ViT-B@384 sliced -0.5 %, Swin-T -0.45 % —
//     beside another slice's MFMA kernels the packed forms cost their neighbours more than they save (the guide's "anti-lever").
// A wash within +-1 % does not pay for a second build unit and a relaxed library-wide ISA guard.
// Compile WITHOUT -target-feature -packed-fp32-ops (the assembler rejects the mnemonics otherwise) and with -fno-slp-vectorize.
#pragma once
#include "../../i-vit_amd/csrc/ivit_layernorm.h"

typedef float ln_v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ ln_v2f lnpk_mul(ln_v2f a, ln_v2f b) { ln_v2f r; asm("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ ln_v2f lnpk_add(ln_v2f a, ln_v2f b) { ln_v2f r; asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ ln_v2f lnpk_sub(ln_v2f a, ln_v2f b) { ln_v2f r; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }
// a * b - c, one rounding
__device__ __forceinline__ ln_v2f lnpk_fms(ln_v2f a, ln_v2f b, ln_v2f c) { ln_v2f r; asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
// -a * b + c, one rounding
__device__ __forceinline__ ln_v2f lnpk_fnma(ln_v2f a, ln_v2f b, ln_v2f c) { ln_v2f r; asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
// fl(fl(q * d) / d) on a pair (requotient_m)
__device__ __forceinline__ ln_v2f lnpk_requotient(ln_v2f q, ln_v2f d, ln_v2f yd) {
    const ln_v2f n = lnpk_mul(q, d);
    const ln_v2f e = lnpk_fms(q, d, n);          // exact: q*d - fl(q*d)
    return lnpk_fnma(e, yd, q);
}

template <int CC, int S>
__global__ __launch_bounds__(LNR_THREADS(S), LNR_MIN_WAVES(CC, S)) void layernorm_pk_kernel(const int16_t *__restrict__ x, long long rows,
                                                                    long long row_stride, float s,
                                                                    const float *__restrict__ bias_int,
                                                                    const float *__restrict__ sc,
                                                                    const ivit_dyadic *__restrict__ dy,
                                                                    int8_t *__restrict__ out) {
    static_assert(S == 2 || S == 4, "pairs of elements per lane");
    typedef LnGroup<CC, S> G;
    constexpr int LPR = G::LPR, EPC = G::EPC, NP = EPC / 2, NSTEP = G::NSTEP, RPW = G::RPW, RPB = (LNR_THREADS(S) / 64) * RPW;
    __shared__ __attribute__((aligned(16))) double cC[CC];
    __shared__ __attribute__((aligned(16))) float cB[CC], cSc[CC], cY[CC];
    const int tid = threadIdx.x;
    const bool fastrq = ln_stage_constants<CC, LNR_THREADS(S)>(bias_int, sc, dy, cC, cB, cSc, cY);
    const int lane = tid & 63, j = lane % LPR, k = j / S, hh = j % S, cb0 = 8 * k + EPC * hh;
    const long long row_raw = (long long)blockIdx.x * RPB + (tid >> 6) * RPW + lane / LPR;
    const bool live = row_raw < rows;
    const long long row = live ? row_raw : rows - 1;
    const int16_t *xp = x + row * row_stride + cb0;
    const float ys = rcp_rn(s);
    const ln_v2f s2 = {s, s}, ys2 = {ys, ys};

    // ---- load, x = fl(fl(Q*s)/s)
    ln_v2f xv[NSTEP][NP];
#pragma unroll
    for (int i = 0; i < NSTEP; ++i) {
        const typename LnRaw<EPC>::T t = *reinterpret_cast<const typename LnRaw<EPC>::T *>(xp + 32 * i);
#pragma unroll
        for (int e = 0; e < NP; ++e) xv[i][e] = lnpk_requotient(ln_v2f{(float)t[2 * e], (float)t[2 * e + 1]}, s2, ys2);
    }
    // ---- first sum
    ln_v2f a0[NP], a1[NP];
    float f0[EPC], f1[EPC];
    auto unpack = [&]() {
#pragma unroll
        for (int e = 0; e < NP; ++e) { f0[2 * e] = a0[e][0]; f0[2 * e + 1] = a0[e][1]; f1[2 * e] = a1[e][0]; f1[2 * e + 1] = a1[e][1]; }
    };
    auto cascade = [&](int i) {
        if (((i + 1) & 15) == 0 && i + 1 <= (NSTEP & ~15)) {
#pragma unroll
            for (int e = 0; e < NP; ++e) { a1[e] = lnpk_add(a1[e], a0[e]); a0[e] = ln_v2f{0.f, 0.f}; }
        }
    };
#pragma unroll
    for (int e = 0; e < NP; ++e) { a0[e] = ln_v2f{0.f, 0.f}; a1[e] = ln_v2f{0.f, 0.f}; }
#pragma unroll
    for (int i = 0; i < NSTEP; ++i) {
#pragma unroll
        for (int e = 0; e < NP; ++e) a0[e] = lnpk_add(a0[e], xv[i][e]);
        cascade(i);
    }
    unpack();
    const float mean = rintf(G::finish(f0, f1, j, k) / (float)CC);
    const ln_v2f mean2 = {mean, mean};
    // ---- pass 2: y = x - mean (kept), second sum
#pragma unroll
    for (int e = 0; e < NP; ++e) { a0[e] = ln_v2f{0.f, 0.f}; a1[e] = ln_v2f{0.f, 0.f}; }
#pragma unroll
    for (int i = 0; i < NSTEP; ++i) {
#pragma unroll
        for (int e = 0; e < NP; ++e) {
            const ln_v2f y = lnpk_sub(xv[i][e], mean2);
            xv[i][e] = y;
            a0[e] = lnpk_add(a0[e], lnpk_mul(y, y));
        }
        cascade(i);
    }
    unpack();
    const float var = G::finish(f0, f1, j, k);
    float kk = 65536.0f;
    for (int n = 0; n < 10; ++n) {
        const float kn = floorf((kk + floorf(var / kk)) * 0.5f);
        const bool same = (kn == kk);
        kk = kn;
        if (__all(same)) break;
    }
    const float F = floorf((1.0f / kk) * 2147483648.0f);
    const float Fh = F * 0.5f;
    const ln_v2f Fh2 = {Fh, Fh};
    // ---- pass 3
    int8_t *op = out + row * CC + cb0;
    auto pass3 = [&](auto fast) {
#pragma unroll
    for (int i = 0; i < NSTEP; ++i) {
        const int cb = 32 * i + cb0;
        ln_v2f bi[NP], scv[NP], yv[NP];
        double cv[EPC];
        if constexpr (EPC >= 4) {
#pragma unroll
            for (int e4 = 0; e4 < EPC; e4 += 4) {
                const v4f b4 = *reinterpret_cast<const v4f *>(cB + cb + e4), s4 = *reinterpret_cast<const v4f *>(cSc + cb + e4),
                          y4 = *reinterpret_cast<const v4f *>(cY + cb + e4);
                bi[e4 / 2] = ln_v2f{b4[0], b4[1]}; bi[e4 / 2 + 1] = ln_v2f{b4[2], b4[3]};
                scv[e4 / 2] = ln_v2f{s4[0], s4[1]}; scv[e4 / 2 + 1] = ln_v2f{s4[2], s4[3]};
                yv[e4 / 2] = ln_v2f{y4[0], y4[1]}; yv[e4 / 2 + 1] = ln_v2f{y4[2], y4[3]};
            }
        } else {
            bi[0] = *reinterpret_cast<const ln_v2f *>(cB + cb); scv[0] = *reinterpret_cast<const ln_v2f *>(cSc + cb); yv[0] = *reinterpret_cast<const ln_v2f *>(cY + cb);
        }
#pragma unroll
        for (int e = 0; e < EPC; e += 2) {
            typedef double v2d __attribute__((ext_vector_type(2)));
            const v2d c2 = *reinterpret_cast<const v2d *>(cC + cb + e);
            cv[e] = c2[0]; cv[e + 1] = c2[1];
        }
        unsigned pk[2] = {0, 0};
#pragma unroll
        for (int e = 0; e < NP; ++e) {
            const ln_v2f t = lnpk_mul(xv[i][e], Fh2);
            const ln_v2f o = lnpk_add(ln_v2f{floorf(t[0]), floorf(t[1])}, bi[e]);
            const ln_v2f q = lnpk_requotient(o, scv[e], yv[e]);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float zz = rintf(q[h]);
                const int el = 2 * e + h;
                if constexpr (decltype(fast)::value) {
                    const int v = __double2loint((double)zz * cv[el] + (6755399441055744.0 + 128.0));
                    pk[el >> 2] |= (unsigned)min(max(v, 0), 255) << (8 * (el & 3));
                } else {
                    const int v = rq_c((double)zz, cv[el], -128, 127);
                    pk[el >> 2] |= ((unsigned)v & 0xffu) << (8 * (el & 3));
                }
            }
        }
        if constexpr (decltype(fast)::value) { pk[0] ^= 0x80808080u; pk[1] ^= 0x80808080u; }
        if (live) {
            if constexpr (EPC == 8) *reinterpret_cast<v2i *>(op + 32 * i) = v2i{(int)pk[0], (int)pk[1]};
            else if constexpr (EPC == 4) *reinterpret_cast<unsigned *>(op + 32 * i) = pk[0];
            else *reinterpret_cast<unsigned short *>(op + 32 * i) = (unsigned short)pk[0];
        }
    }
    };
    if (fastrq) pass3(std::true_type{});
    else pass3(std::false_type{});
}
