// ivit_layernorm.h — I-LayerNorm + per-channel 8-bit requant (a7 + a3), register-resident form.
//
// Reference sequence (quant_modules.py:353-386 followed by the next QuantAct, quant_utils.py:213-253; restated in
// oracle/ivit_oracle.c::ivit_o_layernorm): x = fl(fl(Q*s)/s); mean = rne(SUM(x)/C); y = x - mean; v = SUM(fl(y*y));
// ten integer Newton steps for sqrt(v); F = floor(2^31/k); o = floor(fl(y*F)/2) + bias_int; z = rne(fl(fl(o*sc)/sc));
// out = clamp(rne(z*c), -128, 127).  Both SUMs are torch-CPU's vectorised order (ivit_device.h::torch_order_sum32):
// element 32 i + 8 k + l goes to accumulator (k, l) at step i.
//
// Layout: 4 S lanes per row (S = 1, 2, 4 for C <= 384, 768, 1536).  Lane (k, h) of a row owns the 8 / S vector lanes
// l = (8 / S) h ... of accumulator group k, i.e. the contiguous 16 / S bytes of every 32-element step: the row is read
// ONCE with plain vector loads, lives in registers as fp32 (C / (4 S) <= 96 values per lane) through both sums and the
// output pass, and never touches the LDS (the round-2 kernel staged it there, swizzled, and re-read it three times).
// The accumulator groups of a row meet once per sum: k = 0..3 by DPP quad broadcasts (S = 1) in the reference's order
// ((a0 + a1) + a2) + a3, then the eight vector lanes sequentially.
//
// Divisions by a per-tensor / per-channel constant d: with yd = RN(1/d) and a faithful first guess q0, Markstein's
// correction q = fma(fma(-d, q0, n), yd, q0) is the correctly rounded n / d.  Here n = fl(q0 * d) for a 24-bit q0
// (the quotient IS the operand we multiplied by d, up to the product's rounding), so q0 is faithful and the residual
// fma(q0, d, -n) is exact: three VALU operations instead of the eleven of an IEEE division or the six of the hoisted
// Newton form (ivit_device.h::lean_div).  Pinned against `/` in tests/test_gpu_parity.py::test_markstein_requotient.
#pragma once
#include <type_traits>
#include "ivit_device.h"

// fl(fl(q * d) / d) for a float q with |q| < 2^24 * ulp-safe range, yd = RN(1 / d)
__device__ __forceinline__ float requotient_m(float q, float d, float yd) {
    const float n = q * d;
    const float e = __builtin_fmaf(q, d, -n);       // exact: q*d - fl(q*d)
    return __builtin_fmaf(-e, yd, q);
}
// correctly rounded reciprocal: fl64(1/d) rounded to binary32 cannot sit on a binary32 midpoint unless d is a power of two
// (then it is exact), so the double rounding is harmless
__device__ __forceinline__ float rcp_rn(float d) { return (float)(1.0 / (double)d); }

template <int Q>
__device__ __forceinline__ float ln_quad_bcast(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), Q * 0x55, 0xf, 0xf, true));
}

// v[lane'] by a DPP control word: 0x100 + n = row_shl:n (lane i reads lane i + n), 0x110 + n = row_shr:n (lane i reads
// lane i - n), both inside a row of 16 lanes; lanes without a source read 0
template <int CTRL>
__device__ __forceinline__ float ln_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

// S = 1 (4 lanes per row, quad_perm broadcasts) is NOT dispatched (S = 2 is the faster form) and does not compile outside the
// probe build (-DIVIT_PROBE_LN192_S1=1, tools/ln_s1_probe.sh, tools/ubench/ln_s1_standalone.hip).  History: it returned one-LSB
// differences in rows 12..15 of a wave in 1-4 % of the launches that shared a SIMD with MFMA-issuing waves.  Round 5 reduced that
// to one instruction form hipcc used in its running sums — v_pk_add_f32 ... op_sel:[0,1] op_sel_hi:[1,0], whose low lane takes
// src1's HIGH dword: beside MFMAs that dword is occasionally read as 0 on lanes 48..63 (profiles/r05_hazard/README.md,
// tools/ubench/pk_opsel_hazard.hip).  The library is built with -packed-fp32-ops off (i-vit_amd/_lib.py), which removes the form
// here, in the S = 4 kernels and in the token-order kernels whose ISA had packed fp32.
// 32 rows per block whatever the split: the per-block staging of the channel constants (an fp64 division each) stays ~8 %
#ifndef LNR_TB
#define LNR_TB 128
#endif
#define LNR_THREADS(S) (LNR_TB * (S))
// timing probes only (tools/ubench/ln_probe.hip): 1 = no output-pass arithmetic, 2 = no second sum, 4 = no Newton loop
#ifndef LNR_ABLATE
#define LNR_ABLATE 0
#endif
// probes of the S = 1 form only (tools/ln_s1_probe.sh): 1 = s_nop 7 around every DPP group, 2 = ds_bpermute instead of DPP,
// 3 = per-channel constants straight from global memory instead of the LDS copy
#ifndef LNR_S1_VARIANT
#define LNR_S1_VARIANT 0
#endif
#ifndef IVIT_PROBE_LN192_S1
#define IVIT_PROBE_LN192_S1 0
#endif
// register budget by values per lane (CC / 4S): <= 24 -> 8 waves per SIMD, <= 48 -> 5, <= 64 -> 4, more -> 3 (no scratch in any
// instantiation the dispatcher uses)
#define LNR_VPL(CC, S) ((CC) / (4 * (S)))
#ifndef LNR_W48
#define LNR_W48 5
#endif
#define LNR_MIN_WAVES(CC, S) (LNR_VPL(CC, S) <= 24 ? 8 : (LNR_VPL(CC, S) <= 48 ? LNR_W48 : (LNR_VPL(CC, S) <= 64 ? 4 : 3)))
// Everything after the row is in registers as x = fl(fl(Q*s)/s): both torch-order sums, the integer Newton iteration, the
// output pass with the per-channel constants of the block's LDS copy, the store.  Shared by the one-shot kernel and the
// pipelined one below.
template <int CC, int S>
struct LnGroup {
    static constexpr int LPR = 4 * S, EPC = 8 / S, NSTEP = CC / 32, RPW = 64 / LPR;
    static_assert(CC % 32 == 0 && NSTEP < 256, "whole 32-element steps, at most one cascade level above the first");

    // ((g0 + g1) + g2) + g3 over the accumulator groups, then vector lanes 0..7 in order; every lane of the row gets it.
    // Lane j = S k + h of a row: group k + 1 is S lanes up (DPP row shifts; a row's 4 S lanes never straddle a DPP row of
    // 16), the partial sums are valid on the k = 0 lanes, the total on lane j = 0, from where it is broadcast.
    static __device__ __forceinline__ float finish(const float (&a0)[EPC], const float (&a1)[EPC], int j, int k) {
        float p[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            const float a = NSTEP >= 16 ? a0[e] + a1[e] : a0[e];
            if constexpr (S == 1) {
#if LNR_S1_VARIANT == 2
                // probe: the same four-lane exchange through the LDS crossbar (ds_bpermute) instead of DPP
                const int l0 = (int)(threadIdx.x & 63) & ~3;
                float t = __shfl(a, l0);
                t += __shfl(a, l0 + 1);
                t += __shfl(a, l0 + 2);
                t += __shfl(a, l0 + 3);
                p[e] = t;
#else
                float aa = a;
#if LNR_S1_VARIANT == 1
                asm volatile("s_nop 7" : "+v"(aa));        // probe: wait states between the producer of `a` and its DPP readers
#endif
                float t = ln_quad_bcast<0>(aa);
                t += ln_quad_bcast<1>(aa);
                t += ln_quad_bcast<2>(aa);
                t += ln_quad_bcast<3>(aa);
#if LNR_S1_VARIANT == 1
                asm volatile("s_nop 7" : "+v"(t));
#endif
                p[e] = t;
#endif
            } else {
                float t = a + ln_dpp<0x100 + S>(a);
                t += ln_dpp<0x100 + 2 * S>(a);
                t += ln_dpp<0x100 + 3 * S>(a);
                p[e] = t;
            }
        }
        float fin = 0.f;
#pragma unroll
        for (int e = 0; e < EPC; ++e) fin += p[e];
        if constexpr (S == 2) {
#pragma unroll
            for (int e = 0; e < EPC; ++e) fin += ln_dpp<0x101>(p[e]);
            const float q0 = ln_quad_bcast<0>(fin), q1 = ln_dpp<0x114>(q0);      // lanes 0..3 | 4..7 of the row
            fin = (j & 4) ? q1 : q0;
        } else if constexpr (S == 4) {
#pragma unroll
            for (int e = 0; e < EPC; ++e) fin += ln_dpp<0x101>(p[e]);
#pragma unroll
            for (int e = 0; e < EPC; ++e) fin += ln_dpp<0x102>(p[e]);
#pragma unroll
            for (int e = 0; e < EPC; ++e) fin += ln_dpp<0x103>(p[e]);
            const float q0 = ln_quad_bcast<0>(fin), q1 = ln_dpp<0x114>(q0), q2 = ln_dpp<0x118>(q0), q3 = ln_dpp<0x11c>(q0);
            fin = k == 0 ? q0 : (k == 1 ? q1 : (k == 2 ? q2 : q3));
        }
        return fin;
    }

    // torch's level-1 fold after every 16 whole steps
    static __device__ __forceinline__ void cascade(int i, float (&a0)[EPC], float (&a1)[EPC]) {
        if (((i + 1) & 15) == 0 && i + 1 <= (NSTEP & ~15)) {
#pragma unroll
            for (int e = 0; e < EPC; ++e) { a1[e] += a0[e]; a0[e] = 0.f; }
        }
    }

    // cb0 = 8 k + EPC h: this lane's first channel of every 32-element step; op = out + row * CC + cb0
    // DMA_SYNC: the constants arrive by LDS-DMA (planned kernel): wait for them — this wave's own transfers by vmcnt, the other
    // waves' by the barrier — in front of the output pass, the first reader
    // OP: int8_t * (the row's bytes go to op + 32 i, global memory), or a callable op(i, pk0, pk1) that places step i's
    // EPC bytes itself (ivit_gemm_ws.h: the consumer's LDS image)
    template <typename DMA_SYNC = std::false_type, typename OP = int8_t *>
    static __device__ __forceinline__ void run(float (&xv)[NSTEP][EPC], int j, int k, int cb0, bool fastrq, bool live,
                                               const double *cC, const float *cB, const float *cSc, const float *cY,
                                               const float *bias_int, const float *sc, const ivit_dyadic *dy, OP op,
                                               DMA_SYNC = DMA_SYNC{}) {
        // ---- first sum
        float a0[EPC], a1[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) { a0[e] = 0.f; a1[e] = 0.f; }
#pragma unroll
        for (int i = 0; i < NSTEP; ++i) {
#pragma unroll
            for (int e = 0; e < EPC; ++e) a0[e] += xv[i][e];
            cascade(i, a0, a1);
        }
        const float mean = rintf(finish(a0, a1, j, k) / (float)CC);

        // ---- pass 2: y = x - mean (kept), second sum
#pragma unroll
        for (int e = 0; e < EPC; ++e) { a0[e] = 0.f; a1[e] = 0.f; }
#pragma unroll
        for (int i = 0; i < NSTEP; ++i) {
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                const float y = xv[i][e] - mean;
                xv[i][e] = y;
                if (!(LNR_ABLATE & 2)) a0[e] += y * y;
            }
            cascade(i, a0, a1);
        }
        const float var = finish(a0, a1, j, k);
        // integer Newton iteration; k' == k is a fixed point of the remaining steps, so the early exit is exact
        float kk = 65536.0f;
        for (int n = 0; n < ((LNR_ABLATE & 4) ? 0 : 10); ++n) {
            const float kn = floorf((kk + floorf(var / kk)) * 0.5f);
            const bool same = (kn == kk);
            kk = kn;
            if (__all(same)) break;
        }
        const float F = floorf((1.0f / kk) * 2147483648.0f);
        const float Fh = F * 0.5f;       // fl(fl(y * F) * 0.5) == fl(y * (F * 0.5)): the power of two commutes with the rounding
        if constexpr (DMA_SYNC::value) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }

        // ---- pass 3: normalise, requotient by the channel scale, 8-bit requant, store
        auto pass3 = [&](auto fast) {
#pragma unroll
        for (int i = 0; i < NSTEP; ++i) {
            const int cb = 32 * i + cb0;
            float bi[EPC], scv[EPC], yv[EPC];
            double cv[EPC];
#pragma unroll
            for (int e4 = 0; e4 < EPC; e4 += (EPC >= 4 ? 4 : 2)) {
                if constexpr (EPC >= 4) {
#if LNR_S1_VARIANT == 3
                    v4f b4 = *reinterpret_cast<const v4f *>(bias_int + cb + e4), s4 = *reinterpret_cast<const v4f *>(sc + cb + e4), y4;
                    for (int e = 0; e < 4; ++e) y4[e] = rcp_rn(s4[e]);
#else
                    const v4f b4 = *reinterpret_cast<const v4f *>(cB + cb + e4), s4 = *reinterpret_cast<const v4f *>(cSc + cb + e4),
                              y4 = *reinterpret_cast<const v4f *>(cY + cb + e4);
#endif
#pragma unroll
                    for (int e = 0; e < 4; ++e) { bi[e4 + e] = b4[e]; scv[e4 + e] = s4[e]; yv[e4 + e] = y4[e]; }
                } else {
                    bi[0] = cB[cb]; bi[1] = cB[cb + 1]; scv[0] = cSc[cb]; scv[1] = cSc[cb + 1]; yv[0] = cY[cb]; yv[1] = cY[cb + 1];
                }
            }
#pragma unroll
            for (int e = 0; e < EPC; e += 2) {
                typedef double v2d __attribute__((ext_vector_type(2)));
#if LNR_S1_VARIANT == 3
                cv[e] = dy[cb + e].m * dy[cb + e].r; cv[e + 1] = dy[cb + e + 1].m * dy[cb + e + 1].r;
#else
                const v2d c2 = *reinterpret_cast<const v2d *>(cC + cb + e);
                cv[e] = c2[0]; cv[e + 1] = c2[1];
#endif
            }
            unsigned pk[2] = {0, 0};
            if constexpr (decltype(fast)::value) {
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    const float o = floorf(xv[i][e] * Fh) + bi[e];
                    if (LNR_ABLATE & 1) { pk[e >> 2] |= ((unsigned)__float_as_int(o) & 0xffu) << (8 * (e & 3)); continue; }
                    const float zz = rintf(requotient_m(o, scv[e], yv[e]));
                    const int v = __double2loint((double)zz * cv[e] + (6755399441055744.0 + 128.0));
                    pk[e >> 2] |= (unsigned)min(max(v, 0), 255) << (8 * (e & 3));
                }
                pk[0] ^= 0x80808080u; pk[1] ^= 0x80808080u;
            } else {
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    const float o = floorf(xv[i][e] * Fh) + bi[e];
                    const float zz = rintf(requotient_m(o, scv[e], yv[e]));
                    const int v = rq_c((double)zz, cv[e], -128, 127);
                    pk[e >> 2] |= ((unsigned)v & 0xffu) << (8 * (e & 3));
                }
            }
            if constexpr (!std::is_pointer<OP>::value) {
                if (live) op(i, pk[0], pk[1]);
            } else if (live) {
                if constexpr (EPC == 8) *reinterpret_cast<v2i *>(op + 32 * i) = v2i{(int)pk[0], (int)pk[1]};
                else if constexpr (EPC == 4) *reinterpret_cast<unsigned *>(op + 32 * i) = pk[0];
                else *reinterpret_cast<unsigned short *>(op + 32 * i) = (unsigned short)pk[0];
            }
        }
        };
        // one branch around the whole pass (left inside, both forms are evaluated per element and selected)
        if (fastrq) pass3(std::true_type{});
        else pass3(std::false_type{});
    }
};

// per-channel constants of one LayerNorm -> the block's LDS copy; returns whether every channel admits the two-operation
// 8-bit requant.  The requant rne(fl64(z * c)) is taken as the low dword of fl64(z * c) + (1.5 * 2^52 + 128) — the same two
// roundings as the reference (quant_utils.py:229-231) in two fp64 operations instead of four, biased to 0..255 so the four
// bytes of a dword pack without masks.  That needs |z * c| < 2^31: |y| <= 2^16 and k >= 2^16 / 2^10 after ten halvings at most
// bound |o| by 2^40 + |bias|; a block with a channel where that bound fails keeps v_rndne_f64 + the saturating v_cvt_i32_f64.
template <int CC, int THREADS>
__device__ __forceinline__ bool ln_stage_constants(const float *__restrict__ bias_int, const float *__restrict__ sc,
                                                   const ivit_dyadic *__restrict__ dy, double *cC, float *cB, float *cSc, float *cY) {
    bool wide = false;
    for (int c = threadIdx.x; c < CC; c += THREADS) {
        const float scv = sc[c], bv = bias_int[c];
        const double cv = dy[c].m * dy[c].r;
        cSc[c] = scv;
        cY[c] = rcp_rn(scv);
        cB[c] = bv;
        cC[c] = cv;
        wide |= !(fabs(cv) * (1.2e12 + 1.01 * fabs((double)bv)) < 2147483000.0);
    }
    return !__syncthreads_or(wide);
}

template <int EPC>
struct LnRaw;
template <> struct LnRaw<8> { typedef short T __attribute__((ext_vector_type(8))); };
template <> struct LnRaw<4> { typedef short T __attribute__((ext_vector_type(4))); };
template <> struct LnRaw<2> { typedef short T __attribute__((ext_vector_type(2))); };

// MERGE_R > 0 (round 6): the row is PatchMerging's 2 x 2 gather (swin_quant.py:336-342) of x [B, R, R, CC / 4] done in the load —
// merged row (b, yo, xo) = the four tokens (2 yo + (q & 1), 2 xo + (q >> 1)), q = 0..3, side by side in the reference's
// torch.cat order; a 32-element step never straddles two of them (CC / 4 is a multiple of 32), so a step's quarter is a
// compile-time number and the gather costs four row offsets per lane instead of a 2 x 77 MB pass of its own
// (ivit_patch_merge_gather).  `merge_R` = R, `rows` = B (R / 2)^2, `row_stride` unused.
template <int CC, int S, bool MERGE = false>
__global__ __launch_bounds__(LNR_THREADS(S), LNR_MIN_WAVES(CC, S)) void layernorm_reg_kernel(const int16_t *__restrict__ x, long long rows,
                                                                     long long row_stride, float s,
                                                                     const float *__restrict__ bias_int,
                                                                     const float *__restrict__ sc,
                                                                     const ivit_dyadic *__restrict__ dy,
                                                                     int8_t *__restrict__ out, int merge_R = 0) {
#if !IVIT_PROBE_LN192_S1
    static_assert(S != 1, "the 4-lanes-per-row form is a probe (see the note above)");
#endif
    typedef LnGroup<CC, S> G;
    constexpr int LPR = G::LPR, EPC = G::EPC, NSTEP = G::NSTEP, RPW = G::RPW, RPB = (LNR_THREADS(S) / 64) * RPW;
    __shared__ __attribute__((aligned(16))) double cC[CC];
    __shared__ __attribute__((aligned(16))) float cB[CC], cSc[CC], cY[CC];
    const int tid = threadIdx.x;
    const bool fastrq = ln_stage_constants<CC, LNR_THREADS(S)>(bias_int, sc, dy, cC, cB, cSc, cY);
    const int lane = tid & 63, j = lane % LPR, k = j / S, hh = j % S;
    const long long row_raw = (long long)blockIdx.x * RPB + (tid >> 6) * RPW + lane / LPR;
    const bool live = row_raw < rows;
    const long long row = live ? row_raw : rows - 1;          // a dead lane group recomputes the last row, stores nothing
    const int16_t *xp = x + row * row_stride + 8 * k + EPC * hh;
    const float ys = rcp_rn(s);
    const int16_t *xq[4] = {xp, xp, xp, xp};
    if constexpr (MERGE) {
        static_assert((CC / 4) % 32 == 0, "a 32-element step lies inside one of the four merged tokens");
        const int R2 = merge_R >> 1;
        const long long per = (long long)R2 * R2;
        const int b = (int)(row / per), rem = (int)(row - (long long)b * per), yo = rem / R2, xo = rem - yo * R2;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            xq[q] = x + ((((long long)b * merge_R + 2 * yo + (q & 1)) * merge_R + 2 * xo + (q >> 1)) * (CC / 4)) + 8 * k + EPC * hh;
    }

    // ---- load, x = fl(fl(Q*s)/s)
    float xv[NSTEP][EPC];
#pragma unroll
    for (int i = 0; i < NSTEP; ++i) {
        const int16_t *src = MERGE ? xq[(32 * i) / (CC / 4)] + (32 * i) % (CC / 4) : xp + 32 * i;
        const typename LnRaw<EPC>::T t = *reinterpret_cast<const typename LnRaw<EPC>::T *>(src);
#pragma unroll
        for (int e = 0; e < EPC; ++e) xv[i][e] = requotient_m((float)t[e], s, ys);
    }
    G::run(xv, j, k, 8 * k + EPC * hh, fastrq, live, cC, cB, cSc, cY, bias_int, sc, dy, out + row * CC + 8 * k + EPC * hh);
}

// diagnostics: requotient_m against the IEEE sequence fl(fl(q*d)/d), element-wise
__global__ __launch_bounds__(256) void debug_requotient_kernel(const float *__restrict__ q, const float *__restrict__ d,
                                                               float *__restrict__ r_ieee, float *__restrict__ r_m,
                                                               long long count) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < count) {
        const float n = q[i] * d[i];
        r_ieee[i] = n / d[i];
        r_m[i] = requotient_m(q[i], d[i], rcp_rn(d[i]));
    }
}
