// NUMERICALLY WRONG AS IT STANDS (round 4: the probe's equality check against the shipped kernel is RED — its 12-byte global_load_lds_dwordx3
// pieces were never validated): a timing experiment kept for its measurements (profiles/README.md, round 4), outside the build.
// ivit_gemm4.h — token-stationary QuantLinear for K = 384 (the qkv projection of DeiT-S / Swin stage 2):
//   out = requant((x (M x 384 int8) * W^T + bias) * c) scattered into q / k / v^T        (quant_modules.py:67-97, vit_quant.py:64-68)
//
// Round 4.  The persistent kernels of ivit_gemm3.h walk (256-token panel, 128-channel tile) units: two LDS fragment reads
// per MFMA, a staged output tile, a workgroup barrier per k-step.  Here the roles are turned around (the structure measured
// in tools/ubench/mlpr_experiment, with what that experiment taught about this compiler):
//   * a wave owns 32 TOKENS for the whole launch: their 384 input bytes are the MFMA's B operand and stay in 48 registers;
//     eight waves per workgroup (two per SIMD), one workgroup per CU, so DeiT-S's 197 tokens per CU are ONE pass;
//   * the weights stream L2 -> LDS once per workgroup through an 8 x 12 KB global_load_lds ring (a stage = one 32-channel
//     tile = 12 fragments of 1 KB in MFMA-fragment order, laid out at plan time), six stages ahead, ONE raw s_barrier per
//     stage; every wave reads every fragment with one conflict-free ds_read_b128 per v_mfma_i32_32x32x32_i8;
//   * the accumulator layout (lane = token, 4 consecutive channels per register) packs a requantised dword for free; the
//     exact fp64 requant of tile s - 1 (v_cvt_f64_i32 + v_fma_f64, per-channel constants from LDS) runs beside the MFMAs of
//     tile s; results go straight to q / k (dwords) and v^T (bytes, token-contiguous across lanes): no staging tile;
//   * hipcc turns every LDS wait into lgkmcnt(0) while an LDS-DMA is pending, so a stage is two sections that issue all
//     the reads of the NEXT section first and force the drain at their end (T4_DRAIN), behind six MFMAs;
//   * s_waitcnt vmcnt(n) with n counted at compile time over the static schedule of DMAs and stores (t4_younger).
#pragma once
#include <type_traits>
#include "../../../i-vit_amd/csrc/ivit_gemm.h"

#define T4_K 384
#define T4_KS 12
#define T4_WAVES 8
#define T4_THREADS 512
#define T4_TOK 32
#define T4_STAGE 12288
#define T4_NSTG 8
#define T4_DIST 6
#define T4_RING 0
#define T4_OFFC (T4_NSTG * T4_STAGE)          // double c[N]
#define T4_SMEM(N) (T4_OFFC + (N) * 12)       // + int bias[N]
#define T4_MAGIC 6755399441055744.0
#ifndef T4_TRACE
#define T4_TRACE 0
#endif
#ifndef T4_ABLATE                 // probes only: 1 = no output stores, 2 = no requant arithmetic, 4 = no ring barrier
#define T4_ABLATE 0
#endif

struct Tok4Args {
    const int8_t *x;          // [M, 384]
    const int8_t *wf;         // N / 32 stages x 12 KB (t4_swizzle_kernel)
    const double *cq;         // c = m * 2^-e per channel
    const int32_t *bias;      // never null
    int8_t *q, *k, *vt;       // [B*H, T, 64] x 2, [B*H, 64, ldv]
    long long M;
    int T, H, ldv;
    unsigned long long *trace;
};

template <int I, int N, class F>
__device__ __forceinline__ void t4_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        t4_for<I + 1, N>(f);
    }
}

// W [N][384] -> fragment (tile, ks), lane l: 16 bytes W[32 tile + (l & 31)][32 ks + 16 (l >> 5) ...]
__global__ __launch_bounds__(256) void t4_swizzle_kernel(const int8_t *__restrict__ w, int N, int8_t *__restrict__ wf) {
    const long long total = (long long)N * T4_K;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int j = (int)(i & 15), l = (int)((i >> 4) & 63);
        const long long f = i >> 10;
        const int t = (int)(f / T4_KS), ks = (int)(f % T4_KS);
        wf[i] = w[(long long)(32 * t + (l & 31)) * T4_K + 32 * ks + 16 * (l >> 5) + j];
    }
}

typedef __attribute__((address_space(3))) const char t4_lds_c;
typedef __attribute__((address_space(3))) const v4i t4_lds_v4i;
typedef double t4_v2d __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const t4_v2d t4_lds_v2d;

__device__ __forceinline__ v4i t4_load16_async_a(const void *ptr) {
    v4i v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(v) : "v"(ptr) : "memory");
    return v;
}
// see tools/ubench/mlpr_experiment/ivit_mlpr.h: the compiler's wait lands at this use, i.e. at the END of the section that
// issued the reads
#define T4_DRAIN(x) do { __builtin_amdgcn_sched_barrier(0); asm volatile("" :: "v"(x)); __builtin_amdgcn_sched_barrier(0); } while (0)
template <int N> __device__ __forceinline__ void t4_wait_vm() {
    static_assert(N >= 0 && N <= 63, "vmcnt is six bits");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// ---- the static schedule of vector-memory operations of one pass of the qkv flavour (NT tiles, the first 2/3 go to q / k
// as 4 dword stores per tile, the last third to v^T as 16 byte stores), for the counted waits:
//   stage s:  [section A: half the stores of tile s - 1]  [ring step: wait, barrier, 3 DMAs of stage s + 1 + DIST]
//             [section B: the other half]        (2 DMA instructions per wave and stage: dma_at)
constexpr int t4_half_stores(int tile, int NT) { return tile < 0 || tile >= NT ? 0 : (tile < 2 * NT / 3 ? 2 : 8); }
constexpr int t4_dmas(int s, int NT) { return s >= 0 && s + 1 + T4_DIST < NT ? 2 : 0; }
// operations issued after the DMAs of ring step s0 up to (and excluding) the wait of ring step s
constexpr int t4_younger(int s0, int s, int NT) {
    int n = t4_half_stores(s0 - 1, NT);                                     // section B of stage s0
    for (int i = s0 + 1; i < s; ++i) n += 2 * t4_half_stores(i - 1, NT) + t4_dmas(i, NT);
    n += t4_half_stores(s - 1, NT);                                         // section A of stage s
    return n > 63 ? 63 : n;
}

template <int NT, bool FMA>
__global__ __launch_bounds__(T4_THREADS, 2) void tok4_qkv_kernel(Tok4Args p) {
    extern __shared__ __attribute__((aligned(256))) char sm[];
    constexpr int N = NT * 32, D = N / 3, TPW = D / 32;          // channels, channels per q / k / v, tiles per q / k / v
    static_assert(NT % 3 == 0 && D % 64 == 0, "three equal parts of whole 64-channel heads");
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long tb = p.M * blockIdx.x / gridDim.x, te = p.M * (blockIdx.x + 1) / gridDim.x;
    const int cnt = (int)(te - tb);
    if (cnt <= 0) return;
    const int np = (cnt + T4_WAVES * T4_TOK - 1) / (T4_WAVES * T4_TOK);
    const int pbase = (int)((unsigned)cnt / (unsigned)np), prem = cnt - pbase * np;

    // per-layer constants -> LDS (plain loads: no DMA is in flight yet)
    for (int i = tid; i < N / 2; i += T4_THREADS)
        reinterpret_cast<v4i *>(sm + T4_OFFC)[i] = reinterpret_cast<const v4i *>(p.cq)[i];
    for (int i = tid; i < N / 4; i += T4_THREADS)
        reinterpret_cast<v4i *>(sm + T4_OFFC + N * 8)[i] = reinterpret_cast<const v4i *>(p.bias)[i];
    __syncthreads();

    // a stage is 12 KB = 16 pieces of 768 B (global_load_lds_dwordx3: 12 bytes per lane, lane-linear on both sides): every
    // wave issues pieces w and w + 8 — the same two instructions in every wave, no branch
    auto dma_at = [&](auto sc, const int8_t *wf, unsigned voff) __attribute__((always_inline)) {
        constexpr int S = decltype(sc)::value;
        if constexpr (S < NT) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int piece = wave + 8 * i;                   // wave-uniform
                const int8_t *sb = wf + ((size_t)S * T4_STAGE) + (size_t)piece * 768;
                asm volatile("" : "+s"(sb));
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(sb + (size_t)voff),
                                                 (__attribute__((address_space(3))) void *)(sm + T4_RING + (S % T4_NSTG) * T4_STAGE + piece * 768),
                                                 12, 0, 0);
            }
        }
    };

    for (int ps = 0; ps < np; ++ps) {
        int tid_l = threadIdx.x;
        asm volatile("" : "+v"(tid_l));
        const int lane = tid_l & 63, n = lane & 31, h = lane >> 5;
        const int8_t *wf = p.wf;
        asm volatile("" : "+s"(wf));
        const unsigned dma_voff = (unsigned)lane * 12u;
        t4_lds_c *cbase = (t4_lds_c *)sm + T4_OFFC + 32 * h, *bbase = (t4_lds_c *)sm + T4_OFFC + N * 8 + 16 * h;
        t4_lds_c *wbase = (t4_lds_c *)sm + T4_RING + lane * 16;
        asm volatile("" : "+v"(cbase), "+v"(bbase), "+v"(wbase));

        // this wave's tokens: the workgroup's range cut evenly into passes, a pass evenly into eight waves; lanes beyond the
        // wave's tokens repeat its last token (a wave without tokens: the pass's first) and store the same bytes again
        const int p0 = ps * pbase + min(ps, prem), len = pbase + (ps < prem ? 1 : 0);
        const int qn = (len + T4_WAVES - 1) >> 3;
        int my0 = p0 + wave * qn;
        const int nvalid = max(0, min(qn, p0 + len - my0));
        if (nvalid == 0) my0 = p0;
        const long long tok = tb + my0 + min(n, max(nvalid - 1, 0));
        // token -> (image, position): float estimate + one correction (tok < 2^23, host-checked)
        int bimg = (int)((float)tok * (1.0f / (float)p.T));
        int tpos = (int)tok - bimg * p.T;
        if (tpos < 0) { --bimg; tpos += p.T; }
        if (tpos >= p.T) { ++bimg; tpos -= p.T; }
        const unsigned voff_qk = (unsigned)((bimg * p.H * p.T + tpos) * 64 + 4 * h);
        const unsigned voff_v = (unsigned)((bimg * p.H * 64 + 4 * h) * p.ldv + tpos);

        // ---- prologue: activations (B operand, AGPRs), the first ring stages
        v4i xf[T4_KS];
        {
            const int8_t *xp = p.x + tok * T4_K + 16 * h;
#pragma unroll
            for (int ks = 0; ks < T4_KS; ++ks) xf[ks] = t4_load16_async_a(xp + 32 * ks);
        }
        t4_for<0, T4_DIST + 1>([&](auto sc) __attribute__((always_inline)) { dma_at(sc, wf, dma_voff); });
        t4_wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);

        auto read_w = [&](v4i (&w)[6], auto sc, auto jc) __attribute__((always_inline)) {
            constexpr int S = decltype(sc)::value, slot = S % T4_NSTG, j0 = decltype(jc)::value;
            if constexpr (S < NT) {
#pragma unroll
                for (int j = 0; j < 6; ++j) w[j] = *reinterpret_cast<t4_lds_v4i *>(wbase + slot * T4_STAGE + (j0 + j) * 1024);
            }
        };
        auto read_bias = [&](v16i &bias, auto sc) __attribute__((always_inline)) {
            constexpr int S = decltype(sc)::value;
            if constexpr (S < NT) {
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const v4i bb = *reinterpret_cast<t4_lds_v4i *>(bbase + 128 * S + 32 * b);
#pragma unroll
                    for (int e = 0; e < 4; ++e) bias[4 * b + e] = bb[e];
                }
            }
        };
        // multipliers of output groups b0, b0 + 1 of tile TL
        auto read_c = [&](t4_v2d (&c)[4], auto tc, auto bc) __attribute__((always_inline)) {
            constexpr int TL = decltype(tc)::value, B0 = decltype(bc)::value;
            if constexpr (TL >= 0 && TL < NT) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    c[2 * i] = *reinterpret_cast<t4_lds_v2d *>(cbase + 256 * TL + 64 * (B0 + i));
                    c[2 * i + 1] = *reinterpret_cast<t4_lds_v2d *>(cbase + 256 * TL + 64 * (B0 + i) + 16);
                }
            }
        };
        int tr_stage = 0;
        auto stamp = [&]() __attribute__((always_inline)) {
            if (T4_TRACE) {
                if (blockIdx.x == 0 && ps == 0 && (threadIdx.x & 63) == 0) p.trace[wave * 64 + tr_stage] = __builtin_readcyclecounter();
                ++tr_stage;
            }
        };

        v4i wA[6], wB[6];
        v16i bias, acc[2];
        t4_v2d c1[4], c2[4];
        read_w(wA, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        read_bias(bias, std::integral_constant<int, 0>{});
        T4_DRAIN(wA[5]);

        // requant + store of output groups B0, B0 + 1 (8 channels) of tile TL
        auto epi = [&](auto tc, auto bc, t4_v2d (&c)[4]) __attribute__((always_inline)) {
            constexpr int TL = decltype(tc)::value, B0 = decltype(bc)::value;
            if constexpr (TL >= 0 && TL < NT) {
                constexpr int which = TL / TPW, head = (TL % TPW) / 2, d32 = 32 * (TL & 1);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    int o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const double cc = c[2 * i + (e >> 1)][e & 1];
                        const int z = acc[TL & 1][4 * (B0 + i) + e];
                        if (T4_ABLATE & 2) { o[e] = z; continue; }
                        const double t = FMA ? __builtin_fma((double)z, cc, T4_MAGIC) : ((double)z * cc + T4_MAGIC);
                        o[e] = min(max(__double2loint(t), -128), 127);
                    }
                    const unsigned w01 = __builtin_amdgcn_perm((unsigned)o[1], (unsigned)o[0], 0x0c0c0400u);
                    const unsigned w23 = __builtin_amdgcn_perm((unsigned)o[3], (unsigned)o[2], 0x0c0c0400u);
                    unsigned w = __builtin_amdgcn_perm(w23, w01, 0x05040100u);
                    if (T4_ABLATE & 1) { asm volatile("" : "+v"(w)); continue; }
                    if constexpr (which < 2) {
                        int8_t *base = (which == 0 ? p.q : p.k) + (size_t)head * p.T * 64 + d32 + 8 * (B0 + i);
                        *reinterpret_cast<unsigned *>(base + voff_qk) = w;
                    } else {
                        int8_t *base = p.vt + (size_t)(head * 64 + d32 + 8 * (B0 + i)) * p.ldv;
#pragma unroll
                        for (int e = 0; e < 4; ++e) base[(size_t)voff_v + (size_t)e * p.ldv] = (int8_t)(w >> (8 * e));
                    }
                }
            }
        };

        t4_for<0, NT>([&](auto sc) __attribute__((always_inline)) {
            constexpr int S = decltype(sc)::value;
            using TPrev = std::integral_constant<int, S - 1>;
            // ---- section A
            __builtin_amdgcn_sched_barrier(0);
            read_w(wB, sc, std::integral_constant<int, 6>{});
            read_c(c2, TPrev{}, std::integral_constant<int, 2>{});
            __builtin_amdgcn_sched_barrier(0);
            t4_for<0, 6>([&](auto kc) __attribute__((always_inline)) {
                constexpr int ks = decltype(kc)::value;
                acc[S & 1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wA[ks], xf[ks], ks == 0 ? bias : acc[S & 1], 0, 0, 0);
                if constexpr (ks == 1) epi(TPrev{}, std::integral_constant<int, 0>{}, c1);
            });
            T4_DRAIN(wB[5]);
            // ---- ring step: stage S + 1 is complete for everybody afterwards
            stamp();
            if constexpr (S + 1 < NT) {
                t4_wait_vm<(S - T4_DIST < 0) ? 63 : ((T4_ABLATE & 1) ? 2 * (T4_DIST - 1) : t4_younger(S - T4_DIST, S, NT))>();
                if (!(T4_ABLATE & 4)) __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                dma_at(std::integral_constant<int, S + 1 + T4_DIST>{}, wf, dma_voff);
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- section B
            read_bias(bias, std::integral_constant<int, S + 1>{});
            read_w(wA, std::integral_constant<int, S + 1>{}, std::integral_constant<int, 0>{});
            read_c(c1, sc, std::integral_constant<int, 0>{});
            __builtin_amdgcn_sched_barrier(0);
            t4_for<6, 12>([&](auto kc) __attribute__((always_inline)) {
                constexpr int ks = decltype(kc)::value;
                acc[S & 1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wB[ks - 6], xf[ks], acc[S & 1], 0, 0, 0);
                if constexpr (ks == 7) epi(TPrev{}, std::integral_constant<int, 2>{}, c2);
            });
            T4_DRAIN(c1[3]);
        });
        // the last tile
        __builtin_amdgcn_sched_barrier(0);
        read_c(c2, std::integral_constant<int, NT - 1>{}, std::integral_constant<int, 2>{});
        epi(std::integral_constant<int, NT - 1>{}, std::integral_constant<int, 0>{}, c1);
        epi(std::integral_constant<int, NT - 1>{}, std::integral_constant<int, 2>{}, c2);
        __builtin_amdgcn_sched_barrier(0);
        if (ps + 1 < np) __syncthreads();              // the ring restarts: nobody may still read its slots
    }
}
