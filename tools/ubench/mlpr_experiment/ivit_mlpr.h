// ivit_mlpr.h — Mlp.forward + the block's residual QuantAct for D = 384 / hidden 1536 with the hidden row in REGISTERS:
//   fc1 -> qact_gelu (8 bit) -> ShiftGELU -> qact1 (8 bit) -> fc2 -> qact2 (16 bit) -> qact4(+identity) (16 bit)
// (models/layers_quant.py:144-153, then vit_quant.py:141-142 / swin_quant.py:296-300).  Successor of mlp384_kernel
// (ivit_mlp.h), same entry point (ivit_mlp_fused_planned), same results.
//
// What the round-3 kernel paid for and this one does not: eight waves in lock-step phases around a shared 120 KB hidden
// tile (three workgroup barriers per unit, the matrix pipe idle during ShiftGELU), two LDS fragment reads per MFMA, the
// weights re-read from L2 by every wave (1.18 MB per 80 tokens through the CU's 64 B/clk vector-memory path), fp64
// requants.  Here:
//   * one wave per SIMD (512 registers), v_mfma_i32_32x32x32_i8, weights as the A operand, TOKENS as the B operand:
//     a wave owns 32 tokens for a whole pass and never exchanges activations with another wave;
//   * the fc1 output layout (lane = token, 4 consecutive channels per register) IS the B-fragment layout of fc2 once
//     the K order of W2 is permuted to match (plan time): the 1536-byte hidden row of a token lives in 2 lanes x 192
//     registers from its production to its consumption — no LDS image, no barrier between the GEMMs;
//   * ShiftGELU's row maximum is lane-local + one cross-lane exchange; its table line (256 B per token, XOR-skewed so
//     that 32 lanes looking up the same value hit 32 banks) sits in LDS; the byte gathers of k-step t + 1 are issued
//     between the MFMAs of k-step t;
//   * both weight matrices stream L2 -> LDS once per 128 tokens through an 8-stage global_load_lds ring (12 KB stages,
//     six stages ahead, one raw s_barrier per stage); every fragment read (one ds_read_b128 per MFMA) is shared by
//     nothing and conflicts with nothing (lane-linear 1 KB fragments);
//   * qact_gelu is ONE fp32 FMA + v_cvt_pk_u8_f32 per element: the per-channel fp32 multiplier is chosen and PROVEN
//     against the reference's fp64 expression at every rounding boundary of the 8-bit range at plan time
//     (mlpr_rq8_plan_kernel); a layer that cannot be proven keeps mlp384_kernel.
#pragma once
#include <type_traits>
#include "../../../i-vit_amd/csrc/ivit_device.h"

#define MR_C 384
#define MR_HD 1536
#define MR_WAVES 4
#define MR_THREADS 256
#define MR_TOK 32                         // tokens per wave and pass
#define MR_KS1 (MR_C / 32)                // 12 k-steps of fc1
#define MR_T1 (MR_HD / 32)                // 48 hidden tiles = k-steps of fc2
#define MR_R2 (MR_C / 32)                 // 12 output row tiles of fc2
#define MR_STAGE 12288                    // one ring stage: 12 fragments of 1 KB
#define MR_NSTG 8
#define MR_DIST 6
#define MR_NSTAGES (2 * MR_T1)            // 96 stages per pass: 48 of W1, 48 of W2
#define MR_RING 0
#define MR_C1F (MR_NSTG * MR_STAGE)       // float c1[1536]
#define MR_B1 (MR_C1F + MR_HD * 4)        // int b1[1536]
#define MR_CQ2 (MR_B1 + MR_HD * 4)        // double c2[384]
#define MR_B2 (MR_CQ2 + MR_C * 8)         // int b2[384]
#define MR_TAB (MR_B2 + MR_C * 4)         // 4 waves x 32 tokens x 256 B table lines
#define MR_SMEM (MR_TAB + MR_WAVES * MR_TOK * 256)
#define MR_MAGIC 6755399441055744.0
#ifndef MR_HV
#define MR_HV 34                          // hidden tiles kept in VGPRs; the other 12 wait in AGPRs
#endif
#ifndef MR_TRACE
#define MR_TRACE 0
#endif
// timing ablations (probe builds only; results invalid): 1 = no fc1 requant arithmetic, 2 = no ShiftGELU gathers,
// 4 = no MFMAs, 8 = no weight-fragment reads after the first, 16 = no ring DMA after the prologue, 32 = no ring barrier
#ifndef MR_ABLATE
#define MR_ABLATE 0
#endif
#ifndef MR_RES_STAGE
#define MR_RES_STAGE 88                   // the stage that requests the identity rows (consumed after stage 95)
#endif

struct MlprArgs {
    const int8_t *x;          // [M, 384]
    const int8_t *wf;         // 96 stages x 12 KB: W1 fragments (tile-major), then W2 fragments (k-step-major)
    const float *c1f;         // fc1: proven fp32 multipliers
    const int32_t *b1, *b2;
    const double *cq2;
    const int8_t *tab;        // ShiftGELU(+requant) table [256 maxima][256 values]
    const int16_t *residual;
    int16_t *out;
    double cm, cr;
    float d1;                 // additive constant of the fc1 requant (128.5 for a truncating v_cvt_pk_u8_f32)
    long long M;
    unsigned long long *trace;
};

template <int I, int N, class F>
__device__ __forceinline__ void mr_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        mr_for<I + 1, N>(f);
    }
}

// fc1's requant as the hardware evaluates it: u = q + 128 in [0, 255]
__device__ __forceinline__ unsigned mr_rq8_u(int z, float c, float d, unsigned old, int byte) {
    const float pf = __builtin_fmaf((float)z, c, d);
    return __builtin_amdgcn_cvt_pk_u8_f32(pf, (unsigned)byte, old);
}

// ---- plan: weights -> fragment streams ----------------------------------------------------------------------------
// W1 [1536][384]: fragment (t, ks), lane l: 16 bytes W1[32 t + (l & 31)][32 ks + 16 (l >> 5) ...]
// W2 [384][1536]: two output halves (rows 0-191, then 192-383), each 24 stages of two k-steps x six row tiles:
//   fragment (half, i, k2, r'), lane l, byte j: W2[192 half + 32 r' + (l & 31)][32 t + 8 (j >> 2) + 4 (l >> 5) + (j & 3)], t = 2 i + k2
//   (k' = 16 h + 4 b + e of the B operand is hidden channel 32 t + 8 b + 4 h + e: the register order fc1 leaves behind)
__global__ __launch_bounds__(256) void mlpr_swizzle_kernel(const int8_t *__restrict__ w1, const int8_t *__restrict__ w2,
                                                           int8_t *__restrict__ wf) {
    const long long total = 2LL * MR_C * MR_HD;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int j = (int)(i & 15), l = (int)((i >> 4) & 63);
        const long long f = i >> 10;
        int8_t v;
        if (f < MR_T1 * MR_KS1) {
            const int t = (int)(f / MR_KS1), ks = (int)(f % MR_KS1);
            v = w1[(long long)(32 * t + (l & 31)) * MR_C + 32 * ks + 16 * (l >> 5) + j];
        } else {
            const int g = (int)(f - MR_T1 * MR_KS1);             // 0 .. 575
            const int half = g / 288, st = (g % 288) / 12, k2 = (g % 12) / 6, r = g % 6;
            const int t = 2 * st + k2;
            v = w2[(long long)(192 * half + 32 * r + (l & 31)) * MR_HD + 32 * t + 8 * (j >> 2) + 4 * (l >> 5) + (j & 3)];
        }
        wf[i] = v;
    }
}

// ---- plan: fp32 multipliers of an 8-bit requant, proven -----------------------------------------------------------
// Reference (quant_utils.py:229-231,247-251): q = clamp(rne(fl64(fl64(z * m) * 2^-e)), -128, 127).  Both the reference
// and u(z) = v_cvt_pk_u8_f32(fma(float(z), c32, d)) - 128 are non-decreasing step functions of the integer z, so they
// agree on [-zmax, zmax] iff every step of the reference (255 of them inside the clamp) is a step of u at the same z.
// One block per channel: lane k checks step k for each candidate multiplier {rn(c), rn(c) -+ 1 ulp}; the first candidate
// that passes every step is written to c32[n]; none -> bad |= 1.  zmax = 128 sum_k |W[n,k]| + |bias[n]| must be < 2^24
// (float(z) exact) and c > 0.
__device__ __forceinline__ int mr_ref_q(long long z, double m, double r) {
    double v = __builtin_rint(((double)z * m) * r);
    v = v < -128.0 ? -128.0 : (v > 127.0 ? 127.0 : v);
    return (int)v;
}
__global__ __launch_bounds__(256) void mlpr_rq8_plan_kernel(const int8_t *__restrict__ w, const int32_t *__restrict__ bias,
                                                            const ivit_dyadic *__restrict__ dy, int N, int K, float d,
                                                            float *__restrict__ c32, int *__restrict__ bad) {
    __shared__ int s_l1, s_fail[3];
    const int n = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) s_l1 = 0;
    if (tid < 3) s_fail[tid] = 0;
    __syncthreads();
    int l1 = 0;
    for (int k = tid; k < K; k += 256) l1 += abs((int)w[(long long)n * K + k]);
    atomicAdd(&s_l1, l1);
    __syncthreads();
    const double m = dy[n].m, r = dy[n].r, c = m * r;
    const long long zmax = 128LL * s_l1 + llabs((long long)(bias ? bias[n] : 0));
    if (!(c > 0.0) || zmax >= (1LL << 24) || !(c * (double)zmax < 4194304.0)) {
        if (tid == 0) { c32[n] = (float)c; atomicAdd(bad, 1 << 16); }
        return;
    }
    const float c0 = (float)c;
    float cand[3] = {c0, __uint_as_float(__float_as_uint(c0) - 1u), __uint_as_float(__float_as_uint(c0) + 1u)};      // c0 > 0: the neighbours
    if (tid < 255) {
        const int k = tid - 127;                       // step INTO value k: k = -127 .. 127
        long long z0 = (long long)__builtin_ceil(((double)k - 0.5) / c);
        for (int it = 0; it < 8 && z0 - 1 >= -zmax && mr_ref_q(z0 - 1, m, r) >= k; ++it) --z0;
        for (int it = 0; it < 8 && z0 <= zmax && mr_ref_q(z0, m, r) < k; ++it) ++z0;
        // reference: f(z0 - 1) <= k - 1 < k <= f(z0) (when both sides are inside [-zmax, zmax])
        const bool lo_in = z0 - 1 >= -zmax && z0 - 1 <= zmax, hi_in = z0 >= -zmax && z0 <= zmax;
        const bool ref_ok = (!hi_in || mr_ref_q(z0, m, r) >= k) && (!lo_in || mr_ref_q(z0 - 1, m, r) < k);
#pragma unroll
        for (int ci = 0; ci < 3; ++ci) {
            bool ok = ref_ok;
            if (hi_in) ok = ok && (int)(mr_rq8_u((int)z0, cand[ci], d, 0u, 0) & 255u) - 128 >= k;
            if (lo_in) ok = ok && (int)(mr_rq8_u((int)(z0 - 1), cand[ci], d, 0u, 0) & 255u) - 128 < k;
            if (!ok) atomicOr(&s_fail[ci], 1);
        }
    } else if (tid == 255) {
        // the ends of the range: saturation on both sides
#pragma unroll
        for (int ci = 0; ci < 3; ++ci) {
            const bool ok = (int)(mr_rq8_u((int)zmax, cand[ci], d, 0u, 0) & 255u) - 128 == mr_ref_q(zmax, m, r) &&
                            (int)(mr_rq8_u((int)-zmax, cand[ci], d, 0u, 0) & 255u) - 128 == mr_ref_q(-zmax, m, r);
            if (!ok) atomicOr(&s_fail[ci], 1);
        }
    }
    __syncthreads();
    if (tid == 0) {
        int pick = -1;
        for (int ci = 0; ci < 3; ++ci)
            if (!s_fail[ci]) { pick = ci; break; }
        c32[n] = cand[pick < 0 ? 0 : pick];
        if (pick < 0) atomicAdd(bad, 1);
        else if (pick > 0) atomicAdd(bad + 1, 1);
    }
}

// ---- the kernel ---------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) const unsigned char mr_lds_u8;
typedef __attribute__((address_space(3))) const char mr_lds_c;

__device__ __forceinline__ v16i mr_mfma(v4i a, v4i b, v16i c, int, int, int) {
    if (MR_ABLATE & 4) { c[0] ^= a[0] ^ b[0]; return c; }
    return __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ v4i mr_load16_async(const void *ptr) {
    v4i v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(ptr) : "memory");
    return v;
}
// the same into the accumulator half of the register file (an MFMA reads its B operand from there as well)
__device__ __forceinline__ v4i mr_load16_async_a(const void *ptr) {
    v4i v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(v) : "v"(ptr) : "memory");
    return v;
}
__device__ __forceinline__ v2i mr_load8_async(const void *ptr) {
    v2i v;
    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v) : "v"(ptr) : "memory");
    return v;
}
#define MR_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
// a use of the youngest LDS result of a section, at the section's END: the compiler's wait (always a full lgkmcnt(0) drain
// while an LDS-DMA is pending) lands here, behind the section's MFMAs, instead of in front of the next section's first MFMA —
// i.e. behind the NEXT batch of reads, whose whole latency it would then expose
#define MR_DRAIN(x) do { __builtin_amdgcn_sched_barrier(0); asm volatile("" :: "v"(x)); __builtin_amdgcn_sched_barrier(0); } while (0)
// the same for six weight fragments that should live in the accumulator half of the register file: the tie to an "a"
// register is the use (placed right behind the loads it would make them wait at once)
#define MR_DRAIN_W_A(w) do { __builtin_amdgcn_sched_barrier(0); asm volatile("" : "+a"(w[0]), "+a"(w[1]), "+a"(w[2]), "+a"(w[3]), "+a"(w[4]), "+a"(w[5])); __builtin_amdgcn_sched_barrier(0); } while (0)
#define MR_WAIT_VM_LGKM(n) asm volatile("s_waitcnt vmcnt(" #n ") lgkmcnt(0)" ::: "memory")

// Stage anatomy.  hipcc (ROCm 7.2) turns every LDS wait into lgkmcnt(0) while an LDS-DMA is pending — here: always —
// so a wait costs the latency of the YOUNGEST read in flight.  A stage is therefore two sections; a section first issues
// every LDS read the NEXT section consumes (6 weight fragments, the tile's constants, ShiftGELU gathers), then computes on
// what the previous section fetched: one drain per section, ~6 MFMAs after its youngest read.  The ring's barrier sits
// between the two sections of stage g and publishes stage g + 1, whose first fragments section B reads.
__global__ __launch_bounds__(MR_THREADS, 1) void mlpr_kernel(MlprArgs p) {
    extern __shared__ __attribute__((aligned(256))) char sm[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const long long tb = p.M * blockIdx.x / gridDim.x, te = p.M * (blockIdx.x + 1) / gridDim.x;
    const int cnt = (int)(te - tb);
    if (cnt <= 0) return;
    const int np = (cnt + MR_WAVES * MR_TOK - 1) / (MR_WAVES * MR_TOK);

    // ---- per-layer constants -> LDS (before the first DMA: plain loads here do not meet the ring)
    for (int i = tid; i < MR_HD / 4; i += MR_THREADS) {
        reinterpret_cast<v4f *>(sm + MR_C1F)[i] = reinterpret_cast<const v4f *>(p.c1f)[i];
        reinterpret_cast<v4i *>(sm + MR_B1)[i] = reinterpret_cast<const v4i *>(p.b1)[i];
    }
    for (int i = tid; i < MR_C / 2; i += MR_THREADS)
        reinterpret_cast<v4i *>(sm + MR_CQ2)[i] = reinterpret_cast<const v4i *>(p.cq2)[i];
    for (int i = tid; i < MR_C / 4; i += MR_THREADS)
        reinterpret_cast<v4i *>(sm + MR_B2)[i] = reinterpret_cast<const v4i *>(p.b2)[i];
    __syncthreads();

    // ---- weight ring: stage s of the endless stream = stage s % 96 of the fragment array, slot s % 8
    const int8_t *wsrc0 = p.wf + wave * 3072;
    auto dma_at = [&](auto sc, const int8_t *wsrc, unsigned dma_voff) __attribute__((always_inline)) {
        constexpr int S = decltype(sc)::value % MR_NSTAGES;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int8_t *sb = wsrc + ((size_t)S * MR_STAGE + i * 1024);
            asm volatile("" : "+s"(sb));          // opaque: otherwise (base + lane offset) + constant is kept per DMA, in registers, forever
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(sb + (size_t)dma_voff),
                                             (__attribute__((address_space(3))) void *)(sm + MR_RING + (S % MR_NSTG) * MR_STAGE + wave * 3072 + i * 1024),
                                             16, 0, 0);
        }
    };
    mr_for<0, MR_DIST + 1>([&](auto sc) __attribute__((always_inline)) { dma_at(sc, wsrc0, (unsigned)(tid & 63) * 16u); });

    // tokens of pass ps: the workgroup's range cut evenly into np passes, a pass cut evenly into four waves
    int my0 = 0, nvalid = 0;
    const int pbase = (int)((unsigned)cnt / (unsigned)np), prem = cnt - pbase * np;      // no division inside the pass loop
    auto pass_tokens = [&](int ps) __attribute__((always_inline)) {
        const int p0 = ps * pbase + min(ps, prem), len = pbase + (ps < prem ? 1 : 0);
        const int q = (len + MR_WAVES - 1) >> 2;
        my0 = p0 + wave * q;
        nvalid = max(0, min(q, p0 + len - my0));
        if (nvalid == 0) my0 = p0;
    };
    v4i xf[MR_KS1];
    auto x_fetch = [&](int n, int h) __attribute__((always_inline)) {
        const long long tok = tb + my0 + min(n, max(nvalid - 1, 0));
        const int8_t *xp = p.x + tok * MR_C + 16 * h;
#pragma unroll
        for (int ks = 0; ks < MR_KS1; ++ks) xf[ks] = mr_load16_async_a(xp + 32 * ks);
    };
    pass_tokens(0);
    x_fetch(tid & 31, (tid >> 5) & 1);

    const unsigned sm_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char *)sm;
    int tr_pass = 0;
    // trace builds: every lane of a wave writes the same stamp to the same LDS word (no branch, no vector-memory traffic
    // between the counted waits); workgroup 0 copies the stamps out at the end
    auto stamp_stage = [&](int stage) __attribute__((always_inline)) {
        if (MR_TRACE)
            *reinterpret_cast<unsigned long long *>(sm + MR_SMEM + ((min(tr_pass, 2) * MR_WAVES + wave) * 128 + stage) * 8) = __builtin_readcyclecounter();
    };
    auto stamp = [&](int pt) __attribute__((always_inline)) { stamp_stage(96 + pt); };

    // the first stage's first fragments and bias: what "section B of stage -1" would have fetched
    v4i wA[6], wB[6];
    v16i bias;
    MR_WAIT_VM(0);                                     // stages 0..6 and the first activations have landed (mine)
    __builtin_amdgcn_s_barrier();                      // (everybody's)
    {
        const int lane = tid & 63, h = lane >> 5;
#pragma unroll
        for (int j = 0; j < 6; ++j) wA[j] = *reinterpret_cast<const v4i *>(sm + MR_RING + j * 1024 + lane * 16);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const v4i bb = *reinterpret_cast<const v4i *>(sm + MR_B1 + 16 * h + 32 * b);
#pragma unroll
            for (int e = 0; e < 4; ++e) bias[4 * b + e] = bb[e];
        }
    }

    for (int ps = 0; ps < np; ++ps) {
        // per-lane indices and the stream base from opaque copies, once per pass: left visible, every address of the
        // unrolled pass (288 DMA sources, ~200 LDS addresses) is loop-invariant, gets hoisted and lives in scratch
        int tid_l = threadIdx.x;
        asm volatile("" : "+v"(tid_l));
        const int lane = tid_l & 63, n = lane & 31, h = lane >> 5;
        const int8_t *wsrc = wsrc0;
        asm volatile("" : "+s"(wsrc));
        const unsigned dma_voff = (unsigned)lane * 16u;
        const unsigned skew = (unsigned)(n * 8) & 0xf8u;
        const unsigned lineaddr = sm_lds + MR_TAB + wave * (MR_TOK * 256) + n * 256;
        const unsigned gbase = lineaddr | skew;                     // gather address of biased byte u: gbase ^ u
        // this lane's 4 channels of a tile's constants: + 128 tile + 32 b (4-byte tables), + 256 tile + 64 b (fc2's doubles);
        // opaque, so that the tile terms stay immediates of the ds_read instead of becoming one address register each
        mr_lds_c *cbase = (mr_lds_c *)sm + MR_C1F + 16 * h, *cbase2 = (mr_lds_c *)sm + MR_CQ2 + 32 * h;
        mr_lds_c *wbase = (mr_lds_c *)sm + MR_RING + lane * 16;
        asm volatile("" : "+v"(cbase), "+v"(cbase2), "+v"(wbase));
        typedef __attribute__((address_space(3))) const v4i lds_v4i;
        typedef __attribute__((address_space(3))) const v4f lds_v4f;
        auto dma = [&](auto sc) __attribute__((always_inline)) { dma_at(sc, wsrc, dma_voff); };
        // fragments j0 .. j0 + 5 of stream stage S
        // fc1's fragments (consumed in stages < 48) go to the accumulator half of the register file, which fc1 leaves
        // half empty; fc2's stay in VGPRs (its 192 accumulators and the parked hidden tiles own the other half)
        auto read_w = [&](v4i (&w)[6], auto sc, auto jc) __attribute__((always_inline)) {
            constexpr int S = decltype(sc)::value % MR_NSTAGES, slot = S % MR_NSTG, j0 = decltype(jc)::value;
#pragma unroll
            for (int j = 0; j < 6; ++j)
                if (!(MR_ABLATE & 8)) {
                    w[j] = *reinterpret_cast<lds_v4i *>(wbase + slot * MR_STAGE + (j0 + j) * 1024);
                }
        };
        // the ring step between the two sections of stream stage S: stage S + 1 is complete for everybody afterwards
        auto ring = [&](auto sc) __attribute__((always_inline)) {
            constexpr int S = decltype(sc)::value;
            __builtin_amdgcn_sched_barrier(0);
            stamp_stage(S);
            // its DMA was issued six ring steps ago; 15 younger DMAs.  Steps 0..5 of a pass: everything that old was already
            // waited for (epilogue / prologue), and the epilogue's stores must not be waited for
            if (S < MR_DIST || (S >= MR_T1 + 24 && S < MR_T1 + 24 + MR_DIST)) MR_WAIT_VM(63); else MR_WAIT_VM(15);
            if (!(MR_ABLATE & 32)) __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (!(MR_ABLATE & 16)) dma(std::integral_constant<int, S + 1 + MR_DIST>{});
            __builtin_amdgcn_sched_barrier(0);
        };

        // the hidden rows: u = q + 128, 4 consecutive channels per register.  Tiles < MR_HV in VGPRs, the rest parked in
        // AGPRs through empty asm statements with an "a" constraint (the copies are the compiler's own v_accvgpr_write /
        // _read): left to itself the register allocator spills VGPRs to AGPRs only where an AGPR is free over the WHOLE
        // function — fc2's 192 accumulators leave none — and sends the hidden rows to scratch
        unsigned Hv[MR_HV][4], Ha[MR_T1 - MR_HV][4];
        auto set_H = [&](auto tc, auto bc, unsigned w) __attribute__((always_inline)) {
            constexpr int T = decltype(tc)::value, B = decltype(bc)::value;
            if constexpr (T < MR_HV) Hv[T][B] = w;
            else { unsigned a; asm("" : "=a"(a) : "0"(w)); Ha[T - MR_HV][B] = a; }
        };
        auto get_H = [&](auto tc, auto bc) __attribute__((always_inline)) -> unsigned {
            constexpr int T = decltype(tc)::value, B = decltype(bc)::value;
            if constexpr (T < MR_HV) return Hv[T][B];
            else { const unsigned a = Ha[T - MR_HV][B]; unsigned x; asm("" : "=v"(x) : "0"(a)); return x; }
        };
        stamp(0);

        // ================= fc1: stage S = hidden tile S; the requant of tile S - 1 runs beside its MFMAs
        v16i acc[2];
        v4f cprev[4], cnext[4];
        float mx = 0.0f;
        auto fc1_epi = [&](auto tc, auto bc) __attribute__((always_inline)) {
            constexpr int T = decltype(tc)::value, B = decltype(bc)::value;
            if (MR_ABLATE & 1) { unsigned w0 = (unsigned)acc[T & 1][4 * B]; asm volatile("" : "+v"(w0)); set_H(tc, bc, w0); return; }
            unsigned w = 0;
            float pf[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                pf[e] = __builtin_fmaf((float)acc[T & 1][4 * B + e], cprev[B][e], p.d1);
                w = __builtin_amdgcn_cvt_pk_u8_f32(pf[e], (unsigned)e, w);
            }
            asm("v_max3_f32 %0, %0, %1, %2" : "+v"(mx) : "v"(pf[0]), "v"(pf[1]));
            asm("v_max3_f32 %0, %0, %1, %2" : "+v"(mx) : "v"(pf[2]), "v"(pf[3]));
            asm volatile("" : "+v"(w));           // here, not where the first reader is: the four products would wait in scratch
            set_H(tc, bc, w);
        };
        mr_for<0, MR_T1>([&](auto sc) __attribute__((always_inline)) {
            constexpr int S = decltype(sc)::value;
            // ---- section A
            __builtin_amdgcn_sched_barrier(0);
            read_w(wB, sc, std::integral_constant<int, 6>{});
            __builtin_amdgcn_sched_barrier(0);
            mr_for<0, 6>([&](auto kc) __attribute__((always_inline)) {
                constexpr int ks = decltype(kc)::value;
                acc[S & 1] = mr_mfma(wA[ks], xf[ks], ks == 0 ? bias : acc[S & 1], 0, 0, 0);
                if constexpr (S > 0 && (ks == 1 || ks == 4)) fc1_epi(std::integral_constant<int, S - 1>{}, std::integral_constant<int, ks / 3>{});
            });
            MR_DRAIN_W_A(wB);
            ring(sc);
            // ---- section B: the next stage's first fragments (stage 48 is fc2's first k-step), this tile's multipliers, the next bias
            if constexpr (S + 1 < MR_T1) {
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const v4i bb = *reinterpret_cast<lds_v4i *>(cbase + (MR_B1 - MR_C1F) + 128 * (S + 1) + 32 * b);
#pragma unroll
                    for (int e = 0; e < 4; ++e) bias[4 * b + e] = bb[e];
                }
            }
            read_w(wA, std::integral_constant<int, S + 1>{}, std::integral_constant<int, 0>{});
#pragma unroll
            for (int b = 0; b < 4; ++b) cnext[b] = *reinterpret_cast<lds_v4f *>(cbase + 128 * S + 32 * b);
            __builtin_amdgcn_sched_barrier(0);
            mr_for<6, 12>([&](auto kc) __attribute__((always_inline)) {
                constexpr int ks = decltype(kc)::value;
                acc[S & 1] = mr_mfma(wB[ks - 6], xf[ks], acc[S & 1], 0, 0, 0);
                if constexpr (S > 0 && (ks == 7 || ks == 10)) fc1_epi(std::integral_constant<int, S - 1>{}, std::integral_constant<int, 2 + (ks - 6) / 3>{});
            });
            if constexpr (S + 1 < MR_T1) MR_DRAIN_W_A(wA);
            MR_DRAIN(cnext[3]);
#pragma unroll
            for (int b = 0; b < 4; ++b) cprev[b] = cnext[b];
        });
        __builtin_amdgcn_sched_barrier(0);
        mr_for<0, 4>([&](auto bc) __attribute__((always_inline)) { fc1_epi(std::integral_constant<int, MR_T1 - 1>{}, bc); });
        stamp(1);

        // ================= row maximum -> table line of the token -> LDS (XOR-skewed)
        {
            const float mo = __shfl_xor(mx, 32);
            const unsigned um = __builtin_amdgcn_cvt_pk_u8_f32(mx > mo ? mx : mo, 0u, 0u) & 255u;
            const int8_t *tp = p.tab + (size_t)um * 256 + 128 * h;
            v4i tl[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) tl[j] = mr_load16_async(tp + 16 * j);
            MR_WAIT_VM(0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned o = (unsigned)(128 * h + 16 * j);
                *reinterpret_cast<v2i *>(sm + MR_TAB + wave * (MR_TOK * 256) + n * 256 + (o ^ skew)) = v2i{tl[j][0], tl[j][1]};
                *reinterpret_cast<v2i *>(sm + MR_TAB + wave * (MR_TOK * 256) + n * 256 + ((o + 8) ^ skew)) = v2i{tl[j][2], tl[j][3]};
            }
        }
        stamp(2);

        // ================= fc2 in two output halves (96 accumulators at a time: with all 192 live beside the 192 registers
        // of hidden rows the allocator overflows into scratch, and every scratch reload drains the DMA ring).  Stage
        // 48 + 24 half + i = k-steps 2i (section A) and 2i + 1 (section B) for six row tiles.  Half 0 applies ShiftGELU on the
        // fly — gathers of tile T + 2 issued, tile T + 1 combined and written back over the hidden tile, during k-step T —
        // half 1 reads the rewritten tiles.
        typedef double v2d __attribute__((ext_vector_type(2)));
        typedef __attribute__((address_space(3))) const v2d lds_v2d;
        v4i hg[2];
        unsigned gq[16];
        auto gelu_issue = [&](auto tc) __attribute__((always_inline)) {
            constexpr int T = decltype(tc)::value;
            if constexpr (T < MR_T1) {
                mr_for<0, 4>([&](auto dc) __attribute__((always_inline)) {
                    constexpr int D = decltype(dc)::value;
                    const unsigned x = get_H(tc, dc);
                    if (MR_ABLATE & 2) { gq[4 * D + 0] = x & 255u; gq[4 * D + 1] = (x >> 8) & 255u; gq[4 * D + 2] = (x >> 16) & 255u; gq[4 * D + 3] = x >> 24; return; }
                    gq[4 * D + 0] = *(mr_lds_u8 *)(size_t)(gbase ^ (x & 0xffu));
                    gq[4 * D + 1] = *(mr_lds_u8 *)(size_t)(gbase ^ ((x >> 8) & 0xffu));
                    gq[4 * D + 2] = *(mr_lds_u8 *)(size_t)(gbase ^ ((x >> 16) & 0xffu));
                    gq[4 * D + 3] = *(mr_lds_u8 *)(size_t)(gbase ^ (x >> 24));
                });
            }
        };
        auto gelu_combine = [&](auto tc) __attribute__((always_inline)) {
            constexpr int T = decltype(tc)::value;
            if constexpr (T < MR_T1) {
                mr_for<0, 4>([&](auto dc) __attribute__((always_inline)) {
                    constexpr int D = decltype(dc)::value;
                    const unsigned g = gq[4 * D] | (gq[4 * D + 1] << 8) | (gq[4 * D + 2] << 16) | (gq[4 * D + 3] << 24);
                    hg[T & 1][D] = (int)g;
                    set_H(tc, dc, g);
                });
            }
        };
        auto tile_of = [&](auto tc) __attribute__((always_inline)) -> v4i {
            v4i t;
            mr_for<0, 4>([&](auto dc) __attribute__((always_inline)) { t[decltype(dc)::value] = (int)get_H(tc, dc); });
            return t;
        };
        // lanes beyond the wave's tokens carry copies of its last token (a wave without tokens: of the pass's first one) and
        // store the same values to the same place: no predicate, no divergence
        const long long tok_out = tb + my0 + min(n, max(nvalid - 1, 0));
        v16i acc2[6];
        // qact2 (16 bit) + qact4 with the identity branch for one half (six row tiles = 24 groups of 4 channels)
        auto epilogue = [&](auto hc) __attribute__((always_inline)) {
            constexpr int HALF = decltype(hc)::value;
            __builtin_amdgcn_sched_barrier(0);
            v2i rs[24];
            const int16_t *rp = p.residual + tok_out * MR_C + 192 * HALF + 4 * h;
#pragma unroll
            for (int g = 0; g < 24; ++g) rs[g] = mr_load8_async(rp + 8 * g);
            if constexpr (HALF == 1) {
                // no branch (after the last pass the first pass's rows are fetched again, for nobody): a block boundary in front
                // of the epilogue makes the register allocator copy all the accumulators to VGPRs at the block's entry
                pass_tokens(ps + 1 < np ? ps + 1 : 0);
                x_fetch(n, h);
                MR_WAIT_VM(12);                       // the 12 activation loads are younger than the identity rows
            } else {
                MR_WAIT_VM(0);
            }
            __builtin_amdgcn_sched_barrier(0);
            v2d c2q[2][2];                            // multipliers of group g + 1 travel while group g is requantised
            c2q[0][0] = *reinterpret_cast<lds_v2d *>(cbase2 + 1536 * HALF);
            c2q[0][1] = *reinterpret_cast<lds_v2d *>(cbase2 + 1536 * HALF + 16);
            mr_for<0, 24>([&](auto gc) __attribute__((always_inline)) {
                constexpr int G = decltype(gc)::value, r = G >> 2, b = G & 3;
                if constexpr ((G & 1) == 0) __builtin_amdgcn_sched_barrier(0);
                if constexpr (G + 1 < 24) {
                    c2q[(G + 1) & 1][0] = *reinterpret_cast<lds_v2d *>(cbase2 + 1536 * HALF + 64 * (G + 1));
                    c2q[(G + 1) & 1][1] = *reinterpret_cast<lds_v2d *>(cbase2 + 1536 * HALF + 64 * (G + 1) + 16);
                }
                int o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const double c = c2q[G & 1][e >> 1][e & 1];
                    const double t64 = (double)acc2[r][4 * b + e] * c + MR_MAGIC;
                    const int t = min(max(__double2loint(t64), -32768), 32767);
                    const int rr = (int)(short)((unsigned)rs[G][e >> 1] >> (16 * (e & 1)));
                    o[e] = min(max(rq_fast(rr, p.cr) + rq_fast(t, p.cm), -32768), 32767);
                }
                *reinterpret_cast<v2i *>(p.out + tok_out * MR_C + 192 * HALF + 8 * G + 4 * h) =
                    v2i{(int)__builtin_amdgcn_perm((unsigned)o[1], (unsigned)o[0], 0x05040100u),
                        (int)__builtin_amdgcn_perm((unsigned)o[3], (unsigned)o[2], 0x05040100u)};
            });
            __builtin_amdgcn_sched_barrier(0);
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the table lines are in LDS (wave-private: no barrier)
        __builtin_amdgcn_sched_barrier(0);
        gelu_issue(I0{});
        MR_DRAIN(gq[15]);
        gelu_combine(I0{});
        gelu_issue(I1{});
        mr_for<0, 2>([&](auto hc) __attribute__((always_inline)) {
            constexpr int HALF = decltype(hc)::value;
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const v4i bb = *reinterpret_cast<lds_v4i *>(cbase + (MR_B2 - MR_C1F) + 768 * HALF + 128 * r + 32 * b);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc2[r][4 * b + e] = bb[e];
                }
            mr_for<0, 24>([&](auto ic) __attribute__((always_inline)) {
                constexpr int I = decltype(ic)::value, S = MR_T1 + 24 * HALF + I, TA = 2 * I, TB = 2 * I + 1;
                using SC = std::integral_constant<int, S>;
                // ---- section A: k-step TA
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (HALF == 0) gelu_combine(std::integral_constant<int, TA + 1>{});
                __builtin_amdgcn_sched_barrier(0);
                read_w(wB, SC{}, std::integral_constant<int, 6>{});
                if constexpr (HALF == 0) gelu_issue(std::integral_constant<int, TA + 2>{});
                __builtin_amdgcn_sched_barrier(0);
                {
                    const v4i bop = HALF == 0 ? hg[TA & 1] : tile_of(std::integral_constant<int, TA>{});
                    mr_for<0, 6>([&](auto rc) __attribute__((always_inline)) {
                        constexpr int r = decltype(rc)::value;
                        acc2[r] = mr_mfma(wA[r], bop, acc2[r], 0, 0, 0);
                    });
                }
                MR_DRAIN(wB[5]);
                if constexpr (HALF == 0 && TA + 2 < MR_T1) MR_DRAIN(gq[15]);
                ring(SC{});
                // ---- section B: k-step TB
                if constexpr (HALF == 0) gelu_combine(std::integral_constant<int, TB + 1>{});
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (S + 1 == MR_NSTAGES) {           // the next pass's first bias
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const v4i bb = *reinterpret_cast<lds_v4i *>(cbase + (MR_B1 - MR_C1F) + 32 * b);
#pragma unroll
                        for (int e = 0; e < 4; ++e) bias[4 * b + e] = bb[e];
                    }
                }
                read_w(wA, std::integral_constant<int, S + 1>{}, std::integral_constant<int, 0>{});
                if constexpr (HALF == 0) gelu_issue(std::integral_constant<int, TB + 2>{});
                __builtin_amdgcn_sched_barrier(0);
                {
                    const v4i bop = HALF == 0 ? hg[TB & 1] : tile_of(std::integral_constant<int, TB>{});
                    mr_for<0, 6>([&](auto rc) __attribute__((always_inline)) {
                        constexpr int r = decltype(rc)::value;
                        acc2[r] = mr_mfma(wB[r], bop, acc2[r], 0, 0, 0);
                    });
                }
                if constexpr (S + 1 < MR_NSTAGES || true) MR_DRAIN(wA[5]);
                if constexpr (HALF == 0 && TB + 2 < MR_T1) MR_DRAIN(gq[15]);
            });
            stamp(3 + HALF);
            epilogue(hc);
        });
        __builtin_amdgcn_sched_barrier(0);
        MR_WAIT_VM(24);                               // the next pass's activations (older than the last 24 stores) are in
        stamp(5);
        ++tr_pass;
    }
    MR_WAIT_VM(0);
    if (MR_TRACE) {
        __syncthreads();
        if (blockIdx.x == 0)
            for (int i = tid; i < 2 * MR_WAVES * 128; i += MR_THREADS) p.trace[i] = reinterpret_cast<const unsigned long long *>(sm + MR_SMEM)[i];
    }
}
