// ivit_mlp.h — Mlp.forward + the block's residual QuantAct as ONE kernel for the D = 384 models (DeiT-S, Swin stage 2):
//   fc1 -> qact_gelu (8 bit) -> ShiftGELU -> qact1 (8 bit) -> fc2 -> qact2 (16 bit) -> qact4(+identity) (16 bit)
// (models/layers_quant.py:144-153, then vit_quant.py:141-142 / swin_quant.py:296-300).  The 4 D-wide hidden tensor never
// exists in HBM: per 64-token unit it is produced into LDS by fc1, rewritten in place by the ShiftGELU table and consumed
// from LDS by fc2.  Unfused, the hidden tensor crosses HBM four times (fc1 write, GELU read + write, fc2 read: 310 MB of
// the layer's 775 MB at batch 256) and ShiftGELU is a launch of its own.
//
// Why 64 tokens and why it pays although each unit re-reads both weight matrices (1.18 MB) from L2: ShiftGELU's row
// maximum couples all 1536 hidden channels of a token, so a unit must own whole hidden rows; 64 rows x 1536 B = 96 KB is
// what the LDS holds next to the activation tile.  The weights stream L2 -> registers (never through LDS): they are laid
// out at plan time in MFMA-fragment order (1 KB per (16-channel tile, 64-column step), lane-linear), every wave reads only
// the fragments of ITS output channels, and a fragment feeds four MFMAs (four 16-token tiles).  No barrier inside the two
// GEMM phases: the eight waves drift apart, one wave's requant epilogue runs beside its SIMD mate's MFMAs.
//
// Shapes: v_mfma_i32_16x16x64_i8, weights as the A operand (rows = channels), activations as B (columns = tokens): a lane
// holds 4 consecutive channels of one token per accumulator, which pack into one dword.  fc1: wave w owns hidden channels
// [192 w, 192 w + 192) in four chunks of 48 (3 channel tiles x 4 token tiles = 48 accumulator registers); fc2: wave w owns
// output channels [48 w, 48 w + 48).  288 MFMAs per wave and phase.
//
// LDS images are K-blocked, [64-column block][64 tokens][64 B], with the four 16-byte chunks of a token's 64 B permuted by
// phi(token, chunk) = swapbits(chunk) ^ ((token >> 3) & 1): a ds_read_b128 is served in four groups of 16 lanes whose
// (token, chunk) sets are {0-3, 12-15} x {c} with {4-11} x {c + 1} (and the mirror image) — a row-major image with any
// padded stride has a 2-way conflict in every group; this permutation has none and needs no padding.
#pragma once
#include <type_traits>
#include "ivit_device.h"

#define MLP_C 384
#define MLP_HD 1536
#define MLP_TT 5                               // token tiles (of 16) a unit may have: 4 or 5
#ifndef MLP_WAVES
#define MLP_WAVES 8
#endif
#define MLP_NJ (MLP_C / 16 / MLP_WAVES)        // channel tiles per wave and step: 2 (12 waves) or 3 (8 waves)
//                         // three per SIMD: a lone wave issues a 16x16x64 MFMA every ~34 cycles, the pipe takes one per ~17
#define MLP_THREADS (MLP_WAVES * 64)
#define MLP_KS1 (MLP_C / 64)                  // 6 column steps of fc1
#define MLP_KS2 (MLP_HD / 64)                 // 24 column steps of fc2
#define MLP_KBLK (MLP_TT * 16 * 64)            // one 64-column block of an LDS image: [80 tokens][64 B]
#define MLP_SH 0                              // hidden tile [24][80][64 B]
#define MLP_SA (MLP_KS2 * MLP_KBLK)           // activation tile [6][80][64 B]
#define MLP_STAB (MLP_SA + MLP_KS1 * MLP_KBLK)    // one ShiftGELU table line (256 B) per half-wave
#define MLP_SMEM (MLP_STAB + 2 * MLP_WAVES * 256)
#define MLP_MAGIC 6755399441055744.0
#ifndef MLP_PRIO_YOUNG
#define MLP_PRIO_YOUNG 0
#endif
#ifndef MLP_FC2_SYNC
#define MLP_FC2_SYNC 0                        // raw s_barrier every n k-steps of the fc2 K loop (0: none)
#endif
#ifndef MLP_WD
#define MLP_WD 3                              // weight fragments in flight ahead of the MFMAs that consume them
#endif
// timeline instrumentation (tools/ubench/mlp_probe.hip, -DMLP_TRACE=1): every wave of workgroup 0 stamps s_memtime at the
// phase boundaries of its first units into p.trace[(unit_index * 8 + wave) * 8 + point]
#ifndef MLP_TRACE
#define MLP_TRACE 0
#endif
// timing ablations (probe builds only; results invalid): 1 = no weight loads after the prologue, 2 = no activation-fragment
// LDS reads after the prologue, 4 = no requant epilogue arithmetic
#ifndef MLP_ABLATE
#define MLP_ABLATE 0
#endif

struct MlpArgs {
    const int8_t *x;          // [M, 384] int8 (LayerNorm + requant output)
    const v4i *w1f, *w2f;     // fragment-ordered weights (mlp_swizzle_kernel)
    const int32_t *b1, *b2;   // biases (never null: the plans' bias_eff)
    const double *cq1, *cq2;  // per-channel c = m * 2^-e
    const int8_t *tab;        // ShiftGELU(+requant) table [256 maxima][256 values]
    const int16_t *residual;  // [M, 384] identity branch
    int16_t *out;             // [M, 384]
    double cm, cr;            // qact4: main and identity multipliers
    long long M;
    int balanced;             // unit schedule: 0 = 64-token units dealt round-robin, 1 = contiguous tile ranges cut into units of <= 5 tiles
    unsigned long long *trace;   // MLP_TRACE builds only
    // mlp384rs_kernel<FMA, LNH = true> (ivit_layernorm_mlp_fused_planned): norm2 + qact3 of this workgroup's rows first, from the block's 16-bit
    // stream (`residual` is that stream), into x (a scratch of M x 384 bytes that only this launch reads)
    float ln_s;
    const float *ln_bias_int, *ln_sc;
    const ivit_dyadic *ln_dy;
};

__device__ __forceinline__ int mlp_phi(int tok, int chunk) {
    return (((chunk & 1) << 1) | (chunk >> 1)) ^ ((tok >> 3) & 1);
}

// weights [N][K] int8 -> fragments of 64 lanes x 16 B, lane l = W[ct*16 + (l & 15)][ks*64 + (l >> 4)*16 ...], in the order the
// kernel consumes them: fragment index f = step * 24 + wave * 2 + j, where step = chunk * (K / 64) + ks walks the wave's
// chunks of two channel tiles (ct = wave * T + chunk * 2 + j, T = N / 16 / MLP_WAVES tiles per wave) and the 64-column
// steps inside a chunk.  What the twelve waves of a workgroup request in one step is ONE contiguous 24 KB window: the
// requests spread over all L2 channels.  (With each wave's fragments contiguous instead — 24 streams a multiple of 4 KB
// apart advancing in lock-step — the fc2 weight stream ran at half the rate of the fc1 one: +9.5k cycles per unit.)
__global__ __launch_bounds__(256) void mlp_swizzle_kernel(const int8_t *__restrict__ w, int N, int K, v4i *__restrict__ wf) {
    const int nks = K >> 6, T = (N >> 4) / MLP_WAVES;
    const long long total = (long long)(N >> 4) * nks * 64;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int l = (int)(i & 63);
        const int f = (int)(i >> 6), step = f / (MLP_NJ * MLP_WAVES), r = f - step * (MLP_NJ * MLP_WAVES);
        const int chunk = step / nks, ks = step - chunk * nks, ct = (r / MLP_NJ) * T + chunk * MLP_NJ + (r % MLP_NJ);
        wf[i] = *reinterpret_cast<const v4i *>(w + (long long)(ct * 16 + (l & 15)) * K + ks * 64 + (l >> 4) * 16);
    }
}

template <bool FMA>
__device__ __forceinline__ int mlp_rq(int z, double c) {
    const double t = FMA ? __builtin_fma((double)z, c, MLP_MAGIC) : ((double)z * c + MLP_MAGIC);
    return __double2loint(t);
}

// FMA: both plans prove |z * m| < 2^53 (one fused rounding == the reference's two), else multiply and add separately.
// Both plans prove |z * c| < 2^31 (the host refuses the kernel otherwise); |cm|, |cr| < 2^9 (host-checked) for rq_fast.
//
// Units and balance.  The token axis is cut into tiles of 16.  A unit costs one pass over both weight matrices whatever its
// size (measured, one unit per CU: 29.8 / 31.9 / 34.2 / 37.2 / 44.3 us for 1..5 tiles), so units are as large as the LDS
// allows and as few as possible.  Two schedules, chosen by the host: 64-token units dealt round-robin, or — when that
// needs one more round than the work — workgroup b owns the contiguous tile range [T b / G, T (b + 1) / G) and walks it in
// units of <= 5 tiles (MLP_TT = 5: hidden 120 KB + activations 30 KB + table lines 4 KB of LDS): DeiT-S at batch 256 is
// 3152 tiles on 256 CUs = 12.3 per CU, three units of (5,) 4, 4 tiles instead of 3.08 -> 4 rounds of 64-token units.
// The unit body is instantiated for 4 and for 5 tiles (a unit with fewer tiles runs the 4-tile body on clamped rows).
//
// Software pipeline of both GEMM phases (pinned with scheduling fences: left alone the scheduler sinks every load to just
// before its first use and each step waits out a full LDS / L2 latency with the matrix pipe idle — measured 2.5-3.5x the
// MFMA time; hoisted to the top of the unrolled phase they are all live at once and spill): step s issues the weight
// fragments of step s + WD and the activation fragments of step s + 1, then its own MFMAs.
template <bool FMA>
__global__ __launch_bounds__(MLP_THREADS, MLP_WAVES / 4) void mlp384_kernel(MlpArgs p) {
    extern __shared__ __attribute__((aligned(256))) char sm[];
    constexpr int NJ = MLP_NJ;                        // channel tiles per step
    constexpr int CT1 = MLP_HD / 16 / MLP_WAVES;      // channel tiles of fc1 per wave, in chunks of NJ
    constexpr int NCH = CT1 / NJ, NS1 = NCH * MLP_KS1, WD = MLP_WD;   // fc1 chunks, fc1 steps, weight prefetch distance
    constexpr int AREG = (MLP_TT * 16 * 24 + MLP_THREADS - 1) / MLP_THREADS;
    static_assert(NJ * 16 * MLP_WAVES == MLP_C && CT1 % NJ == 0, "wave count must split 96 / 24 channel tiles evenly");
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (MLP_PRIO_YOUNG && wave >= MLP_WAVES / 2) __builtin_amdgcn_s_setprio(1);      // probe: static priority for the younger half
    typedef double v2d __attribute__((ext_vector_type(2)));

    // ---- this workgroup's units: (first tile, tiles) of unit i
    const long long ntiles = (p.M + 15) >> 4;
    const long long t_beg = ntiles * blockIdx.x / gridDim.x, t_end = ntiles * (blockIdx.x + 1) / gridDim.x;
    const int n_own = (int)(t_end - t_beg);
    const long long nfix = (ntiles + MLP_TT - 2) / (MLP_TT - 1);                 // 64-token units
    const int nu = p.balanced ? (n_own + MLP_TT - 1) / MLP_TT
                              : (int)((nfix - (long long)blockIdx.x + gridDim.x - 1) / gridDim.x);
    if (nu <= 0) return;
    auto unit_tile0 = [&](int i) -> long long {
        if (p.balanced) return t_beg + (long long)n_own * i / nu;
        return min(((long long)blockIdx.x + (long long)i * gridDim.x) * (MLP_TT - 1), ntiles);
    };
    auto unit_ntt = [&](int i) -> int {
        if (i >= nu) return 0;
        if (p.balanced) return (int)(unit_tile0(i + 1) - unit_tile0(i));
        const long long t0 = unit_tile0(i);
        return (int)min((long long)(MLP_TT - 1), ntiles - t0);
    };

    int tr_unit = 0;
    auto stamp = [&](int pt) __attribute__((always_inline)) {
        if (MLP_TRACE) {
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");       // the phase's own loads and stores count
            if (blockIdx.x == 0 && tr_unit < 4 && (threadIdx.x & 63) == 0)
                p.trace[(tr_unit * MLP_WAVES + wave) * 8 + pt] = __builtin_readcyclecounter();
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // activation tile of a unit (rows x 24 chunks of 16 B): global -> registers (a_fetch), registers -> LDS (a_commit)
    v4i areg[AREG];
    auto a_fetch = [&](long long tile0, int ntt) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < AREG; ++i) {
            const int ch = (int)threadIdx.x + i * MLP_THREADS, row = ch / 24, c16 = ch - row * 24;
            if (ch < ntt * 16 * 24) {
                const long long grow = min(tile0 * 16 + row, p.M - 1);
                areg[i] = *reinterpret_cast<const v4i *>(p.x + grow * MLP_C + c16 * 16);
            }
        }
    };
    auto a_commit = [&](int ntt) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < AREG; ++i) {
            const int ch = (int)threadIdx.x + i * MLP_THREADS, row = ch / 24, c16 = ch - row * 24;
            if (ch < ntt * 16 * 24)
                *reinterpret_cast<v4i *>(sm + MLP_SA + (c16 >> 2) * MLP_KBLK + row * 64 + mlp_phi(row, c16 & 3) * 16) = areg[i];
        }
    };

    // ------------------------------------------------------------------------------------------------------------------
    // one unit of NTT token tiles starting at tile `tile0`; (next_tile0, next_ntt): the unit whose activations to prefetch
    // Barriers: B1 before the first hidden write (every wave is done reading the previous unit's hidden tile; placed AFTER the
    // first chunk's K loop, so a wave that finished its fc2 early already multiplies for the next unit), B2 hidden tile
    // complete / activation tile dead, B3 hidden tile rewritten by ShiftGELU and the NEXT unit's activation tile committed.
    auto unit_body = [&](auto ntt_c, const int ntt, const long long tile0, const long long next_tile0, const int next_ntt) __attribute__((always_inline)) {
        constexpr int NTT = decltype(ntt_c)::value;       // tiles the body multiplies; `ntt` <= NTT of them belong to this unit
        const long long tok0 = tile0 * 16;
        stamp(0);
        // per-lane indices from an opaque copy of the thread id: every LDS address below is (a handful of per-lane bases) +
        // immediates, recomputed per unit — left visible, the ~150 loop-invariant addresses of the unrolled phases are
        // hoisted out of the unit loop into registers and spilled
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, tl = lane & 15, g = lane >> 4;
        const unsigned fb = tl * 64 + mlp_phi(tl, g) * 16;          // this lane's B-fragment chunk inside a K block, token tile 0
        stamp(1);

        // ---- fc1 + qact_gelu (8 bit) into the hidden tile
        {
            const v4i *w1 = p.w1f + (size_t)(wave * NJ) * 64 + lane;
            v4i wf[WD + 1][NJ], bf[2][NTT], acc[NJ][NTT], bias_n[NJ];
            v2d cq[NJ][2];
            auto load_w = [&](int s, int slot) __attribute__((always_inline)) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) wf[slot][j] = w1[(size_t)(s * NJ * MLP_WAVES + j) * 64];
            };
            auto load_b = [&](int s, int slot) __attribute__((always_inline)) {
                const int ks = s % MLP_KS1;
#pragma unroll
                for (int tt = 0; tt < NTT; ++tt)
                    bf[slot][tt] = *reinterpret_cast<const v4i *>(sm + MLP_SA + ks * MLP_KBLK + tt * 1024 + fb);
            };
            auto load_bias = [&](int chunk) __attribute__((always_inline)) {
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    bias_n[j] = *reinterpret_cast<const v4i *>(p.b1 + (wave * CT1 + chunk * NJ + j) * 16 + 4 * g);
            };
#pragma unroll
            for (int s = 0; s < WD; ++s) load_w(s, s);
            load_b(0, 0);
            load_bias(0);
#pragma unroll
            for (int s = 0; s < NS1; ++s) {
                const int chunk = s / MLP_KS1, ks = s - chunk * MLP_KS1, ct0 = wave * CT1 + chunk * NJ;
                __builtin_amdgcn_sched_barrier(0);
                if (s + WD < NS1 && !(MLP_ABLATE & 1)) load_w(s + WD, (s + WD) % (WD + 1));
                if (s + 1 < NS1 && !(MLP_ABLATE & 2)) load_b(s + 1, (s + 1) & 1);
                if (ks == 1) {                       // this chunk's multipliers: consumed five steps on
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const int ch0 = (ct0 + j) * 16 + 4 * g;
                        cq[j][0] = *reinterpret_cast<const v2d *>(p.cq1 + ch0);
                        cq[j][1] = *reinterpret_cast<const v2d *>(p.cq1 + ch0 + 2);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (ks == 0) {
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
#pragma unroll
                        for (int tt = 0; tt < NTT; ++tt) acc[j][tt] = bias_n[j];
                }
                if (ks == 2 && chunk + 1 < NCH) load_bias(chunk + 1);      // the next chunk's bias, four steps ahead
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int tt = 0; tt < NTT; ++tt)
                        acc[j][tt] = (MLP_ABLATE & 8) ? (acc[j][tt] ^ wf[s % (WD + 1)][j] ^ bf[s & 1][tt])
                                                      : __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[s % (WD + 1)][j], bf[s & 1][tt], acc[j][tt], 0, 0, 0);
                if (ks == MLP_KS1 - 1) {
                    if (chunk == 0) __syncthreads();                       // B1: the hidden tile is free
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const int ch0 = (ct0 + j) * 16 + 4 * g;               // this lane's 4 hidden channels
                        const int kb = ch0 >> 6, cc = (ch0 >> 4) & 3;           // fc2 K block and chunk of these channels
#pragma unroll
                        for (int tt = 0; tt < NTT; ++tt) {
                            int o[4];
                            o[0] = mlp_rq<FMA>(acc[j][tt][0], cq[j][0][0]);
                            o[1] = mlp_rq<FMA>(acc[j][tt][1], cq[j][0][1]);
                            o[2] = mlp_rq<FMA>(acc[j][tt][2], cq[j][1][0]);
                            o[3] = mlp_rq<FMA>(acc[j][tt][3], cq[j][1][1]);
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] = min(max(o[e], -128), 127);
                            const unsigned w01 = __builtin_amdgcn_perm((unsigned)o[1], (unsigned)o[0], 0x0c0c0400u);
                            const unsigned w23 = __builtin_amdgcn_perm((unsigned)o[3], (unsigned)o[2], 0x0c0c0400u);
                            const int tok = tt * 16 + tl;
                            *reinterpret_cast<unsigned *>(sm + MLP_SH + kb * MLP_KBLK + tok * 64 + mlp_phi(tok, cc) * 16 + 4 * g) =
                                __builtin_amdgcn_perm(w23, w01, 0x05040100u);
                        }
                    }
                }
            }
        }
        stamp(2);
        __syncthreads();                                                    // B2
        stamp(3);

        // ---- ShiftGELU (+ qact1) in place, half a wavefront per token, NTT tokens per half-wave: the token's 1536 hidden
        // bytes are read once (12 dwords per lane) and stay in registers from the row maximum (packed byte maxima, then 5
        // shuffles) over the fetch of the maximum's 256-byte table line (global -> this half-wave's LDS slot) to the byte
        // gathers and the write-back.  No workgroup barrier inside.  The next unit's activations travel meanwhile.
        if (next_ntt > 0) a_fetch(next_tile0, next_ntt);
        {
            const int hw = wave * 2 + (lane >> 5), l32 = lane & 31;
            typedef __attribute__((address_space(3))) const unsigned char lds_u8;
            typedef unsigned short v2us __attribute__((ext_vector_type(2)));
            const unsigned sm_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char *)sm;
            const unsigned base = sm_lds + MLP_STAB + hw * 256;           // 256-byte aligned: byte | base is the address
            constexpr int NTOK = (NTT * 16 + 2 * MLP_WAVES - 1) / (2 * MLP_WAVES);     // tokens per half-wave
            unsigned w[NTOK][12];
            v2i line[NTOK];
            // pass 1: rows -> registers, row maxima, all table-line requests in flight together (one exposed L2 latency
            // per unit instead of one per token)
#pragma unroll
            for (int i = 0; i < NTOK; ++i) {
                const int t = hw + i * 2 * MLP_WAVES;
                if (t < NTT * 16) {
                    const unsigned *hp = reinterpret_cast<const unsigned *>(sm + MLP_SH + t * 64) + (l32 & 15) + (l32 >> 4) * (MLP_KBLK / 4);
                    v2us me = {0, 0}, mo = {0, 0};                          // running maxima of the even / odd bytes (biased)
#pragma unroll
                    for (int m = 0; m < 12; ++m) {
                        w[i][m] = hp[m * (MLP_KBLK / 2)] ^ 0x80808080u;     // K blocks 2m, 2m + 1 (the upper 16 lanes): Q + 128
                        me = __builtin_elementwise_max(me, __builtin_bit_cast(v2us, __builtin_amdgcn_perm(0u, w[i][m], 0x0c020c00u)));
                        mo = __builtin_elementwise_max(mo, __builtin_bit_cast(v2us, __builtin_amdgcn_perm(0u, w[i][m], 0x0c030c01u)));
                    }
                    const v2us m2 = __builtin_elementwise_max(me, mo);
                    int qb = max((int)m2[0], (int)m2[1]);                    // biased row maximum of this lane
#pragma unroll
                    for (int o = 16; o > 0; o >>= 1) qb = max(qb, __shfl_xor(qb, o));
                    line[i] = reinterpret_cast<const v2i *>(p.tab + (size_t)qb * 256)[l32];
                }
            }
            // pass 2: table line -> this half-wave's LDS slot, byte gathers, write-back.  Wave-level ordering only: the slot
            // belongs to this half-wave and the previous token's gathers were consumed by its write-back
#pragma unroll
            for (int i = 0; i < NTOK; ++i) {
                const int t = hw + i * 2 * MLP_WAVES;
                if (t < NTT * 16) {
                    unsigned *hp = reinterpret_cast<unsigned *>(sm + MLP_SH + t * 64) + (l32 & 15) + (l32 >> 4) * (MLP_KBLK / 4);
                    reinterpret_cast<v2i *>(sm + MLP_STAB + hw * 256)[l32] = line[i];
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
                    for (int m = 0; m < 12; ++m) {
                        const unsigned x = w[i][m];
                        const unsigned b0 = *(lds_u8 *)(size_t)(base | (x & 0xffu)), b1 = *(lds_u8 *)(size_t)(base | ((x >> 8) & 0xffu));
                        const unsigned b2 = *(lds_u8 *)(size_t)(base | ((x >> 16) & 0xffu)), b3 = *(lds_u8 *)(size_t)(base | (x >> 24));
                        hp[m * (MLP_KBLK / 2)] = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
                    }
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                }
            }
        }
        if (next_ntt > 0) a_commit(next_ntt);
        __syncthreads();                                                    // B3
        stamp(5);

        // ---- fc2 + qact2 (16 bit) + qact4 with the identity branch (16 bit)
        {
            const v4i *w2 = p.w2f + (size_t)(wave * NJ) * 64 + lane;
            v4i wf[WD + 1][NJ], bf[2][NTT], acc[NJ][NTT];
            auto load_w = [&](int s, int slot) __attribute__((always_inline)) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) wf[slot][j] = w2[(size_t)(s * NJ * MLP_WAVES + j) * 64];
            };
            auto load_b = [&](int s, int slot) __attribute__((always_inline)) {
#pragma unroll
                for (int tt = 0; tt < NTT; ++tt)
                    bf[slot][tt] = *reinterpret_cast<const v4i *>(sm + MLP_SH + s * MLP_KBLK + tt * 1024 + fb);
            };
#pragma unroll
            for (int s = 0; s < WD; ++s) load_w(s, s);
            load_b(0, 0);
            // identity rows and multipliers of this lane's outputs: requested now, consumed after the K loop
            v2i rs[NJ][NTT];
            v2d c2[NJ][2];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int ch0 = (wave * NJ + j) * 16 + 4 * g;
                c2[j][0] = *reinterpret_cast<const v2d *>(p.cq2 + ch0);
                c2[j][1] = *reinterpret_cast<const v2d *>(p.cq2 + ch0 + 2);
                const v4i b4 = *reinterpret_cast<const v4i *>(p.b2 + ch0);
#pragma unroll
                for (int tt = 0; tt < NTT; ++tt) {
                    acc[j][tt] = b4;
                    const long long tok = min(tok0 + tt * 16 + tl, p.M - 1);
                    rs[j][tt] = *reinterpret_cast<const v2i *>(p.residual + tok * MLP_C + ch0);
                }
            }
#pragma unroll
            for (int s = 0; s < MLP_KS2; ++s) {
                __builtin_amdgcn_sched_barrier(0);
                // keep the two waves of a SIMD abreast: the older one wins every MFMA slot, finishes its K loop thousands of
                // cycles early and leaves the younger one alone at the single-wave 16x16x64 rate (half the pipe)
                if (MLP_FC2_SYNC && s > 0 && (s % (MLP_FC2_SYNC ? MLP_FC2_SYNC : 1)) == 0) __builtin_amdgcn_s_barrier();
                if (s + WD < MLP_KS2 && !(MLP_ABLATE & 1)) load_w(s + WD, (s + WD) % (WD + 1));
                if (s + 1 < MLP_KS2 && !(MLP_ABLATE & 2)) load_b(s + 1, (s + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int tt = 0; tt < NTT; ++tt)
                        acc[j][tt] = (MLP_ABLATE & 8) ? (acc[j][tt] ^ wf[s % (WD + 1)][j] ^ bf[s & 1][tt])
                                                      : __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[s % (WD + 1)][j], bf[s & 1][tt], acc[j][tt], 0, 0, 0);
            }
            stamp(6);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int ch0 = (wave * NJ + j) * 16 + 4 * g;
#pragma unroll
                for (int tt = 0; tt < NTT; ++tt) {
                    int t16[4];
                    t16[0] = mlp_rq<FMA>(acc[j][tt][0], c2[j][0][0]);
                    t16[1] = mlp_rq<FMA>(acc[j][tt][1], c2[j][0][1]);
                    t16[2] = mlp_rq<FMA>(acc[j][tt][2], c2[j][1][0]);
                    t16[3] = mlp_rq<FMA>(acc[j][tt][3], c2[j][1][1]);
                    int o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int t = min(max(t16[e], -32768), 32767);
                        const int r = (int)(short)((unsigned)rs[j][tt][e >> 1] >> (16 * (e & 1)));
                        // both terms are integers < 2^31: the sum is the reference's fp64 sum (quant_utils.py:238-244)
                        o[e] = min(max(rq_fast(r, p.cr) + rq_fast(t, p.cm), -32768), 32767);
                    }
                    const long long tok = tok0 + tt * 16 + tl;
                    if (tok < p.M && tt < ntt)         // a short unit's surplus tiles belong to the next unit
                        *reinterpret_cast<v2i *>(p.out + tok * MLP_C + ch0) =
                            v2i{(int)__builtin_amdgcn_perm((unsigned)o[1], (unsigned)o[0], 0x05040100u),
                                (int)__builtin_amdgcn_perm((unsigned)o[3], (unsigned)o[2], 0x05040100u)};
                }
            }
        }
        stamp(7);
        ++tr_unit;
    };

    // ---- the unit stream
    a_fetch(unit_tile0(0), unit_ntt(0));
    a_commit(unit_ntt(0));
    __syncthreads();
    for (int i = 0; i < nu; ++i) {
        const long long tile0 = unit_tile0(i), tile1 = unit_tile0(i + 1);
        const int ntt = unit_ntt(i), next_ntt = unit_ntt(i + 1);
        if (ntt == MLP_TT) unit_body(std::integral_constant<int, MLP_TT>{}, ntt, tile0, tile1, next_ntt);
        else unit_body(std::integral_constant<int, MLP_TT - 1>{}, ntt, tile0, tile1, next_ntt);
    }
}
