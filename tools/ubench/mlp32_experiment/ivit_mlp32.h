// EXPERIMENT, not part of the product (nothing under i-vit_amd/ includes this file): the 32x32x32 / twelve-wave rewrite of the
// fused Mlp kernel measured in round 3 — bit-exact, but 142-170 us at M = 50432 against 112-125 us for the shipped
// 16x16x64 / eight-wave kernel (i-vit_amd/csrc/ivit_mlp.h).  Kept with its probe so that the timeline and the findings in
// profiles/README.md ("Round 3") can be reproduced:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off
//   [-DMLP_TRACE=1|2] [-DMLP_ABLATE=n] [-DMLP_WD=n -DMLP_BD=n] tools/ubench/mlp32_experiment/mlp32_probe.hip -o mlp32_probe
//
// ivit_mlp.h — Mlp.forward + the block's residual QuantAct as ONE kernel for the D = 384 models (DeiT-S, Swin stage 2):
//   fc1 -> qact_gelu (8 bit) -> ShiftGELU -> qact1 (8 bit) -> fc2 -> qact2 (16 bit) -> qact4(+identity) (16 bit)
// (models/layers_quant.py:144-153, then vit_quant.py:141-142 / swin_quant.py:296-300).  The 4 D-wide hidden tensor never
// exists in HBM: per 64-token unit it is produced into LDS by fc1, rewritten in place by the ShiftGELU table and consumed
// from LDS by fc2.  Unfused, the hidden tensor crosses HBM four times (fc1 write, GELU read + write, fc2 read: 310 MB of
// the layer's 775 MB at batch 256) and ShiftGELU is a launch of its own.
//
// Why 64 tokens and why it pays although each unit re-reads both weight matrices (1.18 MB) from L2: ShiftGELU's row
// maximum couples all 1536 hidden channels of a token, so a unit must own whole hidden rows; 64 rows x 1536 B is what the
// LDS holds next to the activation tile.  The weights stream L2 -> registers (never through LDS): they are laid out at
// plan time in MFMA-fragment order (1 KB per (32-channel tile, 32-column step), lane-linear), every wave reads only the
// fragments of ITS output channels, and a fragment feeds two MFMAs (two 32-token tiles).  No barrier inside the two GEMM
// phases: the twelve waves drift apart, one wave's requant epilogue runs beside its SIMD mates' MFMAs.
//
// Shapes: v_mfma_i32_32x32x32_i8, twelve waves (three per SIMD).  Measured (tools/ubench/requant_mix.hip,
// profiles/r03_ubench_mfma_valu.txt): ONE wave keeps a SIMD's matrix pipe full with 32x32x32 (3.44 POP/s chip-wide at one
// wave per SIMD) but only half full with 16x16x64 (1.9-2.4) — in a kernel whose waves take turns in epilogues, table
// look-ups and load waits the pipe is fed by a lone wave much of the time (the 16x16x64 / 8-wave predecessor of this
// kernel spent 925 cycles per token; timeline in profiles/README.md round 3).  Weights are the A operand (rows =
// channels), activations B (columns = tokens): a lane holds, per accumulator tile, 4 x 4 consecutive channels of one
// token, each group packing into one dword (fc1) or one 8-byte store (fc2).  fc1: wave w owns the 32-channel tiles
// w, 12 + w, 24 + w, 36 + w, one after the other (2 accumulator tiles = 32 registers); fc2: wave w owns output
// channels [32 w, 32 w + 32).  Both phases are the same 48-step loop: one weight fragment (1 KB from L2), two
// activation fragments (2 x ds_read_b128), two MFMAs.
//
// LDS images are K-blocked: [64-column block][64 tokens][80 B] (64 B of data + 16 B of padding), blocks 5184 B apart.
// The 80-byte token pitch makes every 16-lane group of a ds_read_b128 fragment read ({0-3,12-15,20-27}, ... of 32
// consecutive tokens, one 16-byte chunk each: 5 * token mod 16 distinct) conflict-free; the 64 B by which a block
// exceeds a multiple of 256 B puts the two blocks a half-wave reads in ShiftGELU on disjoint banks.
#pragma once
#include <type_traits>
#include "../../../i-vit_amd/csrc/ivit_device.h"

#define MLP_C 384
#define MLP_HD 1536
#define MLP_WAVES 12
#define MLP_THREADS (MLP_WAVES * 64)
#define MLP_TOK 64                            // tokens per unit: two MFMA token tiles
#define MLP_KS1 (MLP_C / 32)                  // 12 column steps of fc1 (per 32-channel tile)
#define MLP_KS2 (MLP_HD / 32)                 // 48 column steps of fc2
#define MLP_PITCH 80                          // bytes between tokens inside a K block
#define MLP_KBLK (MLP_TOK * MLP_PITCH + 64)   // one 64-column block of an LDS image
#define MLP_SH 0                              // hidden tile: 24 blocks
#define MLP_SA (24 * MLP_KBLK)                // activation tile: 6 blocks
#define MLP_STAB ((30 * MLP_KBLK + 255) / 256 * 256)   // one ShiftGELU table line (256 B, 256-byte aligned) per half-wave
#define MLP_SMEM (MLP_STAB + 2 * MLP_WAVES * 256)
#define MLP_MAGIC 6755399441055744.0
// timeline instrumentation (tools/ubench/mlp_probe.hip, -DMLP_TRACE=1): every wave of workgroup 0 stamps the cycle counter
// at the phase boundaries of its first units into p.trace[(unit_index * MLP_WAVES + wave) * 8 + point]
#ifndef MLP_TRACE
#define MLP_TRACE 0
#endif
// timing ablations (probe builds only; results invalid): 1 = no weight loads after the prologue, 2 = no activation-fragment
// LDS reads after the prologue, 8 = no MFMA
#ifndef MLP_ABLATE
#define MLP_ABLATE 0
#endif
#ifndef MLP_WD
#define MLP_WD 10                             // weight fragments in flight per wave (12 waves x 10 KB = 120 KB per CU)
#endif
#ifndef MLP_SLACK
#define MLP_SLACK 1                           // ring slots beyond the prefetch distance: a load never targets registers the two
#endif                                        // most recently issued MFMAs read
#ifndef MLP_BD
#define MLP_BD 2                              // steps by which the activation fragments (LDS) run ahead of their MFMAs
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
    unsigned long long *trace;   // MLP_TRACE builds only
};

// weights [N][K] int8 -> fragments of 64 lanes x 16 B, lane l = W[ct*32 + (l & 31)][ks*32 + (l >> 5)*16 ...], in the order the
// kernel consumes them: fragment index f = step * 12 + wave, where step = chunk * (K / 32) + ks walks the wave's channel
// tiles ct = chunk * 12 + wave and the 32-column steps inside a tile.  What the twelve waves of a workgroup request in one
// step is ONE contiguous 12 KB window: the requests spread over all L2 channels.
__global__ __launch_bounds__(256) void mlp_swizzle_kernel(const int8_t *__restrict__ w, int N, int K, v4i *__restrict__ wf) {
    const int nks = K >> 5;
    const long long total = (long long)(N >> 5) * nks * 64;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int l = (int)(i & 63);
        const int f = (int)(i >> 6), step = f / MLP_WAVES, wv = f - step * MLP_WAVES;
        const int chunk = step / nks, ks = step - chunk * nks, ct = chunk * MLP_WAVES + wv;
        wf[i] = *reinterpret_cast<const v4i *>(w + (long long)(ct * 32 + (l & 31)) * K + ks * 32 + (l >> 5) * 16);
    }
}

// compile-time loop: the 48-step phases must be straight-line code (register-resident fragment rings, per-step immediates);
// "#pragma unroll" gives up silently above its size threshold and leaves a loop that indexes registers through M0
template <int I, int N, class F>
__device__ __forceinline__ void mlp_static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        mlp_static_for<I + 1, N>(f);
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
// Units and balance.  The token axis is cut into tiles of 32; workgroup b owns the contiguous tile range
// [T b / G, T (b + 1) / G) and walks it in units of two tiles (one where the count is odd).  A unit costs one pass over
// both weight matrices whatever its size; the one-tile body issues half the MFMAs.
//
// Software pipeline of both GEMM phases (pinned with scheduling fences: left alone the scheduler sinks every load to just
// before its first use and each step waits out a full LDS / L2 latency with the matrix pipe idle; hoisted to the top of
// the unrolled phase they are all live at once and spill): step s issues the weight fragment of step s + MLP_WD and the
// activation fragments of step s + 1, then its own MFMAs.
template <bool FMA>
__global__ __launch_bounds__(MLP_THREADS, 1) void mlp384_kernel(MlpArgs p) {
    extern __shared__ __attribute__((aligned(256))) char sm[];
    constexpr int WD = MLP_WD, BD = MLP_BD, WR = MLP_WD + MLP_SLACK, BR = MLP_BD + MLP_SLACK, NCH = MLP_HD / 32 / MLP_WAVES, NS1 = NCH * MLP_KS1;      // fc1: 4 tiles per wave, 48 steps
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    typedef double v2d __attribute__((ext_vector_type(2)));

    // ---- this workgroup's units: (first tile, tiles) of unit i
    const long long ntiles = (p.M + 31) >> 5;
    const long long t_beg = ntiles * blockIdx.x / gridDim.x, t_end = ntiles * (blockIdx.x + 1) / gridDim.x;
    const int n_own = (int)(t_end - t_beg), nu = (n_own + 1) >> 1;
    if (nu <= 0) return;
    auto unit_tile0 = [&](int i) -> long long { return t_beg + (long long)n_own * i / nu; };
    auto unit_ntt = [&](int i) -> int { return i >= nu ? 0 : (int)(unit_tile0(i + 1) - unit_tile0(i)); };

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
    v4i areg[2];
    auto a_fetch = [&](long long tile0, int ntt) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ch = (int)threadIdx.x + i * MLP_THREADS, row = ch / 24, c16 = ch - row * 24;
            if (row < ntt * 32) {
                const long long grow = min(tile0 * 32 + row, p.M - 1);
                areg[i] = *reinterpret_cast<const v4i *>(p.x + grow * MLP_C + c16 * 16);
            }
        }
    };
    auto a_commit = [&](int ntt) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ch = (int)threadIdx.x + i * MLP_THREADS, row = ch / 24, c16 = ch - row * 24;
            if (row < ntt * 32)
                *reinterpret_cast<v4i *>(sm + MLP_SA + (c16 >> 2) * MLP_KBLK + row * MLP_PITCH + (c16 & 3) * 16) = areg[i];
        }
    };

    // ------------------------------------------------------------------------------------------------------------------
    // one unit of NTT token tiles starting at tile `tile0`; (next_tile0, next_ntt): the unit whose activations to prefetch
    // Barriers: B1 before the first hidden write (every wave is done reading the previous unit's hidden tile; placed AFTER the
    // first tile's K loop, so a wave that finished its fc2 early already multiplies for the next unit), B2 hidden tile
    // complete / activation tile dead, B3 hidden tile rewritten by ShiftGELU and the NEXT unit's activation tile committed.
    auto unit_body = [&](auto ntt_c, const long long tile0, const long long next_tile0, const int next_ntt) __attribute__((always_inline)) {
        constexpr int NTT = decltype(ntt_c)::value;
        const long long tok0 = tile0 * 32;
        stamp(0);
        // per-lane indices from an opaque copy of the thread id: every LDS address below is (a handful of per-lane bases) +
        // immediates, recomputed per unit — left visible, the loop-invariant addresses of the unrolled phases are hoisted
        // out of the unit loop into registers and spilled
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, tl = lane & 31, h = lane >> 5;
        const unsigned fb = tl * MLP_PITCH + h * 16;               // this lane's B-fragment chunk inside a K block, token tile 0
        // Everything a GEMM phase reads from or writes to global memory goes through buffer resources: address = resource
        // base (scalar, per unit) + one per-lane offset register + scalar / immediate offsets, and rows past M are dropped by
        // the resource's range check instead of per-lane clamps and predicates.  No VALU address arithmetic is left inside or
        // in front of the K loops: a VALU instruction of one wave issues only into the gaps the other two waves' MFMAs leave
        // on the SIMD (measured: the third wave of a SIMD needed 8.8k cycles for the ~60 address instructions in front of
        // its fc2 loop and entered it when the first wave was done).
        const unsigned h16 = h * 16, rowoff = tl * (MLP_C * 2) + h * 8;
        const long long rows_left = p.M - tok0;
        const unsigned rbytes = (unsigned)(rows_left < NTT * 32 ? rows_left : NTT * 32) * (MLP_C * 2);
        const __amdgpu_buffer_rsrc_t res_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<int16_t *>(p.residual) + tok0 * MLP_C, 0, rbytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t out_rs = __builtin_amdgcn_make_buffer_rsrc(p.out + tok0 * MLP_C, 0, rbytes, 0x00020000);
        stamp(1);

        // ---- fc1 + qact_gelu (8 bit) into the hidden tile
        if (!(MLP_ABLATE & 32)) {
            // buffer loads: resource + lane offset (one VGPR for the whole phase) + scalar offset — a step's address is SALU work.
            // (Per-lane 64-bit pointers cost VALU instructions per step, and VALU issue waits behind the matrix pipe: measured
            // ~120 of a step's 188 cycles, the three waves of a SIMD taking turns instead of overlapping.)
            const __amdgpu_buffer_rsrc_t w1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<v4i *>(p.w1f), 0, MLP_C * MLP_HD, 0x00020000);
            const unsigned loff = (unsigned)lane * 16u;
            v4i wf[WR], bf[BR][NTT];
            v16i acc[NTT];
            v2d cq[4][2];
            v4i bias[4];
            int so = wave * 1024;                  // scalar offset of the next fragment; advanced by an opaque SALU add (left to the
            //                                        compiler, the 48 offsets are materialised up front and spilled to VGPR lanes)
            auto load_w = [&](int, int slot) __attribute__((always_inline)) {
                wf[slot] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(w1, loff, so, 0));
                asm volatile("s_add_u32 %0, %0, 0x3000" : "+s"(so));
            };
            auto load_b = [&](int s, int slot) __attribute__((always_inline)) {
                const int ks = s % MLP_KS1;
#pragma unroll
                for (int tt = 0; tt < NTT; ++tt)
                    bf[slot][tt] = *reinterpret_cast<const v4i *>(sm + MLP_SA + (ks >> 1) * MLP_KBLK + tt * 32 * MLP_PITCH + (ks & 1) * 32 + fb);
            };
            // A tile's multipliers and biases are requested where no weight fragment issued after them is needed soon: the memory
            // counter retires in order, so a wait for a fragment also waits for every older load — a table line that has left
            // the L2 (1-2 us) in the middle of the ring stalls the K loop for its whole latency.  First tile: ahead of the
            // ring's prologue; tile c + 1: right after tile c's epilogue, MLP_WD steps before the next fragment issued behind
            // them is consumed.
            const __amdgpu_buffer_rsrc_t cq_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(p.cq1), 0, MLP_HD * 8, 0x00020000);
            const __amdgpu_buffer_rsrc_t b_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t *>(p.b1), 0, MLP_HD * 4, 0x00020000);
            auto load_consts = [&](int chunk) __attribute__((always_inline)) {
                const int ct = chunk * MLP_WAVES + wave;              // channels ct*32 + q*8 + h*4 .. + 3
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    cq[q][0] = __builtin_bit_cast(v2d, __builtin_amdgcn_raw_buffer_load_b128(cq_rs, 2 * h16 + q * 64, ct * 256, 0));
                    cq[q][1] = __builtin_bit_cast(v2d, __builtin_amdgcn_raw_buffer_load_b128(cq_rs, 2 * h16 + q * 64 + 16, ct * 256, 0));
                    bias[q] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(b_rs, h16 + q * 32, ct * 128, 0));
                }
            };
            load_consts(0);
#pragma unroll
            for (int s = 0; s < WD; ++s) load_w(s, s);
#pragma unroll
            for (int s = 0; s < BD; ++s) load_b(s, s);
            mlp_static_for<0, NS1>([&](auto s_c) __attribute__((always_inline)) {
                constexpr int s = decltype(s_c)::value, chunk = s / MLP_KS1, ks = s - chunk * MLP_KS1;
                const int ct = chunk * MLP_WAVES + wave;
                __builtin_amdgcn_sched_barrier(0);
                if (s + WD < NS1 && !(MLP_ABLATE & 1)) load_w(s + WD, (s + WD) % WR);
                if (s + BD < NS1 && !(MLP_ABLATE & 2)) load_b(s + BD, (s + BD) % BR);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int tt = 0; tt < NTT; ++tt) {
                    const v16i c0 = ks == 0 ? v16i{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0} : acc[tt];
                    if ((MLP_ABLATE & 16) && tt > 0) {
                        acc[tt][0] = c0[0] ^ wf[s % WR][0] ^ bf[s % BR][tt][0];
                    } else if (MLP_ABLATE & 8) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc[tt][e] = c0[e] ^ wf[s % WR][e & 3] ^ bf[s % BR][tt][e & 3];
                    } else {
                        acc[tt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[s % WR], bf[s % BR][tt], c0, 0, 0, 0);
                    }
                }
                if (ks == MLP_KS1 - 1) {
                    if (chunk == 0) __syncthreads();                       // B1: the hidden tile is free
                    // channels ct*32 + q*8 + h*4 + e sit in accumulator element 4 q + e: fc2 column step ct, K block ct >> 1
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
#pragma unroll
                        for (int tt = 0; tt < NTT; ++tt) {
                            int o[4];
                            o[0] = mlp_rq<FMA>(acc[tt][4 * q + 0] + bias[q][0], cq[q][0][0]);
                            o[1] = mlp_rq<FMA>(acc[tt][4 * q + 1] + bias[q][1], cq[q][0][1]);
                            o[2] = mlp_rq<FMA>(acc[tt][4 * q + 2] + bias[q][2], cq[q][1][0]);
                            o[3] = mlp_rq<FMA>(acc[tt][4 * q + 3] + bias[q][3], cq[q][1][1]);
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] = min(max(o[e], -128), 127);
                            const unsigned w01 = __builtin_amdgcn_perm((unsigned)o[1], (unsigned)o[0], 0x0c0c0400u);
                            const unsigned w23 = __builtin_amdgcn_perm((unsigned)o[3], (unsigned)o[2], 0x0c0c0400u);
                            *reinterpret_cast<unsigned *>(sm + MLP_SH + (ct >> 1) * MLP_KBLK + (tt * 32 + tl) * MLP_PITCH + (ct & 1) * 32 + q * 8 + h * 4) =
                                __builtin_amdgcn_perm(w23, w01, 0x05040100u);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (chunk + 1 < NCH) load_consts(chunk + 1);
                }
            });
        }
        stamp(2);
        __syncthreads();                                                    // B2
        stamp(3);

        // ---- ShiftGELU (+ qact1) in place, half a wavefront per token: the token's 1536 hidden bytes are read once
        // (12 dwords per lane) and stay in registers from the row maximum (packed byte maxima, then 5 shuffles) over the
        // fetch of the maximum's 256-byte table line (global -> this half-wave's LDS slot) to the byte gathers and the
        // write-back.  No workgroup barrier inside.  The next unit's activations travel meanwhile.
        if (next_ntt > 0) a_fetch(next_tile0, next_ntt);
        if (!(MLP_ABLATE & 64)) {
            const int hw = wave * 2 + h;
            typedef __attribute__((address_space(3))) const unsigned char lds_u8;
            typedef unsigned short v2us __attribute__((ext_vector_type(2)));
            const unsigned sm_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char *)sm;
            const unsigned base = sm_lds + MLP_STAB + hw * 256;           // 256-byte aligned: byte | base is the address
            constexpr int NTOK = (NTT * 32 + 2 * MLP_WAVES - 1) / (2 * MLP_WAVES);     // tokens per half-wave
            unsigned w[NTOK][12];
            v2i line[NTOK];
            // pass 1: rows -> registers, row maxima, all table-line requests in flight together (one exposed L2 latency
            // per unit instead of one per token).  Lane l of the half-wave reads dword l & 15 of blocks 2 m + (l >> 4).
#pragma unroll
            for (int i = 0; i < NTOK; ++i) {
                const int t = hw + i * 2 * MLP_WAVES;
                if (t < NTT * 32) {
                    const unsigned *hp = reinterpret_cast<const unsigned *>(sm + MLP_SH + t * MLP_PITCH + (tl >> 4) * MLP_KBLK) + (tl & 15);
                    v2us me = {0, 0}, mo = {0, 0};                          // running maxima of the even / odd bytes (biased)
#pragma unroll
                    for (int m = 0; m < 12; ++m) {
                        w[i][m] = hp[m * (MLP_KBLK / 2)] ^ 0x80808080u;     // Q + 128
                        me = __builtin_elementwise_max(me, __builtin_bit_cast(v2us, __builtin_amdgcn_perm(0u, w[i][m], 0x0c020c00u)));
                        mo = __builtin_elementwise_max(mo, __builtin_bit_cast(v2us, __builtin_amdgcn_perm(0u, w[i][m], 0x0c030c01u)));
                    }
                    const v2us m2 = __builtin_elementwise_max(me, mo);
                    int qb = max((int)m2[0], (int)m2[1]);                    // biased row maximum of this lane
#pragma unroll
                    for (int o = 16; o > 0; o >>= 1) qb = max(qb, __shfl_xor(qb, o));
                    line[i] = reinterpret_cast<const v2i *>(p.tab + (size_t)qb * 256)[tl];
                }
            }
            // pass 2: table line -> this half-wave's LDS slot, byte gathers, write-back.  Wave-level ordering only: the slot
            // belongs to this half-wave and the previous token's gathers were consumed by its write-back
#pragma unroll
            for (int i = 0; i < NTOK; ++i) {
                const int t = hw + i * 2 * MLP_WAVES;
                if (t < NTT * 32) {
                    unsigned *hp = reinterpret_cast<unsigned *>(sm + MLP_SH + t * MLP_PITCH + (tl >> 4) * MLP_KBLK) + (tl & 15);
                    reinterpret_cast<v2i *>(sm + MLP_STAB + hw * 256)[tl] = line[i];
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
            const __amdgpu_buffer_rsrc_t w2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<v4i *>(p.w2f), 0, MLP_C * MLP_HD, 0x00020000);
            const unsigned loff = (unsigned)lane * 16u;
            v4i wf[WR], bf[BR][NTT];
            v16i acc[NTT];
            int so = wave * 1024;                  // scalar offset of the next fragment; advanced by an opaque SALU add (left to the
            //                                        compiler, the 48 offsets are materialised up front and spilled to VGPR lanes)
            auto load_w = [&](int, int slot) __attribute__((always_inline)) {
                wf[slot] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(w2, loff, so, 0));
                asm volatile("s_add_u32 %0, %0, 0x3000" : "+s"(so));
            };
            auto load_b = [&](int s, int slot) __attribute__((always_inline)) {
#pragma unroll
                for (int tt = 0; tt < NTT; ++tt)
                    bf[slot][tt] = *reinterpret_cast<const v4i *>(sm + MLP_SH + (s >> 1) * MLP_KBLK + tt * 32 * MLP_PITCH + (s & 1) * 32 + fb);
            };
            // identity rows (HBM) of this lane's outputs: the OLDEST loads of the phase (see fc1); biases and multipliers: behind
            // the last weight fragment
            v2i rs[NTT][4];
            v2d c2[4][2];
            v4i b4[4];
            const __amdgpu_buffer_rsrc_t cq2_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(p.cq2), 0, MLP_C * 8, 0x00020000);
            const __amdgpu_buffer_rsrc_t b2_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t *>(p.b2), 0, MLP_C * 4, 0x00020000);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int tt = 0; tt < NTT; ++tt)
                    rs[tt][q] = __builtin_bit_cast(v2i, __builtin_amdgcn_raw_buffer_load_b64(res_rs, rowoff + q * 16, tt * (32 * MLP_C * 2) + wave * 64, 0));
#pragma unroll
            for (int s = 0; s < WD; ++s) load_w(s, s);
#pragma unroll
            for (int s = 0; s < BD; ++s) load_b(s, s);
            mlp_static_for<0, MLP_KS2>([&](auto s_c) __attribute__((always_inline)) {
                constexpr int s = decltype(s_c)::value;
                __builtin_amdgcn_sched_barrier(0);
                if (s + WD < MLP_KS2 && !(MLP_ABLATE & 1)) load_w(s + WD, (s + WD) % WR);
                if (s + BD < MLP_KS2 && !(MLP_ABLATE & 2)) load_b(s + BD, (s + BD) % BR);
                if (MLP_TRACE == 2 && (s % 6 == 0 || s == MLP_KS2 - 1)) {      // progress of every wave through the K loop
                    unsigned long long t;
                    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t));
                    if (blockIdx.x == 0 && tr_unit == 0 && (threadIdx.x & 63) == 0)
                        p.trace[4 * MLP_WAVES * 8 + wave * 16 + (s == MLP_KS2 - 1 ? 8 : s / 6)] = t;
                }
                if (s == MLP_KS2 - WD) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        c2[q][0] = __builtin_bit_cast(v2d, __builtin_amdgcn_raw_buffer_load_b128(cq2_rs, 2 * h16 + q * 64, wave * 256, 0));
                        c2[q][1] = __builtin_bit_cast(v2d, __builtin_amdgcn_raw_buffer_load_b128(cq2_rs, 2 * h16 + q * 64 + 16, wave * 256, 0));
                        b4[q] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(b2_rs, h16 + q * 32, wave * 128, 0));
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int tt = 0; tt < NTT; ++tt) {
                    const v16i c0 = s == 0 ? v16i{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0} : acc[tt];
                    if ((MLP_ABLATE & 16) && tt > 0) {
                        acc[tt][0] = c0[0] ^ wf[s % WR][0] ^ bf[s % BR][tt][0];
                    } else if (MLP_ABLATE & 8) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc[tt][e] = c0[e] ^ wf[s % WR][e & 3] ^ bf[s % BR][tt][e & 3];
                    } else {
                        acc[tt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[s % WR], bf[s % BR][tt], c0, 0, 0, 0);
                    }
                }
            });
            stamp(6);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int tt = 0; tt < NTT; ++tt) {
                    int t16[4];
                    t16[0] = mlp_rq<FMA>(acc[tt][4 * q + 0] + b4[q][0], c2[q][0][0]);
                    t16[1] = mlp_rq<FMA>(acc[tt][4 * q + 1] + b4[q][1], c2[q][0][1]);
                    t16[2] = mlp_rq<FMA>(acc[tt][4 * q + 2] + b4[q][2], c2[q][1][0]);
                    t16[3] = mlp_rq<FMA>(acc[tt][4 * q + 3] + b4[q][3], c2[q][1][1]);
                    int o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int t = min(max(t16[e], -32768), 32767);
                        const int r = (int)(short)((unsigned)rs[tt][q][e >> 1] >> (16 * (e & 1)));
                        // both terms are integers < 2^31: the sum is the reference's fp64 sum (quant_utils.py:238-244)
                        o[e] = min(max(rq_fast(r, p.cr) + rq_fast(t, p.cm), -32768), 32767);
                    }
                    typedef unsigned v2u __attribute__((ext_vector_type(2)));
                    __builtin_amdgcn_raw_buffer_store_b64(v2u{__builtin_amdgcn_perm((unsigned)o[1], (unsigned)o[0], 0x05040100u),
                                                              __builtin_amdgcn_perm((unsigned)o[3], (unsigned)o[2], 0x05040100u)},
                                                          out_rs, rowoff + q * 16, tt * (32 * MLP_C * 2) + wave * 64, 0);
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
        if (ntt == 2) unit_body(std::integral_constant<int, 2>{}, tile0, tile1, next_ntt);
        else unit_body(std::integral_constant<int, 1>{}, tile0, tile1, next_ntt);
    }
}
