// ivit_swin_mlp_rs.h — the narrow-stage fused Mlp of ivit_swin.h (C = 96, hidden 384: Swin-T / S stage 0; layers_quant.py:144-153
// followed by the block's residual QuantAct, swin_quant.py:293-300), role-split: same arguments, same integers as
// swin_mlp_fused_kernel, which walks a 64-token tile through four workgroup-wide phases (fc1, table lines, gathers, fc2 on 6
// of its 16 waves) with a barrier between each — 8-9 k cycles per tile for 1.15 k cycles of MFMA.
//
// Here the weights live in REGISTERS (a wave's own fragments, for the whole launch), the hidden tile is DOUBLE-buffered
// (2 x 24 KB — at this width two tiles fit, the structure the round-4 review asked for) and the sixteen waves split by role:
//   waves 0-7,  producers: fc1 + qact_gelu (8 bit) of tile i + 1 into hidden buffer (i + 1) & 1, row maxima by ds_max
//   waves 8-15, consumers: ShiftGELU (+ qact1) by table, fc2 + qact2 (16 bit) + qact4 with the identity branch of tile i
// hand-over by monotone LDS counters (rs_signal / rs_wait of ivit_mlp_rs.h): no workgroup barrier after the prologue, the
// producers' requant VALU runs beside the consumers' gathers and MFMAs.  MFMA rows of both weight matrices are placed so that a
// lane owns 16 consecutive output channels (one ds_write_b128 into the hidden tile, two 16-byte global stores in the epilogue,
// no permlane exchange); the requants use the magic-number form with saturating packs where |z c| < 2^31 is provable for every
// channel (checked once per workgroup), the v_rndne_f64 form otherwise.
#pragma once
#include "ivit_gemm2.h"
#include "ivit_swin.h"
#include "ivit_mlp_rs.h"

#define SR_X 0                                       // 2 x 6 144, [k-step 3][token tile 2][32 tokens][32 B]
#define SR_H (SR_X + 2 * MF_BM * MF_C)               // 2 x 24 576, [k-step 12][64 tokens][32 B]
#define SR_TAB (SR_H + 2 * MF_BM * MF_HD)            // two ShiftGELU table lines per consumer half-wave (16 x 2 x 256)
#define SR_C1 (SR_TAB + 8192)                        // 384 doubles
#define SR_B1 (SR_C1 + MF_HD * 8)                    // 384 ints
#define SR_C2 (SR_B1 + MF_HD * 4)                    // 96 doubles
#define SR_B2 (SR_C2 + MF_C * 8)                     // 96 ints
#define SR_MAX (SR_B2 + MF_C * 4)                    // 2 x 64 ints: row maxima (biased), per hidden buffer
#define SR_FLAG (SR_MAX + 2 * MF_BM * 4)             // 16 counters
#define SR_SMEM (SR_FLAG + 64)
#define SR_THREADS 1024
#define SR_F_XL 0                                    // activation tiles landed (loader: +1 per tile)
// The producers may be two tiles ahead of the consumers and up to one tile apart from each other, so "all eight producers are
// done with tile i" is counted PER BUFFER (tile parity): a wave that is already through tile i + 1 adds to the other counter.
// (One counter for both parities passed with seven waves through tile i + 1 and the eighth still inside tile i: found by
// test_swin_sliced_concurrency_stress.)  The consumers cannot drift apart by a tile (F_G holds them together): one counter each.
#define SR_F_XF 1                                    // .. 2: producers done with activation buffer b (+8 per tile of parity b)
#define SR_F_H 3                                     // .. 4: hidden buffer b complete (+8 per tile of parity b)
#define SR_F_G 5                                     // ShiftGELU complete (+8 per tile)
#define SR_F_F 6                                     // consumers done reading a hidden buffer (+8 per tile)

__global__ __launch_bounds__(SR_THREADS) void swin_mlp_rs_kernel(MlpFusedArgs p) {
    extern __shared__ __attribute__((aligned(256))) char sm[];
    typedef double v2d __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(3))) char lds_c;
    typedef __attribute__((address_space(3))) v4i lds_v4i;
    typedef __attribute__((address_space(3))) v2i lds_v2i;
    typedef __attribute__((address_space(3))) unsigned lds_u32;
    typedef __attribute__((address_space(3))) int lds_i32;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), half = lane >> 5, l31 = lane & 31;
    const unsigned sm_lds = (unsigned)(size_t)(lds_c *)sm, fl = sm_lds + SR_FLAG;
    double *sC1 = reinterpret_cast<double *>(sm + SR_C1), *sC2 = reinterpret_cast<double *>(sm + SR_C2);
    int *sB1 = reinterpret_cast<int *>(sm + SR_B1), *sB2 = reinterpret_cast<int *>(sm + SR_B2);

    // ---- one-off: constants, counters.  The weights never enter the LDS: every wave keeps the fragments of ITS units in
    // registers for the whole launch (producers 3 channel tiles x 3 k-steps = 36 registers, consumers 12 k-steps = 48), loaded
    // straight from W [N][K] with MFMA row 8 q + 4 h + i of a 32-channel tile holding channel 16 h + 4 q + i — half of the
    // fragment traffic of the LDS-resident form (one 1 KB weight fragment per MFMA) is gone
    bool wide = false;
    if (tid < MF_HD) {
        const double cv = p.dy1[tid].m * p.dy1[tid].r;
        const int bs = p.b1 ? p.b1[tid] : 0;
        sC1[tid] = cv; sB1[tid] = bs;
        wide |= !(fabs(cv) * ((double)MF_C * 16384.0 + fabs((double)bs)) < 2147483000.0);      // magic-number rounding needs |z c| < 2^31
    }
    if (tid < MF_C) {
        const double cv = p.dy2[tid].m * p.dy2[tid].r;
        const int bs = p.b2 ? p.b2[tid] : 0;
        sC2[tid] = cv; sB2[tid] = bs;
        wide |= !(fabs(cv) * ((double)MF_HD * 16384.0 + fabs((double)bs)) < 2147483000.0);
    }
    if (tid < 16) reinterpret_cast<unsigned *>(sm + SR_FLAG)[tid] = 0;
    if (tid < 2 * MF_BM) reinterpret_cast<int *>(sm + SR_MAX)[tid] = (int)0x80000000;
    const bool fastrq = !__syncthreads_or(wide);
    const double cm = p.dy_main.m * p.dy_main.r, cr = p.dy_res.m * p.dy_res.r;
    const bool res_fast = fabs(cm) < RQ_FAST_CLIM && fabs(cr) < RQ_FAST_CLIM;

    const long long ntiles = (p.M + MF_BM - 1) / MF_BM;
    const int nmine = (int)((ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x);       // tiles blockIdx.x + i gridDim.x
    if (nmine <= 0) return;

    if (wave < 8) {
        // =========================================================================================== producers
        // activation tile -> LDS by DMA, 6 pieces of 1 KB (k-step, token tile): lanes of a piece = 32 tokens x 2 chunks
        auto issue_x = [&](int i, int buf) __attribute__((always_inline)) {
            const long long tile = blockIdx.x + (long long)i * gridDim.x;
#pragma unroll
            for (int pc = 0; pc < 6; ++pc) {
                const int kc = pc >> 1, mt = pc & 1;
                const long long t = min(tile * MF_BM + mt * 32 + l31, p.M - 1);
                const int8_t *src = p.x + t * MF_C + kc * 32 + half * 16;
                const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(SR_X + buf * (MF_BM * MF_C) + pc * 1024));
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)(sm + dst), 16, 0, 0);
            }
        };
        if (wave == 0) {
            issue_x(0, 0);
            if (nmine > 1) issue_x(1, 1);
        }
        v4i w1f[3][3];                                  // [unit k3][k-step]: channel tile (wave + 8 k3) >> 1
#pragma unroll
        for (int k3 = 0; k3 < 3; ++k3)
#pragma unroll
            for (int kc = 0; kc < 3; ++kc)
                w1f[k3][kc] = *reinterpret_cast<const v4i *>(p.w1 + (size_t)(((wave + 8 * k3) >> 1) * 32 + rs_chan_of_row(l31)) * MF_C + kc * 32 + half * 16);
        auto produce = [&](auto use_fast) __attribute__((always_inline)) {
            for (int i = 0; i < nmine; ++i) {
                const int b = i & 1;
                if (wave == 0) {                       // the loader: what it requested last iteration (two tiles before the loop) has landed
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    rs_signal(fl + 4 * SR_F_XL);
                    if (i == 0 && nmine > 1) rs_signal(fl + 4 * SR_F_XL);
                }
                rs_wait(fl + 4 * SR_F_XL, (unsigned)i + 1);
                rs_wait(fl + 4 * SR_F_F, i >= 2 ? 8u * (unsigned)(i - 1) : 0u);      // hidden buffer b: the consumers are done with tile i - 2
                const unsigned xa = sm_lds + SR_X + b * (MF_BM * MF_C) + lane * 16;
                const unsigned ha = sm_lds + SR_H + b * (MF_BM * MF_HD) + l31 * 32 + half * 16;
#pragma unroll
                for (int k3 = 0; k3 < 3; ++k3) {
                    const int unit = wave + 8 * k3, nt = unit >> 1, mt = unit & 1;      // 24 (channel tile, token tile) units over 8 waves
                    v16i acc;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const v4i b4 = *reinterpret_cast<const v4i *>(sB1 + nt * 32 + half * 16 + q * 4);
                        acc[4 * q] = b4[0]; acc[4 * q + 1] = b4[1]; acc[4 * q + 2] = b4[2]; acc[4 * q + 3] = b4[3];
                    }
#pragma unroll
                    for (int kc = 0; kc < 3; ++kc) {
                        const v4i xf = *(lds_v4i *)(size_t)(xa + (kc * 2 + mt) * 1024);
                        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(w1f[k3][kc], xf, acc, 0, 0, 0);
                    }
                    // register v of lane (token, h) is hidden channel 32 nt + 16 h + v; the tile holds BIASED bytes (Q + 128)
                    int mx = 0;
                    v4i hw;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        int o[4];
#pragma unroll
                        for (int e = 0; e < 4; e += 2) {
                            const v2d c2 = *reinterpret_cast<const v2d *>(sC1 + nt * 32 + half * 16 + q * 4 + e);
#pragma unroll
                            for (int f = 0; f < 2; ++f) {
                                const double t = (double)acc[4 * q + e + f] * c2[f];
                                o[e + f] = decltype(use_fast)::value ? __double2loint(t + (6755399441055744.0 + 128.0))
                                                                     : min(max(rint_sat_i32(t), -128), 127) + 128;
                            }
                        }
                        mx = max(max(mx, o[0]), o[1]);
                        mx = max(max(mx, o[2]), o[3]);
                        unsigned p01, p23, b01, b23;
                        asm("v_cvt_pk_i16_i32 %0, %1, %2" : "=v"(p01) : "v"(o[0]), "v"(o[1]));
                        asm("v_cvt_pk_i16_i32 %0, %1, %2" : "=v"(p23) : "v"(o[2]), "v"(o[3]));
                        asm("v_sat_pk_u8_i16 %0, %1" : "=v"(b01) : "v"(p01));
                        asm("v_sat_pk_u8_i16 %0, %1" : "=v"(b23) : "v"(p23));
                        hw[q] = (int)__builtin_amdgcn_perm(b23, b01, 0x05040100u);
                    }
                    *(lds_v4i *)(size_t)(ha + nt * (MF_BM * 32) + mt * 1024) = hw;
                    asm volatile("ds_max_i32 %0, %1" ::"v"(sm_lds + SR_MAX + (b * MF_BM + mt * 32 + l31) * 4), "v"(min(mx, 255)) : "memory");
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                rs_signal(fl + 4 * (SR_F_XF + b));
                rs_signal(fl + 4 * (SR_F_H + b));
                if (wave == 0 && i + 2 < nmine) {
                    rs_wait(fl + 4 * (SR_F_XF + b), 8u * (unsigned)((i >> 1) + 1));
                    issue_x(i + 2, b);
                }
            }
        };
        if (fastrq) produce(std::true_type{});
        else produce(std::false_type{});
    } else {
        // =========================================================================================== consumers
        const int cw = wave - 8;                                       // fc2: waves cw < 6 own (channel tile cw >> 1, token tile cw & 1)
        const int hw = cw * 2 + half, l32 = l31;
        v4i w2f[12];                                    // fc2: this wave's channel tile (cw >> 1), all of K
#pragma unroll
        for (int kc = 0; kc < 12; ++kc)
            w2f[kc] = *reinterpret_cast<const v4i *>(p.w2 + (size_t)((cw < 6 ? cw >> 1 : 0) * 32 + rs_chan_of_row(l31)) * MF_HD + kc * 32 + half * 16);
        auto consume = [&](auto use_fast) __attribute__((always_inline)) {
            for (int i = 0; i < nmine; ++i) {
                const int b = i & 1;
                const long long tile = blockIdx.x + (long long)i * gridDim.x;
                const int nt = cw >> 1, mt = cw & 1;
                const long long tok = tile * MF_BM + mt * 32 + l31;
                v4i r0 = {0, 0, 0, 0}, r1 = {0, 0, 0, 0};
                if (cw < 6) {                                          // identity rows: requested now, consumed after fc2
                    const int16_t *rp = p.residual + min(tok, p.M - 1) * MF_C + nt * 32 + half * 16;
                    r0 = *reinterpret_cast<const v4i *>(rp);
                    r1 = *reinterpret_cast<const v4i *>(rp + 8);
                }
                rs_wait(fl + 4 * (SR_F_H + b), 8u * (unsigned)((i >> 1) + 1));
                // ---- ShiftGELU (+ qact1) in place: half-wave hw takes tokens hw, hw + 16, hw + 32, hw + 48; a token's 96 dwords =
                // 3 per lane (dword d of the row sits at [d / 8][token][d % 8])
                {
                    v2i line[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int qb = *(lds_i32 *)(size_t)(sm_lds + SR_MAX + (b * MF_BM + hw + 16 * j) * 4);
                        line[j] = reinterpret_cast<const v2i *>(p.tab + (size_t)qb * 256)[l32];
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int t = hw + 16 * j;
                        if (l32 == 0) *(lds_i32 *)(size_t)(sm_lds + SR_MAX + (b * MF_BM + t) * 4) = (int)0x80000000;
                        const unsigned base = sm_lds + SR_TAB + (hw * 2 + (j & 1)) * 256;
                        *(lds_v2i *)(size_t)(base + l32 * 8) = line[j];
                        const unsigned ra = sm_lds + SR_H + b * (MF_BM * MF_HD) + t * 32 + (l32 >> 3) * (MF_BM * 32) + (l32 & 7) * 4;
                        unsigned w[3], g[3][4];
#pragma unroll
                        for (int m = 0; m < 3; ++m) w[m] = *(lds_u32 *)(size_t)(ra + m * 4 * (MF_BM * 32));
#pragma unroll
                        for (int m = 0; m < 3; ++m) {
                            unsigned a0, a1, a2, a3;
                            asm("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "=v"(a0) : "v"(w[m]), "v"(base));
                            asm("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(a1) : "v"(w[m]), "v"(base));
                            asm("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(a2) : "v"(w[m]), "v"(base));
                            asm("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(a3) : "v"(w[m]), "v"(base));
                            asm volatile("ds_read_u8 %0, %1" : "=v"(g[m][0]) : "v"(a0) : "memory");
                            asm volatile("ds_read_u8 %0, %1" : "=v"(g[m][1]) : "v"(a1) : "memory");
                            asm volatile("ds_read_u8_d16_hi %0, %1" : "=v"(g[m][2]) : "v"(a2) : "memory");     // byte << 16, low half zeroed (SRAM-ECC d16)
                            asm volatile("ds_read_u8_d16_hi %0, %1" : "=v"(g[m][3]) : "v"(a3) : "memory");
                        }
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                        for (int m = 0; m < 3; ++m) {
                            unsigned o, t13;
                            asm volatile("v_or_b32 %0, %1, %2" : "=v"(t13) : "v"(g[m][1]), "v"(g[m][3]));      // behind the wait
                            asm volatile("v_or3_b32 %0, %1, %2, %3" : "=v"(o) : "v"(g[m][0]), "v"(g[m][2]), "v"(t13 << 8));
                            *(lds_u32 *)(size_t)(ra + m * 4 * (MF_BM * 32)) = o;
                        }
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                rs_signal(fl + 4 * SR_F_G);
                rs_wait(fl + 4 * SR_F_G, 8u * (unsigned)(i + 1));
                if (cw >= 6) { rs_signal(fl + 4 * SR_F_F); continue; }
                // ---- fc2 + qact2 (16 bit) + qact4 with the identity branch: channels 32 nt + 16 half + v of token tok
                v16i acc;                 // (two accumulator chains over even / odd k-steps measured slower: 280 vs 264 us, and spilled)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const v4i b4 = *reinterpret_cast<const v4i *>(sB2 + nt * 32 + half * 16 + q * 4);
                    acc[4 * q] = b4[0]; acc[4 * q + 1] = b4[1]; acc[4 * q + 2] = b4[2]; acc[4 * q + 3] = b4[3];
                }
                const unsigned ga = sm_lds + SR_H + b * (MF_BM * MF_HD) + (mt * 32 + l31) * 32 + half * 16;
#pragma unroll
                for (int kc = 0; kc < 12; ++kc) {
                    const v4i gf = *(lds_v4i *)(size_t)(ga + kc * (MF_BM * 32));
                    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(w2f[kc], gf, acc, 0, 0, 0);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                rs_signal(fl + 4 * SR_F_F);                            // the hidden buffer is free for tile i + 2
                v4i o0, o1;
#pragma unroll
                for (int d = 0; d < 8; ++d) {
                    const v2d c2 = *reinterpret_cast<const v2d *>(sC2 + nt * 32 + half * 16 + 2 * d);
                    const int rw = d < 4 ? r0[d] : r1[d - 4];
                    int o[2];
#pragma unroll
                    for (int f = 0; f < 2; ++f) {
                        const double t = (double)acc[2 * d + f] * c2[f];
                        const int t16 = min(max(decltype(use_fast)::value ? __double2loint(t + 6755399441055744.0) : rint_sat_i32(t), -32768), 32767);
                        const int r = f ? (rw >> 16) : (int)(short)(rw & 0xffff);
                        // both terms are integers < 2^31: the sum is the reference's fp64 sum (quant_utils.py:238-244)
                        o[f] = res_fast ? rq_fast(r, cr) + rq_fast(t16, cm) : rq_lean_wide(r, cr) + rq_lean_wide(t16, cm);
                    }
                    int pk;
                    asm("v_cvt_pk_i16_i32 %0, %1, %2" : "=v"(pk) : "v"(o[0]), "v"(o[1]));      // clamp to 16 bits and pack
                    if (d < 4) o0[d] = pk; else o1[d - 4] = pk;
                }
                if (tok < p.M) {
                    int16_t *op = p.out + tok * MF_C + nt * 32 + half * 16;
                    *reinterpret_cast<v4i *>(op) = o0;
                    *reinterpret_cast<v4i *>(op + 8) = o1;
                }
            }
        };
        if (fastrq) consume(std::true_type{});
        else consume(std::false_type{});
    }
}
