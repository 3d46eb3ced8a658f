// ivit_gemm2.h — the production int8 GEMM for the QuantLinear layers (K % 64 == 0).
//
//   C = A (M x K int8) * W^T (W: N x K int8) + bias, fused requant epilogues.
//
// Block tile 256 x 128 x 64, 512 threads = 8 waves (4 x 2), each wave 64 x 64 as 2x2
// v_mfma_i32_32x32x32_i8.  Operand tiles go HBM/L2 -> LDS with global_load_lds_dwordx4
// (no VGPR round trip) into a 3-stage ring; waits are counted (s_waitcnt vmcnt(3)) so the
// next tile's loads stay in flight across the single s_barrier per K step.  The LDS image
// is XOR-swizzled through the per-lane SOURCE address (the DMA destination is lane-linear),
// so every MFMA-fragment ds_read_b128 is bank-conflict-free.  Tiles that share an A panel
// are placed on the same XCD (private L2) by a bijective block-id remap.
//
// Epilogue: the MFMA operand roles are swapped (weights = "A", activations = "B"), so a
// lane holds ONE token and 4 CONSECUTIVE channels per register quad: four requantised
// values pack into a dword for free.  Requant is the reference arithmetic
// rne((double(z)*m)*2^-e) (quant_utils.py:229-230) evaluated as rne(double(z)*c) with
// c = m*2^-e, which is exact in fp64 (m integer < 2^32, power-of-two scaling) — one
// v_mul_f64, bit-identical.  (Measured on MI355X, tools/ubench/valu_rates.hip: v_mul_f64,
// v_rndne_f64, v_cvt_* all cost ~1.7x a v_mul_f32, so an fp32 "fast path" buys nothing.)
// Packed dwords are staged in LDS and leave as whole 128/256-byte rows (16 B per lane).
#pragma once
#include <type_traits>
#include "ivit_gemm.h"

#define G2_BN 128
#ifndef G2_NSTAGE128
#define G2_NSTAGE128 2
#endif
#define G2_BK 64
// BM = 256 (8 waves, 2 blocks/CU) or 128 (4 waves, 3 blocks/CU: finer tiles for narrow-N GEMMs
// whose 256-row tiling leaves the last block wave nearly empty, and more blocks in flight to
// cover the cold-start latency of short-K tiles)
template <int BM> struct G2Cfg {
    static constexpr int THREADS = BM * 2;
    static constexpr int STAGE = BM * 64 + 8192;       // A tile + B tile
    static constexpr int NSTAGE = (BM == 256) ? 3 : G2_NSTAGE128;
    static constexpr int SMEM = (NSTAGE * STAGE > BM * 264) ? NSTAGE * STAGE : BM * 264;   // ring, aliased by the staging tile
    static constexpr int B_PER_THREAD = 512 / THREADS; // B chunks per thread
};
#define G2_LD8 136              // staged int8 row stride (bytes): 2-way-free ds_write_b32
#define G2_LD16 264             // staged int16 row stride (bytes): conflict-free ds_write_b64

__device__ __forceinline__ int rq_lean(int z, double c, int lo, int hi) {
    int v = rint_sat_i32((double)z * c);
    return min(max(v, lo), hi);
}
__device__ __forceinline__ int rq_lean_wide(int z, double c) { return rint_sat_i32((double)z * c); }

// kchunks: valid 16-byte chunks of this K step (4, or 2 for the 32-wide tail step when K % 64 == 32): lanes whose
// chunk lies beyond K are masked off — their LDS slots keep stale bytes that the tail step never feeds to an MFMA
// abase[i]: this thread's two A rows (row i * THREADS / 4 + tid / 4 of the tile) — the row's first byte, or (IM2COL) the top-left pixel of
// its patch in channel 0; hw = H * W, w = W of the images
template <int BM, bool IM2COL = false>
__device__ __forceinline__ void g2_issue(const int8_t *const (&abase)[2], const int8_t *B, int ldb, int N, int col0, int k0, char *stage,
                                         int tid, int kchunks = 4, long long hw = 0, int w = 0) {
    using Cf = G2Cfg<BM>;
    const int wave = tid >> 6;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int id = tid + i * Cf::THREADS, row = id >> 2, pos = id & 3;
        int c = pos ^ ((row >> 2) & 3);
        const int q = (k0 >> 4) + c;                    // IM2COL: chunk q of the row = channel q / 16, pixel row q % 16 of the patch
        const int8_t *src = IM2COL ? abase[i] + (long long)(q >> 4) * hw + (q & 15) * w : abase[i] + k0 + c * 16;
        unsigned loff = __builtin_amdgcn_readfirstlane((unsigned)(i * (Cf::THREADS * 16) + wave * 1024));
        if (c < kchunks)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(stage + loff), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < Cf::B_PER_THREAD; ++i) {
        int id = tid + i * Cf::THREADS, row = id >> 2, pos = id & 3;
        int c = pos ^ ((row >> 2) & 3);
        int grow = min(col0 + row, N - 1);
        const int8_t *src = B + (long long)grow * ldb + k0 + c * 16;
        unsigned loff = __builtin_amdgcn_readfirstlane((unsigned)(BM * 64 + i * (Cf::THREADS * 16) + wave * 1024));
        if (c < kchunks)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(stage + loff), 16, 0, 0);
    }
}

template <int EPI, int BM, bool IM2COL = false>
__global__ __launch_bounds__(BM * 2, BM == 256 ? 4 : (G2_NSTAGE128 == 2 ? 4 : 3)) void gemm_glds_kernel(GemmArgs p) {
    using Cf = G2Cfg<BM>;
    constexpr int G2_BM = BM, G2_STAGE = Cf::STAGE, G2_SMEM = Cf::SMEM, NT = Cf::THREADS;
    constexpr int NLOADS = 2 + Cf::B_PER_THREAD;       // DMA loads per thread per K step
    constexpr int NS = Cf::NSTAGE;                     // ring depth: loads run NS-1 steps ahead
    __shared__ __attribute__((aligned(16))) char smem[G2_SMEM + 1536 + 512];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, half = lane >> 5;

    // XCD-aware, bijective block -> tile map: each XCD owns a contiguous run of tiles,
    // n-fastest, so the blocks that share an A panel share one L2.
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xq = nwg >> 3, xr = nwg & 7, xcd = bid & 7, xi = bid >> 3;
    const int wg = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + xi;
    const int tile_m = wg / p.tiles_n, tile_n = wg - tile_m * p.tiles_n;
    const int row0 = tile_m * G2_BM, col0 = tile_n * G2_BN;

    const int8_t *A = reinterpret_cast<const int8_t *>(p.A);
    const int8_t *B = p.B;

    const int Kdim = p.K;
    const int nk = (Kdim + G2_BK - 1) / G2_BK;
    const bool ktail = (Kdim % G2_BK) != 0;            // K % 64 == 32: the last step carries 32 columns
    auto kch = [&](int kt) { return (ktail && kt == nk - 1) ? 2 : 4; };
    const int8_t *abase[2];
    const long long img_hw = IM2COL ? (long long)p.img_H * p.img_W : 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int grow = min(row0 + ((tid + i * NT) >> 2), p.M - 1);
        if (IM2COL) {
            const int b = grow / p.pe_P, rem = grow - b * p.pe_P, gy = rem / p.pe_gw, gx = rem - gy * p.pe_gw;
            abase[i] = p.img + ((long long)b * p.img_C * p.img_H + gy * 16) * p.img_W + gx * 16;
        } else {
            abase[i] = A + (long long)grow * p.lda;
        }
    }
    g2_issue<BM, IM2COL>(abase, B, p.ldb, p.N, col0, 0, smem, tid, kch(0), img_hw, p.img_W);
    if (NS == 3 && nk > 1) g2_issue<BM, IM2COL>(abase, B, p.ldb, p.N, col0, G2_BK, smem + G2_STAGE, tid, kch(1), img_hw, p.img_W);

    // per-channel constants of this column block -> LDS (read back in the epilogue; the
    // K-loop barriers order the write): c[n] = m*2^-e (exact in fp64), bias[n]
    double *sC = reinterpret_cast<double *>(smem + G2_SMEM);
    int *sBias = reinterpret_cast<int *>(smem + G2_SMEM + 1024);
    int *sUnsafe = reinterpret_cast<int *>(smem + G2_SMEM + 1536);   // [G2_BN] per-channel flags
    if (tid < G2_BN) {
        const int ch = col0 + tid;
        const bool in = ch < p.N;
        const double cv = in ? p.dy_ch[ch].m * p.dy_ch[ch].r : 0.0;
        const int bs = (in && p.bias) ? p.bias[ch] : 0;
        sC[tid] = cv;
        sBias[tid] = bs;
        // magic-number rounding in the epilogue needs |(acc + bias) * c| < 2^31; |acc| <= K * 2^14
        sUnsafe[tid] = !(fabs(cv) * ((double)Kdim * 16384.0 + fabs((double)bs)) < 2147483000.0);
    }

    // acc[i][j]: C^T sub-tiles — lane holds token (lane&31) of m-tile i and, per register
    // quad g, the 4 consecutive channels 32j + 8g + 4*half + (0..3)
    v16i acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;

    v4i a[2], b[2];
    for (int kt = 0; kt < nk; ++kt) {
        if (NS == 3 && kt + 1 < nk) {
            if (NLOADS == 3) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        }
        // lgkmcnt(0): a raw s_barrier does not wait for this wave's own LDS reads.  The fragment reads of the
        // previous step must have RETURNED before any wave may start the DMA that overwrites their buffer —
        // the scheduler is free to sink their first use (and with it the implicit wait) below the barrier.
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt + NS - 1 < nk)
            g2_issue<BM, IM2COL>(abase, B, p.ldb, p.N, col0, (kt + NS - 1) * G2_BK,
                                 smem + ((kt + NS - 1) % NS) * G2_STAGE, tid, kch(kt + NS - 1), img_hw, p.img_W);
        const char *sA = smem + (kt % NS) * G2_STAGE;
        const char *sB = sA + BM * 64;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            if (kk == 1 && ktail && kt == nk - 1) break;   // 32-wide tail step
            const int chunk = kk * 2 + half;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = *reinterpret_cast<const v4i *>(sA + lds_off(wm * 64 + i * 32 + (lane & 31), chunk));
                b[i] = *reinterpret_cast<const v4i *>(sB + lds_off(wn * 64 + i * 32 + (lane & 31), chunk));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(b[j], a[i], acc[i][j], 0, 0, 0);
        }
    }
    asm volatile("" ::: "memory");
    if (p.dbg == 1) {   // ablation: main loop only
        int sacc = 0;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc ^= acc[i][j][r];
        if (sacc == 0x12345678) reinterpret_cast<int *>(p.out)[tid] = sacc;
        return;
    }
    __syncthreads();   // every wave done with the ring before it is reused as the staging tile

    constexpr bool OUT8 = (EPI == EPI_RQ8_CH || EPI == EPI_QKV);
    constexpr bool CLAMP8 = OUT8 || EPI == EPI_RQ8W16_CH;          // RQ8W16: the 8-bit clamp, staged and stored as 16-bit
    constexpr int OLO = CLAMP8 ? -128 : -32768, OHI = CLAMP8 ? 127 : 32767;
    // ---- phase 1: requant + pack 4 channels per lane -> staged tile [token][channel].
    // rne(fl64(z*c)) as loint(fl64(z*c) + 1.5*2^52): the reference's two roundings (quant_utils.py:229-231),
    // valid while |z*c| < 2^31 — checked per channel when the constants were staged; else the rint form.
    const bool fastrq = !__any(sUnsafe[wn * 64 + lane] != 0);   // this wave's 64 channels
    auto phase1 = [&](auto use_fast) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nl = wn * 64 + j * 32 + g * 8 + half * 4;
                double c[4];
                int bs[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) { c[e] = sC[nl + e]; bs[e] = sBias[nl + e]; }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int ml = wm * 64 + i * 32 + (lane & 31);
                    int o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int z = acc[i][j][g * 4 + e] + bs[e];
                        if (p.dbg == 3) { o[e] = z; continue; }   // ablation: no requant math
                        const double t = (double)z * c[e];
                        const int v = decltype(use_fast)::value ? __double2loint(t + 6755399441055744.0) : rint_sat_i32(t);
                        o[e] = min(max(v, OLO), OHI);
                    }
                    if (OUT8) {
                        unsigned w01 = __builtin_amdgcn_perm((unsigned)o[1], (unsigned)o[0], 0x0c0c0400u);
                        unsigned w23 = __builtin_amdgcn_perm((unsigned)o[3], (unsigned)o[2], 0x0c0c0400u);
                        *reinterpret_cast<unsigned *>(smem + ml * G2_LD8 + nl) = __builtin_amdgcn_perm(w23, w01, 0x05040100u);
                    } else {
                        v2i w = {(int)__builtin_amdgcn_perm((unsigned)o[1], (unsigned)o[0], 0x05040100u),
                                 (int)__builtin_amdgcn_perm((unsigned)o[3], (unsigned)o[2], 0x05040100u)};
                        *reinterpret_cast<v2i *>(smem + ml * G2_LD16 + nl * 2) = w;
                    }
                }
            }
    };
    if (fastrq) phase1(std::true_type{});
    else phase1(std::false_type{});
    __syncthreads();
    if (p.dbg == 2) return;   // ablation: no phase 2 (no global stores)

    // ---- phase 2: whole rows out, 16 bytes per lane
    if (EPI == EPI_RQ8_CH) {
        int8_t *out = reinterpret_cast<int8_t *>(p.out);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int id = tid + i * NT, row = id >> 3, c = id & 7;
            int grow = row0 + row, gcol = col0 + c * 16;
            if (grow < p.M && gcol < p.N) {
                const char *sp = smem + row * G2_LD8 + c * 16;
                v2i lo = *reinterpret_cast<const v2i *>(sp), hi = *reinterpret_cast<const v2i *>(sp + 8);
                v4i v = {lo[0], lo[1], hi[0], hi[1]};
                int8_t *dst = out + (long long)grow * p.ldc + gcol;
                if (gcol + 16 <= p.N && ((p.ldc & 15) == 0)) {
                    *reinterpret_cast<v4i *>(dst) = v;
                } else {
                    for (int e = 0; e < 16 && gcol + e < p.N; ++e) dst[e] = (int8_t)(v[e >> 2] >> (8 * (e & 3)));
                }
            }
        }
    } else if (EPI == EPI_RQ16_CH || EPI == EPI_RQ16_CH_RES || EPI == EPI_RQ8W16_CH) {
        int16_t *out = reinterpret_cast<int16_t *>(p.out);
        const double cm = p.dy_main.m * p.dy_main.r, cr = p.dy_res.m * p.dy_res.r;
        const bool res_fast = fabs(cm) < RQ_FAST_CLIM && fabs(cr) < RQ_FAST_CLIM;   // |int16 * c| < 2^24
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int id = tid + i * NT, row = id >> 4, c = id & 15;
            int grow = row0 + row, gcol = col0 + c * 8;
            if (grow < p.M && gcol < p.N) {
                const char *sp = smem + row * G2_LD16 + c * 16;
                v2i lo = *reinterpret_cast<const v2i *>(sp), hi = *reinterpret_cast<const v2i *>(sp + 8);
                v4i v = {lo[0], lo[1], hi[0], hi[1]};
                // (patch embedding: output row behind its image's class-token row, residual = the patch's position embedding)
                const int pimg = IM2COL ? grow / p.pe_P : 0;
                int16_t *dst = out + (long long)(IM2COL ? grow + pimg + 1 : grow) * p.ldc + gcol;
                const bool vec = (gcol + 8 <= p.N) && ((p.ldc & 7) == 0);
                if (EPI == EPI_RQ16_CH_RES) {
                    const int16_t *rp = p.residual + (long long)(IM2COL ? grow - pimg * p.pe_P + 1 : grow) * p.ldc + gcol;
                    v4i rs = {0, 0, 0, 0};
                    if (vec) {
                        rs = *reinterpret_cast<const v4i *>(rp);
                    } else {
                        for (int e = 0; e < 8 && gcol + e < p.N; ++e)
                            rs[e >> 1] |= ((int)(unsigned short)rp[e]) << (16 * (e & 1));
                    }
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        int t0 = (int)(short)(v[w] & 0xffff), t1 = v[w] >> 16;
                        int r0 = (int)(short)(rs[w] & 0xffff), r1 = rs[w] >> 16;
                        // both terms are integers < 2^31: the sum is the reference's fp64 sum
                        int o0, o1;
                        if (__builtin_expect(res_fast, 1)) {
                            o0 = rq_fast(r0, cr) + rq_fast(t0, cm);
                            o1 = rq_fast(r1, cr) + rq_fast(t1, cm);
                        } else {
                            o0 = rq_lean_wide(r0, cr) + rq_lean_wide(t0, cm);
                            o1 = rq_lean_wide(r1, cr) + rq_lean_wide(t1, cm);
                        }
                        o0 = min(max(o0, -32768), 32767);
                        o1 = min(max(o1, -32768), 32767);
                        v[w] = (o0 & 0xffff) | (o1 << 16);
                    }
                }
                if (vec) {
                    *reinterpret_cast<v4i *>(dst) = v;
                } else {
                    for (int e = 0; e < 8 && gcol + e < p.N; ++e) dst[e] = (int16_t)(v[e >> 1] >> (16 * (e & 1)));
                }
            }
        }
    } else if (EPI == EPI_QKV) {
        const float rcpT = 1.0f / (float)p.T;
        // rows-fastest mapping: a wave covers 64 consecutive tokens of one 16-channel chunk
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int id = tid + i * NT, row = id & (BM - 1), c = id / BM;
            int grow = row0 + row, gcol = col0 + c * 16;
            if (grow < p.M && gcol < p.N) {
                const char *sp = smem + row * G2_LD8 + c * 16;
                v2i lo = *reinterpret_cast<const v2i *>(sp), hi = *reinterpret_cast<const v2i *>(sp + 8);
                v4i v = {lo[0], lo[1], hi[0], hi[1]};
                int which = gcol / p.D, within = gcol - which * p.D;
                int head = within / p.dh, d0 = within - head * p.dh;
                // token -> (image, position): float reciprocal estimate + one correction step (grow < 2^23)
                int b = (int)((float)grow * rcpT);
                int t = grow - b * p.T;
                if (t < 0) { --b; t += p.T; }
                if (t >= p.T) { ++b; t -= p.T; }
                long long bh = (long long)b * p.H + head;
                if (which < 2 || p.ldv == 0) {
                    int8_t *dst = (which == 0 ? p.q : (which == 1 ? p.k : p.vt)) + (bh * p.T + t) * p.dh + d0;
                    *reinterpret_cast<v4i *>(dst) = v;
                } else {
                    int8_t *dst = p.vt + (bh * p.dh + d0) * p.ldv + t;
#pragma unroll
                    for (int e = 0; e < 16; ++e) dst[(long long)e * p.ldv] = (int8_t)(v[e >> 2] >> (8 * (e & 3)));
                }
            }
        }
    }
}
