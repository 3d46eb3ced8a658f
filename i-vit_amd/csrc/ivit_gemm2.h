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
// Requant: the reference arithmetic is rne((double(z)*m)*2^-e) (quant_utils.py:229-230).
// The epilogue evaluates it through an fp32 path with a proven error bound and falls back
// to the fp64 sequence whenever the bound cannot decide the rounding (near-ties, |z| >=
// 2^24) — the result is always identical to the fp64 sequence.
#pragma once
#include "ivit_gemm.h"

#define G2_BM 256
#define G2_BN 128
#define G2_BK 64
#define G2_STAGE 24576          // 256*64 (A) + 128*64 (B)
#define G2_SMEM (3 * G2_STAGE)  // 73728 >= 256*272 (int16 staging)

// ---- exact requant with fp32 fast paths -------------------------------------
struct RqF { float chi, clo; };

__device__ __forceinline__ RqF rqf_make(double m, double r) {
    double c = m * r;
    RqF f;
    f.chi = (float)c;
    f.clo = (float)(c - (double)f.chi);
    return f;
}

// 8-bit: p = fl(z*chi) carries < |y|*1.2e-7 error; for |p| <= 200 that is < 2.4e-5, so
// rint(p) == rne(y) whenever p is farther than 1e-4 from a tie; |p| > 200 clamps anyway.
__device__ __forceinline__ int rq8_exact(int z, RqF f, double m, double r) {
    float zf = (float)z;
    float p = zf * f.chi;
    float rp = rintf(p);
    float d = fabsf(p - rp);
    bool ok = ((d < 0.4999f) || (fabsf(p) > 200.0f)) && ((unsigned)(z + (1 << 24)) < (1u << 25));
    int v = (int)fminf(fmaxf(rp, -128.0f), 127.0f);
    if (!ok) v = clamp_b<8>(rq_f64((double)z, m, r));
    return v;
}

// 16-bit: two-term product, y = p + e2 with |error| < |y|*2^-44; decide the rounding from
// t = (p - rint(p)) + e2 unless t is within 1e-6 of +-0.5 (then the fp64 sequence decides).
__device__ __forceinline__ int rq16_exact(int z, RqF f, double m, double r) {
    float zf = (float)z;
    float p = zf * f.chi;
    float e1 = __builtin_fmaf(zf, f.chi, -p);
    float e2 = __builtin_fmaf(zf, f.clo, e1);
    float rp = rintf(p);
    float t = (p - rp) + e2;
    float at = fabsf(t);
    bool ok = ((fabsf(at - 0.5f) > 1e-6f) || (fabsf(p) > 40000.0f)) && ((unsigned)(z + (1 << 24)) < (1u << 25));
    float adj = at > 0.5f ? (t > 0.f ? 1.0f : -1.0f) : 0.0f;
    int v = (int)fminf(fmaxf(rp + adj, -32768.0f), 32767.0f);
    if (!ok) v = clamp_b<16>(rq_f64((double)z, m, r));
    return v;
}

// unclamped variant for the two terms of the residual add (|result| < 2^22)
__device__ __forceinline__ int rq16_wide(int z, RqF f, double m, double r) {
    float zf = (float)z;
    float p = zf * f.chi;
    float e1 = __builtin_fmaf(zf, f.chi, -p);
    float e2 = __builtin_fmaf(zf, f.clo, e1);
    float rp = rintf(p);
    float t = (p - rp) + e2;
    float at = fabsf(t);
    bool ok = (fabsf(at - 0.5f) > 1e-6f) && (fabsf(p) < 4194304.0f) && ((unsigned)(z + (1 << 24)) < (1u << 25));
    float adj = at > 0.5f ? (t > 0.f ? 1.0f : -1.0f) : 0.0f;
    int v = (int)(rp + adj);
    if (!ok) v = clamp_b<32>(rq_f64((double)z, m, r));
    return v;
}

__device__ __forceinline__ void g2_issue(const int8_t *A, const int8_t *B, int lda, int ldb, int M, int N,
                                         int row0, int col0, int k0, char *stage, int tid) {
    const int wave = tid >> 6;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int id = tid + i * 512, row = id >> 2, pos = id & 3;
        int c = pos ^ ((row >> 2) & 3);
        int grow = min(row0 + row, M - 1);
        const int8_t *src = A + (long long)grow * lda + k0 + c * 16;
        unsigned loff = __builtin_amdgcn_readfirstlane((unsigned)(i * 8192 + wave * 1024));
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)(stage + loff), 16, 0, 0);
    }
    {
        int id = tid, row = id >> 2, pos = id & 3;
        int c = pos ^ ((row >> 2) & 3);
        int grow = min(col0 + row, N - 1);
        const int8_t *src = B + (long long)grow * ldb + k0 + c * 16;
        unsigned loff = __builtin_amdgcn_readfirstlane((unsigned)(16384 + wave * 1024));
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)(stage + loff), 16, 0, 0);
    }
}

template <int EPI>
__global__ __launch_bounds__(512) void gemm_glds_kernel(GemmArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[G2_SMEM];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware, bijective block -> tile map: each XCD owns a contiguous run of tiles,
    // n-fastest, so the blocks that share an A panel share one L2.
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xq = nwg >> 3, xr = nwg & 7, xcd = bid & 7, xi = bid >> 3;
    const int wg = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + xi;
    const int tile_m = wg / p.tiles_n, tile_n = wg - tile_m * p.tiles_n;
    const int row0 = tile_m * G2_BM, col0 = tile_n * G2_BN;

    const int8_t *A = reinterpret_cast<const int8_t *>(p.A);
    const int8_t *B = p.B;

    v16i acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;

    const int nk = p.K / G2_BK;
    g2_issue(A, B, p.lda, p.ldb, p.M, p.N, row0, col0, 0, smem, tid);
    if (nk > 1) g2_issue(A, B, p.lda, p.ldb, p.M, p.N, row0, col0, G2_BK, smem + G2_STAGE, tid);

    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt + 2 < nk)
            g2_issue(A, B, p.lda, p.ldb, p.M, p.N, row0, col0, (kt + 2) * G2_BK,
                     smem + ((kt + 2) % 3) * G2_STAGE, tid);
        const char *sA = smem + (kt % 3) * G2_STAGE;
        const char *sB = sA + 16384;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int chunk = kk * 2 + (lane >> 5);
            v4i a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = *reinterpret_cast<const v4i *>(sA + lds_off(wm * 64 + i * 32 + (lane & 31), chunk));
                b[i] = *reinterpret_cast<const v4i *>(sB + lds_off(wn * 64 + i * 32 + (lane & 31), chunk));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
    asm volatile("" ::: "memory");
    __syncthreads();   // every wave done with the ring before it is reused as the staging tile

    // ---- phase 1: per-lane requant of the accumulator fragments -> staged tile
    double dm[2] = {0, 0}, dr[2] = {0, 0};
    RqF fq[2];
    int bias[2] = {0, 0};
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        int col = min(col0 + wn * 64 + j * 32 + (lane & 31), p.N - 1);
        if (p.bias) bias[j] = p.bias[col];
        dm[j] = p.dy_ch[col].m;
        dr[j] = p.dy_ch[col].r;
        fq[j] = rqf_make(dm[j], dr[j]);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int rl = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                int cl = wn * 64 + j * 32 + (lane & 31);
                int v = acc[i][j][r] + bias[j];
                if (EPI == EPI_RQ16_CH || EPI == EPI_RQ16_CH_RES) {
                    int o = rq16_exact(v, fq[j], dm[j], dr[j]);
                    *reinterpret_cast<int16_t *>(smem + rl * GEMM_SC16_LD + cl * 2) = (int16_t)o;
                } else {
                    int o = rq8_exact(v, fq[j], dm[j], dr[j]);
                    *reinterpret_cast<int8_t *>(smem + rl * GEMM_SC8_LD + cl) = (int8_t)o;
                }
            }
    __syncthreads();

    // ---- phase 2: coalesced write-out
    if (EPI == EPI_RQ8_CH) {
        int8_t *out = reinterpret_cast<int8_t *>(p.out);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int id = tid + i * 512, row = id >> 3, c = id & 7;
            int grow = row0 + row, gcol = col0 + c * 16;
            if (grow < p.M && gcol < p.N) {
                v4i v = *reinterpret_cast<const v4i *>(smem + row * GEMM_SC8_LD + c * 16);
                int8_t *dst = out + (long long)grow * p.ldc + gcol;
                if (gcol + 16 <= p.N && ((p.ldc & 15) == 0)) {
                    *reinterpret_cast<v4i *>(dst) = v;
                } else {
                    for (int e = 0; e < 16 && gcol + e < p.N; ++e) dst[e] = (int8_t)(v[e >> 2] >> (8 * (e & 3)));
                }
            }
        }
    } else if (EPI == EPI_RQ16_CH || EPI == EPI_RQ16_CH_RES) {
        int16_t *out = reinterpret_cast<int16_t *>(p.out);
        const RqF fm = rqf_make(p.dy_main.m, p.dy_main.r), fr = rqf_make(p.dy_res.m, p.dy_res.r);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int id = tid + i * 512, row = id >> 4, c = id & 15;
            int grow = row0 + row, gcol = col0 + c * 8;
            if (grow < p.M && gcol < p.N) {
                v4i v = *reinterpret_cast<const v4i *>(smem + row * GEMM_SC16_LD + c * 16);
                int16_t *dst = out + (long long)grow * p.ldc + gcol;
                const bool vec = (gcol + 8 <= p.N) && ((p.ldc & 7) == 0);
                if (EPI == EPI_RQ16_CH_RES) {
                    const int16_t *rp = p.residual + (long long)grow * p.ldc + gcol;
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
                        // both terms are integers < 2^17: their sum is exact in int
                        int o0 = rq16_wide(r0, fr, p.dy_res.m, p.dy_res.r) + rq16_wide(t0, fm, p.dy_main.m, p.dy_main.r);
                        int o1 = rq16_wide(r1, fr, p.dy_res.m, p.dy_res.r) + rq16_wide(t1, fm, p.dy_main.m, p.dy_main.r);
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
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int id = tid + i * 512, row = id & 255, c = id >> 8;
            int grow = row0 + row, gcol = col0 + c * 16;
            if (grow < p.M && gcol < p.N) {
                v4i v = *reinterpret_cast<const v4i *>(smem + row * GEMM_SC8_LD + c * 16);
                int which = gcol / p.D, within = gcol - which * p.D;
                int head = within / p.dh, d0 = within - head * p.dh;
                int b = grow / p.T, t = grow - b * p.T;
                long long bh = (long long)b * p.H + head;
                if (which < 2) {
                    int8_t *dst = (which == 0 ? p.q : p.k) + (bh * p.T + t) * p.dh + d0;
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
