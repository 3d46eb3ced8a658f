// ivit_gemm.h — int8 MFMA "NT" GEMM for gfx950 with fused requant epilogues.
//
//   C[b] = A[b] (M x K, int8 or uint16) * B[b]^T (B[b]: N x K int8)  [+ bias]
//
// Block tile 128x128x64, 256 threads = 4 waves (2x2), each wave 64x64 as 2x2
// v_mfma_i32_32x32x32_i8.  Operands are staged global -> VGPR -> LDS (16-byte
// chunks, XOR-swizzled so ds_read_b128 of MFMA fragments is conflict-free), the
// next K tile's global loads are in flight while the current one is multiplied.
// The 8/16-bit epilogues requantise in fp64 (reference arithmetic), stage the tile
// in LDS and leave with coalesced 16-byte stores.
//
// uint16 A (Shiftmax probabilities, 0..32768): a - 16384 = 256*hi + lo with
// hi in [-64,64], lo in [-128,127]; two int8 MFMA passes over the same B fragment
// plus 16384 * sum_k B[n,k] (accumulated with v_dot4 on the B fragments).
#pragma once
#include "ivit_device.h"

enum {
    EPI_RAW32 = 0,       // int32 out = acc + bias
    EPI_RQ8_CH = 1,      // int8  = clamp8 (rq(acc + bias, dy_ch[n]))
    EPI_RQ16_CH = 2,     // int16 = clamp16(rq(acc + bias, dy_ch[n]))
    EPI_RQ16_CH_RES = 3, // int16 = clamp16(rq(clamp16(rq(acc+bias, dy_ch[n])), main) + rq(res, resd))
    EPI_RQ8_S = 4,       // int8  = clamp8 (rq(acc, main))
    EPI_QKV = 5,         // RQ8_CH then scatter to q,k [B,H,T,dh] and vT [B,H,dh,ldv]
    EPI_RQ8W16_CH = 6    // int16 = clamp8(rq(acc + bias, dy_ch[n])): an 8-bit QuantAct whose consumer reads the 16-bit stream
                         // (PatchMerging's reduction, swin_quant.py:343-349, in front of the next stage's blocks); gemm_glds_kernel only
};

struct GemmArgs {
    const void *A;
    const int8_t *B;
    int M, N, K;
    int lda, ldb, ldc;
    long long strideA, strideB;
    // output addressing: base = (z / inner) * sC_outer + (z % inner) * sC_inner
    int inner;
    long long sC_outer, sC_inner;
    const int32_t *bias;
    const ivit_dyadic *dy_ch;
    const double *cq;    // per-channel c = m * 2^-e, precomputed by the linear plan (gemm_ps_kernel)
    void *dummy;         // >= 1 KB of device scratch: where lanes outside the matrix store (gemm_as_kernel)
    ivit_dyadic dy_main, dy_res;
    const int16_t *residual;
    void *out;
    int8_t *q, *k, *vt;
    int T, H, dh, ldv, D;
    int tiles_n;
    int dbg;   // debug/ablation switch (env IVIT_GEMM_DBG), 0 in production
    // patch embedding in one launch (gemm_glds_kernel<EPI_RQ16_CH_RES, BM, IM2COL = true>, ivit_patch_embed): A rows are gathered from the
    // images (16 x 16 patches: every 16-byte chunk of an im2col row is one pixel row of the patch), row r of the GEMM is patch r % pe_P of image
    // r / pe_P, its residual row is pos[r % pe_P + 1] and its output row r + r / pe_P + 1 (the class-token rows are written by embed_finish_kernel)
    const int8_t *img;
    int img_C, img_H, img_W, pe_gw, pe_P;
};

#define GEMM_BM 128
#define GEMM_BN 128
#define GEMM_BK 64
#define GEMM_SC8_LD 144   // bytes per staged int8 row
#define GEMM_SC16_LD 272  // bytes per staged int16 row
#define GEMM_SMEM 34816   // max(3*8192, 128*272)

__device__ __forceinline__ int lds_off(int row, int chunk) {
    return row * GEMM_BK + ((chunk ^ ((row >> 2) & 3)) << 4);
}

__device__ __forceinline__ v4i mask_tail_bytes(v4i v, int valid) {  // keep first `valid` bytes
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        int nb = valid - d * 4;
        unsigned m = nb >= 4 ? 0xffffffffu : (nb <= 0 ? 0u : ((1u << (nb * 8)) - 1u));
        v[d] &= (int)m;
    }
    return v;
}

// load a 16-byte chunk of an int8 operand tile, zero outside [rows) x [K)
__device__ __forceinline__ v4i load_chunk_i8(const int8_t *base, int ld, int row, int nrows,
                                             int k, int K) {
    v4i v = {0, 0, 0, 0};
    if (row < nrows && k < K) {
        v = *reinterpret_cast<const v4i *>(base + (long long)row * ld + k);
        if (k + 16 > K) v = mask_tail_bytes(v, K - k);
    }
    return v;
}

template <bool A16, int EPI>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[GEMM_SMEM];
    char *sA = smem;                 // [128][64] int8 (lo plane when A16)
    char *sB = smem + 8192;          // [128][64] int8
    char *sA2 = smem + 16384;        // hi plane (A16)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tile_m = blockIdx.x / p.tiles_n, tile_n = blockIdx.x % p.tiles_n;
    const int z = blockIdx.y;
    const int row0 = tile_m * GEMM_BM, col0 = tile_n * GEMM_BN;

    const int8_t *Bz = p.B + z * p.strideB;
    const int8_t *A8 = reinterpret_cast<const int8_t *>(p.A) + (A16 ? 0 : z * p.strideA);
    const uint16_t *A16p = reinterpret_cast<const uint16_t *>(p.A) + (A16 ? z * p.strideA : 0);

    v16i acc[2][2], acc2[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0; acc2[i][j][r] = 0; }
    int bsum[2] = {0, 0};

    // staging registers
    v4i ra[A16 ? 4 : 2], rb[2];

    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int id = tid + i * 256, row = id >> 2, c = id & 3;
            rb[i] = load_chunk_i8(Bz, p.ldb, col0 + row, p.N, k0 + c * 16, p.K);
            if (!A16) ra[i] = load_chunk_i8(A8, p.lda, row0 + row, p.M, k0 + c * 16, p.K);
        }
        if (A16) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int id = tid + i * 256, row = id >> 3, c8 = id & 7;
                int k = k0 + c8 * 8;
                v4i v = {0x40004000, 0x40004000, 0x40004000, 0x40004000};  // 16384 -> (0,0)
                if (row0 + row < p.M && k < p.K) {
                    v4i t = *reinterpret_cast<const v4i *>(A16p + (long long)(row0 + row) * p.lda + k);
                    if (k + 8 > p.K) {
                        int valid = p.K - k;  // elements
#pragma unroll
                        for (int d = 0; d < 4; ++d) {
                            int e0 = 2 * d, e1 = 2 * d + 1;
                            unsigned w = (unsigned)t[d];
                            unsigned lo16 = e0 < valid ? (w & 0xffffu) : 0x4000u;
                            unsigned hi16 = e1 < valid ? (w >> 16) : 0x4000u;
                            t[d] = (int)(lo16 | (hi16 << 16));
                        }
                    }
                    v = t;
                }
                ra[i] = v;
            }
        }
    };

    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int id = tid + i * 256, row = id >> 2, c = id & 3;
            *reinterpret_cast<v4i *>(sB + lds_off(row, c)) = rb[i];
            if (!A16) *reinterpret_cast<v4i *>(sA + lds_off(row, c)) = ra[i];
        }
        if (A16) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int id = tid + i * 256, row = id >> 3, c8 = id & 7;
                unsigned lo[2] = {0, 0}, hi[2] = {0, 0};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    unsigned w = (unsigned)ra[i][e >> 1];
                    int a = (int)((e & 1) ? (w >> 16) : (w & 0xffffu)) - 16384;
                    int l = (int)(int8_t)(a & 0xff);
                    int hgh = (a - l) >> 8;
                    lo[e >> 2] |= (unsigned)(l & 0xff) << ((e & 3) * 8);
                    hi[e >> 2] |= (unsigned)(hgh & 0xff) << ((e & 3) * 8);
                }
                int off = lds_off(row, c8 >> 1) + (c8 & 1) * 8;
                *reinterpret_cast<v2i *>(sA + off) = v2i{(int)lo[0], (int)lo[1]};
                *reinterpret_cast<v2i *>(sA2 + off) = v2i{(int)hi[0], (int)hi[1]};
            }
        }
    };

    const int nk = (p.K + GEMM_BK - 1) / GEMM_BK;
    gload(0);
    for (int kt = 0; kt < nk; ++kt) {
        lstore();
        __syncthreads();
        if (kt + 1 < nk) gload((kt + 1) * GEMM_BK);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int chunk = kk * 2 + (lane >> 5);
            v4i a[2], a2[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                int ar = wm * 64 + i * 32 + (lane & 31);
                a[i] = *reinterpret_cast<const v4i *>(sA + lds_off(ar, chunk));
                if (A16) a2[i] = *reinterpret_cast<const v4i *>(sA2 + lds_off(ar, chunk));
                int br = wn * 64 + i * 32 + (lane & 31);
                b[i] = *reinterpret_cast<const v4i *>(sB + lds_off(br, chunk));
            }
            if (A16) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int d = 0; d < 4; ++d)
                        bsum[j] = __builtin_amdgcn_sdot4(b[j][d], 0x01010101, bsum[j], false);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i], b[j], acc[i][j], 0, 0, 0);
                    if (A16)
                        acc2[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a2[i], b[j], acc2[i][j], 0, 0, 0);
                }
        }
        __syncthreads();
    }

    if (A16) {
#pragma unroll
        for (int j = 0; j < 2; ++j) bsum[j] += __shfl_xor(bsum[j], 32);
    }

    const long long obase = (long long)(z / p.inner) * p.sC_outer + (long long)(z % p.inner) * p.sC_inner;

    // ---- phase 1: per-lane epilogue on the accumulator fragments
    double dm[2] = {0, 0}, dr[2] = {0, 0};
    int bias[2] = {0, 0};
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        int col = col0 + wn * 64 + j * 32 + (lane & 31);
        if (col < p.N) {
            if (p.bias) bias[j] = p.bias[col];
            if (EPI == EPI_RQ8_CH || EPI == EPI_RQ16_CH || EPI == EPI_RQ16_CH_RES || EPI == EPI_QKV) {
                dm[j] = p.dy_ch[col].m;
                dr[j] = p.dy_ch[col].r;
            }
        }
        if (EPI == EPI_RQ8_S) { dm[j] = p.dy_main.m; dr[j] = p.dy_main.r; }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int rl = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                int cl = wn * 64 + j * 32 + (lane & 31);
                int v = acc[i][j][r];
                if (A16) v = (int)((unsigned)v + ((unsigned)acc2[i][j][r] << 8) + ((unsigned)bsum[j] << 14));
                v += bias[j];
                if (EPI == EPI_RAW32) {
                    if (row0 + rl < p.M && col0 + cl < p.N)
                        reinterpret_cast<int32_t *>(p.out)[obase + (long long)(row0 + rl) * p.ldc + col0 + cl] = v;
                } else if (EPI == EPI_RQ16_CH || EPI == EPI_RQ16_CH_RES) {
                    int o = clamp_b<16>(rq_f64((double)v, dm[j], dr[j]));
                    *reinterpret_cast<int16_t *>(smem + rl * GEMM_SC16_LD + cl * 2) = (int16_t)o;
                } else {
                    int o = clamp_b<8>(rq_f64((double)v, dm[j], dr[j]));
                    *reinterpret_cast<int8_t *>(smem + rl * GEMM_SC8_LD + cl) = (int8_t)o;
                }
            }
    if (EPI == EPI_RAW32) return;
    __syncthreads();

    // ---- phase 2: coalesced write-out from the staged tile
    if (EPI == EPI_RQ8_CH || EPI == EPI_RQ8_S) {
        int8_t *out = reinterpret_cast<int8_t *>(p.out) + obase;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int id = tid + i * 256, row = id >> 3, c = id & 7;
            int grow = row0 + row, gcol = col0 + c * 16;
            if (grow < p.M && gcol < p.N) {
                v4i v = *reinterpret_cast<const v4i *>(smem + row * GEMM_SC8_LD + c * 16);
                int8_t *dst = out + (long long)grow * p.ldc + gcol;
                if (gcol + 16 <= p.N && ((p.ldc & 15) == 0)) {
                    *reinterpret_cast<v4i *>(dst) = v;
                } else {
                    const int8_t *s = reinterpret_cast<const int8_t *>(&v);
                    for (int e = 0; e < 16 && gcol + e < p.N; ++e) dst[e] = s[e];
                }
            }
        }
    } else if (EPI == EPI_RQ16_CH || EPI == EPI_RQ16_CH_RES) {
        int16_t *out = reinterpret_cast<int16_t *>(p.out) + obase;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int id = tid + i * 256, row = id >> 4, c = id & 15;
            int grow = row0 + row, gcol = col0 + c * 8;
            if (grow < p.M && gcol < p.N) {
                v4i v = *reinterpret_cast<const v4i *>(smem + row * GEMM_SC16_LD + c * 16);
                int16_t *dst = out + (long long)grow * p.ldc + gcol;
                const bool vec = (gcol + 8 <= p.N) && ((p.ldc & 7) == 0);
                if (EPI == EPI_RQ16_CH_RES) {
                    const int16_t *rp = p.residual + obase + (long long)grow * p.ldc + gcol;
                    int16_t rs[8];
                    if (vec) {
                        *reinterpret_cast<v4i *>(rs) = *reinterpret_cast<const v4i *>(rp);
                    } else {
                        for (int e = 0; e < 8; ++e) rs[e] = (gcol + e < p.N) ? rp[e] : 0;
                    }
                    int16_t *t = reinterpret_cast<int16_t *>(&v);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        double o = rq_f64((double)rs[e], p.dy_res.m, p.dy_res.r) +
                                   rq_f64((double)t[e], p.dy_main.m, p.dy_main.r);
                        t[e] = (int16_t)clamp_b<16>(o);
                    }
                }
                if (vec) {
                    *reinterpret_cast<v4i *>(dst) = v;
                } else {
                    const int16_t *s = reinterpret_cast<const int16_t *>(&v);
                    for (int e = 0; e < 8 && gcol + e < p.N; ++e) dst[e] = s[e];
                }
            }
        }
    } else if (EPI == EPI_QKV) {
        // rows-fastest mapping: a wave covers 64 consecutive tokens of one 16-column chunk
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int id = tid + i * 256, row = id & 127, c = id >> 7;
            int grow = row0 + row, gcol = col0 + c * 16;
            if (grow < p.M && gcol < p.N) {
                v4i v = *reinterpret_cast<const v4i *>(smem + row * GEMM_SC8_LD + c * 16);
                int which = gcol / p.D, within = gcol - which * p.D;
                int head = within / p.dh, d0 = within - head * p.dh;
                int b = grow / p.T, t = grow - b * p.T;
                long long bh = (long long)b * p.H + head;
                if (which < 2 || p.ldv == 0) {
                    int8_t *dst = (which == 0 ? p.q : (which == 1 ? p.k : p.vt)) + (bh * p.T + t) * p.dh + d0;
                    *reinterpret_cast<v4i *>(dst) = v;
                } else {
                    const int8_t *s = reinterpret_cast<const int8_t *>(&v);
                    int8_t *dst = p.vt + (bh * p.dh + d0) * p.ldv + t;
#pragma unroll
                    for (int e = 0; e < 16; ++e) dst[(long long)e * p.ldv] = s[e];
                }
            }
        }
    }
}
