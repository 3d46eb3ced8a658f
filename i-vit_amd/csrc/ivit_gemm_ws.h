// ivit_gemm_ws.h — the qkv QuantLinear of a D = 384 block (models/vit_quant.py:65-74 through quant_modules.py:21-80) with the
// tokens of a whole CU in LDS and the WEIGHTS of a 64-channel slab in registers — and, optionally, norm1 + qact1
// (quant_modules.py:353-386, quant_utils.py:213-253) computed straight into that LDS image, so that the 8-bit activations of the
// block's first LayerNorm never exist in HBM (vit_quant.py:136-140: norm1's output has one consumer, attn.qkv).
//
//   * a workgroup (8 waves, one per CU) owns a contiguous range of 32-token tiles (<= WS_MAXT per panel: 7 x 32 x 384 B = 86 KB),
//     laid out [64-column block][token][64 B] with the chunk permutation of ivit_mlp_rs.h (conflict-free ds_read_b128
//     B fragments).  LN = false: filled by DMA from the 8-bit activations.  LN = true: every wave normalises 8 rows at a time
//     (ivit_layernorm.h::LnGroup<384, 2>, the same arithmetic as layernorm_reg_kernel, byte for byte) and writes the bytes there;
//   * a wave's task is (64-channel slab = one head of q, k or v; one half of the panel's token tiles): the 2 x 12 16-byte A
//     fragments of the slab (24 KB, rows placed so that accumulator register v of lane (token, h) is channel 16 h + v) are
//     loaded into 96 registers, then the wave sweeps its token tiles two at a time — v_mfma_i32_32x32x32_i8, tokens as the B
//     operand, each B fragment feeding two MFMAs (with one channel tile per wave the LDS read port is the bound: 4 SIMDs x 1 KB
//     per 32-cycle MFMA is its whole 128 B/clk) — and requantises each sweep: fp64 FMA with the magic constant, saturating packs,
//     one 16-byte store per (token, 16 channels);
//   * no workgroup barrier after the prologue; bias and multipliers in LDS, so that the steady state has no vector-memory LOAD
//     behind a store (loads and stores retire through one in-order counter on this chip: a wait for a load is a wait for every
//     store in front of it).
//
// Measured stand-alone at DeiT-S b256 (tools/gemm_ws_probe.py): profiles/README.md, round 6.
#pragma once
#include "ivit_layernorm.h"
#include <type_traits>

#define WS_K 384
#define WS_KS 12                                 // k-steps of 32
#define WS_MAXT 7                                // 32-token tiles of a panel in LDS
#define WS_TOK (WS_MAXT * 32)
#define WS_KBLK (WS_TOK * 64)
#define WS_SOFF (6 * WS_KBLK)                     // output row offset of each token of the panel (int)
#define WS_SBIAS (WS_SOFF + WS_TOK * 4)           // the layer's bias (int32 x N) and multipliers (double x N)
#define WS_MAXN 1536
#define WS_SCQ (WS_SBIAS + WS_MAXN * 4)
#define WS_SLN (WS_SCQ + WS_MAXN * 8)             // LayerNorm's per-channel constants: c (double), bias_int, sc, 1 / sc (float) x 384
#define WS_SMEM (WS_SLN + WS_K * 20)
#define WS_THREADS 512
#define WS_MAGIC 6755399441055744.0
#ifndef WS_TRACE
#define WS_TRACE 0                               // probe builds: cycle stamps of workgroup WS_TRACE - 1 ([8 waves][64])
#endif

struct WsArgs {
    const int8_t *x;          // [M][384] 8-bit activations (LN = false)
    const v4i *wf;            // swizzled weights: fragment (ct * 12 + ks) * 64 + lane
    const int32_t *bias;      // [N]
    const double *cq;         // [N]
    int8_t *q, *k, *v;        // [B*H][T][64] each
    int M, N, T, H;
    void *dummy;              // >= 1 KB: where the lanes of rows >= M store
    // LN = true: the block's 16-bit input and norm1's constants (the arguments of ivit_layernorm_requant)
    const int16_t *x16;
    float ln_s;
    const float *ln_bias_int, *ln_sc;
    const ivit_dyadic *ln_dy;
    // EPI = WS_EPI_RES16 (attn.proj + qact2 with the identity branch, vit_quant.py:137-138 + quant_utils.py:238-244): out16 [M][N]
    // = clamp16(rq(residual, cr) + rq(clamp16(rq(acc + bias, cq)), cm)); cm, cr = m * 2^-e of the two dyadic multipliers, |.| < 2^9
    const int16_t *residual;
    int16_t *out16;
    double cm, cr;
    int8_t *ln_out8;          // EPI_RES16 with LN = true: norm2 + qact3 of out16's rows (ln_s .. ln_dy are norm2's), [M][384]
    long long *trace;
};
#define WS_EPI_QKV8 0
#define WS_EPI_RES16 1
#define WS_EPI_RQ8 2                             // plain QuantLinear -> QuantAct(8): q = out8 [M][N] row-major (Swin's qkv layer)

__device__ __forceinline__ int ws_chan_of_row(int rho) { return ((rho >> 2) & 1) * 16 + (rho >> 3) * 4 + (rho & 3); }
__device__ __forceinline__ int ws_g(int tok) { return ((tok >> 1) & 3) ^ ((tok >> 3) & 3) ^ ((tok >> 4) & 1); }

// weights [N][384] -> fragments of 64 lanes x 16 B: fragment ct * 12 + ks, lane l = (row l & 31, k half l >> 5)
__global__ __launch_bounds__(256) void ws_swizzle_kernel(const int8_t *__restrict__ w, v4i *__restrict__ wf, int N) {
    const int nfrag = N / 32 * WS_KS;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nfrag * 64; i += gridDim.x * 256) {
        const int l = i & 63, f = i >> 6, ct = f / WS_KS, ks = f - ct * WS_KS;
        const int ch = 32 * ct + ws_chan_of_row(l & 31);
        wf[i] = *reinterpret_cast<const v4i *>(w + (size_t)ch * WS_K + 32 * ks + 16 * (l >> 5));
    }
}

template <bool FMA, bool LN, int EPI = WS_EPI_QKV8>
__global__ __launch_bounds__(WS_THREADS, 2) void gemm_ws_qkv_kernel(WsArgs p) {
    extern __shared__ __attribute__((aligned(256))) char sm[];
    typedef double v2d __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(3))) char lds_c;
    typedef __attribute__((address_space(3))) v4i lds_v4i;
    typedef __attribute__((address_space(3))) int lds_i32;
    typedef __attribute__((address_space(3))) unsigned lds_u32;
    typedef __attribute__((address_space(3))) v2d lds_v2d;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned sm_lds = (unsigned)(size_t)(lds_c *)sm;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, tok = lane & 31, kh = lane >> 5, e = kh ^ ws_g(tok);
    int n_stamp = 0;
    auto stamp = [&]() __attribute__((always_inline)) {
        if (WS_TRACE) {
            if (blockIdx.x == WS_TRACE - 1 && (threadIdx.x & 63) == 0 && n_stamp < 64) p.trace[wave * 64 + n_stamp] = __builtin_readcyclecounter();
            ++n_stamp;
        }
    };
    stamp();

    const int ntt = (p.M + 31) >> 5;
    const int t_beg = (int)((long long)ntt * blockIdx.x / gridDim.x), t_end = (int)((long long)ntt * (blockIdx.x + 1) / gridDim.x);
    const int ncp = p.N >> 6, ncp3 = ncp / 3;                 // 64-channel slabs; per q | k | v
    for (int i = tid; i < p.N; i += WS_THREADS) {
        reinterpret_cast<int *>(sm + WS_SBIAS)[i] = p.bias[i];
        reinterpret_cast<double *>(sm + WS_SCQ)[i] = p.cq[i];
    }
    const unsigned lane16 = lane * 16;
    double *cC = reinterpret_cast<double *>(sm + WS_SLN);
    float *cB = reinterpret_cast<float *>(sm + WS_SLN + WS_K * 8), *cSc = cB + WS_K, *cY = cSc + WS_K;
    bool ln_fast = false;
    constexpr bool LN_HEAD = LN && EPI != WS_EPI_RES16, LN_TAIL = LN && EPI == WS_EPI_RES16;
    if constexpr (LN) ln_fast = ln_stage_constants<WS_K, WS_THREADS>(p.ln_bias_int, p.ln_sc, p.ln_dy, cC, cB, cSc, cY);

    for (int t0 = t_beg; t0 < t_end; t0 += WS_MAXT) {
        const int n_own = min(WS_MAXT, t_end - t0);
        if (t0 != t_beg) __syncthreads();            // a later panel: every wave is done with the previous one
        if constexpr (LN_HEAD) {
            // ---- norm1 + qact1 of the panel's rows into the LDS image: 8 rows per wave and pass, 8 lanes per row, lane (k, h)
            // owns channels 32 i + 8 k + 4 h .. + 3 of every step i — 4 bytes of chunk (i & 1) * 2 + (k >> 1) of K block i >> 1
            typedef LnGroup<WS_K, 2> G;
            const int j = lane & 7, k = j >> 1, hh = j & 1;
            const float ys = rcp_rn(p.ln_s);
            // (the phase is VALU-issue-bound like layernorm_reg_kernel itself: requesting pass n + 1's rows ahead, or two rows per lane
            // group for more independent chains, measured 53.5 / 54.4 us against 51.3 for this form)
            for (int r0 = wave * 8; r0 < n_own * 32; r0 += 64) {
                const int tokl = r0 + (lane >> 3);
                const long long row_raw = (long long)t0 * 32 + tokl;
                const bool live = row_raw < p.M;
                const int16_t *xp = p.x16 + (live ? row_raw : (long long)p.M - 1) * WS_K + 8 * k + 4 * hh;
                float xv[G::NSTEP][G::EPC];
#pragma unroll
                for (int i = 0; i < G::NSTEP; ++i) {
                    const LnRaw<4>::T t = *reinterpret_cast<const LnRaw<4>::T *>(xp + 32 * i);
#pragma unroll
                    for (int c = 0; c < 4; ++c) xv[i][c] = requotient_m((float)t[c], p.ln_s, ys);
                }
                const unsigned rowa = sm_lds + tokl * 64 + (k & 1) * 8 + 4 * hh, gk = (unsigned)((k >> 1) ^ ws_g(tokl));
                G::run(xv, j, k, 8 * k + 4 * hh, ln_fast, live, cC, cB, cSc, cY, p.ln_bias_int, p.ln_sc, p.ln_dy,
                       [&](int i, unsigned pk0, unsigned) __attribute__((always_inline)) {
                           *(lds_u32 *)(size_t)(rowa + (i >> 1) * WS_KBLK + ((gk ^ ((i & 1) * 2)) << 4)) = pk0;
                       });
            }
        } else {
            // ---- the panel's tokens: global -> LDS by DMA, 16 tokens x 4 chunk slots per instruction (source chunk = slot ^ g)
            for (int tg = wave; tg < n_own * 2; tg += 8) {
                const int tokl = tg * 16 + (lane >> 2), c = (lane & 3) ^ ws_g(tokl);
                const long long grow = min((long long)t0 * 32 + tokl, (long long)p.M - 1);
                const int8_t *src = p.x + grow * WS_K + c * 16;
#pragma unroll
                for (int kb = 0; kb < 6; ++kb) {
                    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(kb * WS_KBLK + tg * 1024));
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + kb * 64),
                                                     (__attribute__((address_space(3))) void *)(sm + dst), 16, 0, 0);
                }
            }
        }
        // output row offset of every token of the panel: (b * H * T + t_in_image) * 64
        if (tid < WS_TOK) {
            const int row = min(t0 * 32 + tid, p.M - 1), b = row / p.T;
            reinterpret_cast<int *>(sm + WS_SOFF)[tid] = (b * p.H * p.T + (row - b * p.T)) * 64;
        }
        // tasks: (slab, token half); half a = tiles [0, na), half b = [na, n_own)
        const int na = (n_own + 1) >> 1, ntask = n_own > 1 ? 2 * ncp : ncp;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        stamp();

        const unsigned fa0 = sm_lds + tok * 64 + e * 16, fa1 = sm_lds + tok * 64 + (e ^ 2) * 16;
        for (int task = wave; task < ntask; task += 8) {
            const int half = task >= ncp, cp = task - half * ncp;
            const int tb0 = half ? na : 0, te = half ? n_own : na;
            // the slab's weights: NOT carried from task to task (loop-carried and redefined behind their last use, the compiler
            // copies all 96 registers at the back edge and spills); the partner wave of the SIMD works through the latency
            v4i W[2][WS_KS];
            {
                const char *wq = reinterpret_cast<const char *>(p.wf + (size_t)cp * 2 * WS_KS * 64);
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int ks = 0; ks < WS_KS; ++ks) W[c][ks] = *reinterpret_cast<const v4i *>(wq + lane16 + (c * WS_KS + ks) * 1024);
            }
            const int chb = 64 * cp + 16 * kh;
            const int which = EPI == WS_EPI_QKV8 ? cp / max(ncp3, 1) : 0;
            int8_t *obase = (which == 0 ? p.q : which == 1 ? p.k : p.v) + (size_t)(cp - which * ncp3) * p.T * 64 + 16 * kh;
            auto sweep = [&](auto nt_c, const int tb) __attribute__((always_inline)) {
                constexpr int NT = decltype(nt_c)::value;
                v4i bf[2][NT];
                v16i acc[2][NT];
                const unsigned fb0 = fa0 + tb * 2048, fb1 = fa1 + tb * 2048;
                stamp();
                int toff[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) toff[t] = *(lds_i32 *)(size_t)(sm_lds + WS_SOFF + ((tb + t) * 32 + tok) * 4);
                auto load_b = [&](int ks, int slot) __attribute__((always_inline)) {
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        bf[slot][t] = *(lds_v4i *)(size_t)(((ks & 1) ? fb1 : fb0) + (ks >> 1) * WS_KBLK + t * 2048);
                };
                load_b(0, 0);
                // EPI_RES16: the identity rows of this sweep, requested in front of its K loop (a vector load issued behind the
                // previous sweep's stores would wait for them)
                v4i idr[EPI == WS_EPI_RES16 ? 2 : 1][EPI == WS_EPI_RES16 ? NT : 1][2];
                if constexpr (EPI == WS_EPI_RES16) {
#pragma unroll
                    for (int c = 0; c < 2; ++c)
#pragma unroll
                        for (int t = 0; t < NT; ++t) {
                            const long long row = min((long long)(t0 + tb + t) * 32 + tok, (long long)p.M - 1);
                            const int16_t *rp = p.residual + row * p.N + chb + 32 * c;
                            idr[c][t][0] = *reinterpret_cast<const v4i *>(rp);
                            idr[c][t][1] = *reinterpret_cast<const v4i *>(rp + 8);
                        }
                }
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const v4i b4 = *(lds_v4i *)(size_t)(sm_lds + WS_SBIAS + (chb + 32 * c + 4 * q4) * 4);
#pragma unroll
                        for (int t = 0; t < NT; ++t) { acc[c][t][4 * q4] = b4[0]; acc[c][t][4 * q4 + 1] = b4[1]; acc[c][t][4 * q4 + 2] = b4[2]; acc[c][t][4 * q4 + 3] = b4[3]; }
                    }
#pragma unroll
                for (int ks = 0; ks < WS_KS; ++ks) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (ks + 1 < WS_KS) load_b(ks + 1, (ks + 1) & 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int c = 0; c < 2; ++c)
#pragma unroll
                        for (int t = 0; t < NT; ++t)
                            acc[c][t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(W[c][ks], bf[ks & 1][t], acc[c][t], 0, 0, 0);
                }
                stamp();
                // requant to 8 bits: fma(z, c, magic + 128) leaves Q + 128 in the low dword; the two packs saturate to [0, 255]
                // = clamp(Q, -128, 127) + 128; the xor takes the bias off again.  One (channel tile, token tile) at a time:
                // sixteen channels of a token per lane, one 16-byte store
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    v2d cqv[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) cqv[j] = *(lds_v2d *)(size_t)(sm_lds + WS_SCQ + (chb + 32 * c + 2 * j) * 8);
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        if constexpr (EPI == WS_EPI_RES16) {
                            // 16-bit requant, then the residual QuantAct: both terms are integers < 2^31 / 2, their sum is the
                            // reference's fp64 sum; v_cvt_pk_i16_i32 clamps to 16 bits while packing
                            v4i o0, o1;
#pragma unroll
                            for (int d = 0; d < 8; ++d) {
                                const unsigned rw = (unsigned)(d < 4 ? idr[c][t][0][d] : idr[c][t][1][d - 4]);
                                int o[2];
#pragma unroll
                                for (int h2 = 0; h2 < 2; ++h2) {
                                    const int v = 2 * d + h2;
                                    const double m = cqv[v >> 1][v & 1];
                                    const double tq = FMA ? __builtin_fma((double)acc[c][t][v], m, WS_MAGIC) : ((double)acc[c][t][v] * m + WS_MAGIC);
                                    const int t16 = min(max(__double2loint(tq), -32768), 32767);
                                    const int r = h2 ? ((int)rw >> 16) : (int)(short)(rw & 0xffffu);
                                    o[h2] = rq_fast(r, p.cr) + rq_fast(t16, p.cm);
                                }
                                int pk;
                                asm("v_cvt_pk_i16_i32 %0, %1, %2" : "=v"(pk) : "v"(o[0]), "v"(o[1]));
                                if (d < 4) o0[d] = pk; else o1[d - 4] = pk;
                            }
                            asm volatile("" : "+v"(o0), "+v"(o1));
                            const long long row = (long long)(t0 + tb + t) * 32 + tok;
                            if (row < p.M) {
                                int16_t *op = p.out16 + row * p.N + chb + 32 * c;
                                *reinterpret_cast<v4i *>(op) = o0;
                                *reinterpret_cast<v4i *>(op + 8) = o1;
                            }
                            continue;
                        }
                        v4i o4;
#pragma unroll
                        for (int q4 = 0; q4 < 4; ++q4) {
                            int o[4];
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const double m = cqv[2 * q4 + (i >> 1)][i & 1];
                                const double tq = FMA ? __builtin_fma((double)acc[c][t][4 * q4 + i], m, WS_MAGIC + 128.0)
                                                      : ((double)acc[c][t][4 * q4 + i] * m + (WS_MAGIC + 128.0));
                                o[i] = __double2loint(tq);
                            }
                            unsigned p01, p23, b01, b23;
                            asm("v_cvt_pk_i16_i32 %0, %1, %2" : "=v"(p01) : "v"(o[0]), "v"(o[1]));
                            asm("v_cvt_pk_i16_i32 %0, %1, %2" : "=v"(p23) : "v"(o[2]), "v"(o[3]));
                            asm("v_sat_pk_u8_i16 %0, %1" : "=v"(b01) : "v"(p01));
                            asm("v_sat_pk_u8_i16 %0, %1" : "=v"(b23) : "v"(p23));
                            int hq = (int)(__builtin_amdgcn_perm(b23, b01, 0x05040100u) ^ 0x80808080u);
                            asm volatile("" : "+v"(hq));       // pinned: left alone, the optimiser converts every accumulator first
                            o4[q4] = hq;
                        }
                        const int row = (t0 + tb + t) * 32 + tok;
                        if constexpr (EPI == WS_EPI_RQ8) {
                            if (row < p.M) *reinterpret_cast<v4i *>(p.q + (long long)row * p.N + chb + 32 * c) = o4;
                        } else {
                            *reinterpret_cast<v4i *>(row < p.M ? obase + toff[t] + 32 * c : (int8_t *)p.dummy + lane16) = o4;
                        }
                    }
                }
            };
            // a half has one to four tiles
            int tb = tb0;
            if (te - tb > 2) { sweep(std::integral_constant<int, 2>{}, tb); tb += 2; }
            if (te - tb == 2) sweep(std::integral_constant<int, 2>{}, tb);
            else sweep(std::integral_constant<int, 1>{}, tb);
        }
        if constexpr (LN_TAIL) {
            // ---- norm2 + qact3 of the panel's rows (vit_quant.py:139-140): every channel of a row was produced by this workgroup;
            // its stores are complete (vmcnt) and no line of out16 was ever read through this CU's L1, so the rows come back from
            // the L2 they were just written to.  Same arithmetic as layernorm_reg_kernel<384, 2>, bytes to ln_out8 [M][384]
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            typedef LnGroup<WS_K, 2> G;
            const int j = lane & 7, k = j >> 1, hh = j & 1;
            const float ys = rcp_rn(p.ln_s);
            for (int r0 = wave * 8; r0 < n_own * 32; r0 += 64) {
                const long long row_raw = (long long)t0 * 32 + r0 + (lane >> 3);
                const bool live = row_raw < p.M;
                const long long row = live ? row_raw : (long long)p.M - 1;
                const int16_t *xp = p.out16 + row * WS_K + 8 * k + 4 * hh;
                float xv[G::NSTEP][G::EPC];
#pragma unroll
                for (int i = 0; i < G::NSTEP; ++i) {
                    const LnRaw<4>::T t = *reinterpret_cast<const LnRaw<4>::T *>(xp + 32 * i);
#pragma unroll
                    for (int c = 0; c < 4; ++c) xv[i][c] = requotient_m((float)t[c], p.ln_s, ys);
                }
                G::run(xv, j, k, 8 * k + 4 * hh, ln_fast, live, cC, cB, cSc, cY, p.ln_bias_int, p.ln_sc, p.ln_dy, p.ln_out8 + row * WS_K + 8 * k + 4 * hh);
            }
        }
        stamp();
    }
}
