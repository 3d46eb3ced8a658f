// ivit_gemm_ws.h — weight-stationary streaming GEMM for the short-K QuantLinear layers
// (K = 32*KC, KC in {6, 12}: qkv / fc1 of DeiT-T and DeiT-S).
//
// The LDS-tiled kernel in ivit_gemm2.h moves ~1.4 KB through the LDS per MFMA (two operand
// fragments read + the DMA that wrote them), which at K = 384 is as much LDS time as MFMA time.
// Here the WEIGHTS never touch the LDS: a wave keeps its whole 64-channel x K weight panel in
// registers (2 x KC fragments = 96 VGPRs at K = 384) for the life of the block, and only 32-token
// activation tiles stream through a 4-deep global_load_lds ring.  Per 32 tokens a wave issues
// 2*KC MFMAs off KC ds_read_b128 — 0.5 KB of LDS per MFMA and one s_barrier per 2*KC MFMAs.
// The tile is stored chunk-major ([k/32][token][32 B]): a DMA instruction writes, and a fragment
// read fetches, one contiguous 1 KB block — conflict-free with no swizzle.
//
// A block is WS_WAVES (4) waves = 256 channels wide (one wave per SIMD, so three blocks share a CU evenly) and walks a contiguous run of token groups; the grid is
// (N / 256 panels) x (chunks), sized by the host to one resident round (3 blocks per CU).  Blocks
// that share a token run sit on the same XCD (same L2).
//
// Epilogue (per 32 tokens, straight from the accumulators, under the other waves' MFMAs): the
// accumulators start at the bias; requant is rne(fl64(z*c)) evaluated as
// loint(fl64(z*c) + 1.5*2^52) — the same two roundings as the reference's
// round(z.double()*m.double() / 2^e) (quant_utils.py:229-231), valid while |z*c| < 2^31, which is
// checked per channel from K and the bias when the constants are staged (else the rint form).
#pragma once
#include "ivit_gemm2.h"

#define WS_NPANEL 256
#define WS_NS 4
// NSUB = 32-channel sub-tiles per wave: 2 -> 4 waves/block, 2 waves/SIMD (256 VGPRs);
//                                       1 -> 8 waves/block, 4 waves/SIMD (128 VGPRs)
template <int NSUB> struct WsCfg {
    static constexpr int WAVES = WS_NPANEL / (32 * NSUB);
    static constexpr int OCC = NSUB == 2 ? 2 : 4;           // waves per SIMD
    static constexpr int BLOCKS_PER_CU = 2;
};

template <int EPI, int KC, int NSUB>
__global__ __launch_bounds__(WsCfg<NSUB>::WAVES * 64, WsCfg<NSUB>::OCC) void gemm_ws_kernel(GemmArgs p, int nchunks) {
    constexpr int STAGE = KC * 1024;
    constexpr int WAVES = WsCfg<NSUB>::WAVES;
    __shared__ __attribute__((aligned(16))) char smem[WS_NS * STAGE + WS_NPANEL * 12 + 16];
    double *sC = reinterpret_cast<double *>(smem + WS_NS * STAGE);
    int *sBias = reinterpret_cast<int *>(smem + WS_NS * STAGE + WS_NPANEL * 8);
    int *sFlag = reinterpret_cast<int *>(smem + WS_NS * STAGE + WS_NPANEL * 12);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int8_t *A = reinterpret_cast<const int8_t *>(p.A);

    const int panels = (p.N + WS_NPANEL - 1) / WS_NPANEL;
    const int bid = blockIdx.x, xcd = bid & 7, idx = bid >> 3;
    const int panel = idx % panels, chunk = xcd + 8 * (idx / panels);
    const long long G = ((long long)p.M + 31) / 32;
    const long long g0 = chunk * G / nchunks, g1 = (chunk + 1) * G / nchunks;
    const int S = (p.dbg & 32) ? 0 : ((p.dbg & 64) ? 2 : (int)(g1 - g0));
    const int n0 = panel * WS_NPANEL + wave * (32 * NSUB);
    const int ppw = (KC - wave + WAVES - 1) / WAVES;   // DMA pieces this wave issues per step (wave-uniform)

    if (tid == 0) *sFlag = 0;
    __syncthreads();
    if (tid < WS_NPANEL) {
        const int n = panel * WS_NPANEL + tid;
        const bool in = n < p.N;
        const double cv = in ? p.dy_ch[n].m * p.dy_ch[n].r : 0.0;
        const int bs = (in && p.bias) ? p.bias[n] : 0;
        sC[tid] = cv;
        sBias[tid] = bs;
        // |acc + bias| <= K*2^14 + |bias|; the magic-number rounding needs |z*c| < 2^31
        const double zmax = (double)p.K * 16384.0 + fabs((double)bs);
        if (!(fabs(cv) * zmax < 2147483000.0)) atomicOr(sFlag, 1);
    }

    auto issue = [&](int s) {
        const long long t = min((g0 + s) * 32 + l31, (long long)p.M - 1);
        const int8_t *src = A + t * p.lda + half * 16;
        char *st = smem + (s % WS_NS) * STAGE;
#pragma unroll
        for (int kc0 = 0; kc0 < KC; kc0 += WAVES) {
            const int kc = kc0 + wave;
            if (kc >= KC) break;
            unsigned loff = __builtin_amdgcn_readfirstlane((unsigned)(kc * 1024));
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + kc * 32),
                                             (__attribute__((address_space(3))) void *)(st + loff), 16, 0, 0);
        }
    };
    for (int s = 0; s < WS_NS - 1 && s < S; ++s) issue(s);   // first tiles fly while the weights load
    // the wave's weight panel: fragment (j, kc) = channels n0 + 32j + (lane & 31), k = 32kc + 16*half ..+16
    v4i w[NSUB][KC];
#pragma unroll
    for (int j = 0; j < NSUB; ++j) {
        const int row = n0 + 32 * j + l31;
        const int8_t *wp = p.B + (long long)min(row, p.N - 1) * p.ldb + half * 16;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            v4i v = *reinterpret_cast<const v4i *>(wp + kc * 32);
            w[j][kc] = row < p.N ? v : v4i{0, 0, 0, 0};
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // weights and the first WS_NS-1 tiles have landed

    __syncthreads();
    const bool fastrq = (*sFlag == 0);

    constexpr int OLO = -128, OHI = 127;
    // output addressing that does not change from step to step: per sub-tile j the 16-channel run this
    // lane stores (q/k) or its four 4-channel runs (v^T); the token's (image, position) advance by 32
    long long obase[NSUB] = {};
    int vsel[NSUB] = {};                  // QKV: 0 = q, 1 = k, 2 = v^T
    if (EPI == EPI_QKV) {
#pragma unroll
        for (int j = 0; j < NSUB; ++j) {
            const int ncol0 = n0 + j * 32;
            if (ncol0 >= 2 * p.D) {
                vsel[j] = 2;
                const int within = ncol0 + half * 4 - 2 * p.D;     // + 8g below: same head while dh % 32 == 0
                const int head = within / p.dh, d0 = within - head * p.dh;
                obase[j] = ((long long)head * p.dh + d0) * p.ldv;
            } else {
                const int gcol = ncol0 + half * 16;
                const int which = gcol / p.D, within = gcol - which * p.D;
                const int head = within / p.dh, d0 = within - head * p.dh;
                vsel[j] = which;
                obase[j] = (long long)head * p.T * p.dh + d0;
            }
        }
    }
    int tb = 0, tt = 0;                   // image index and position of token g0*32 + l31
    if (EPI == EPI_QKV) {
        const long long t0 = g0 * 32 + l31;
        tb = (int)(t0 / p.T);
        tt = (int)(t0 - (long long)tb * p.T);
    }
    for (int s = 0; s < S; ++s) {
        // loads of step s were issued WS_NS-1 steps ago; PPW younger loads per later step.  Stores of the
        // previous epilogue are younger still: ignoring them only makes the wait longer, never shorter.
        const int later = min(S - 1 - s, WS_NS - 2);
        switch (later * ppw) {
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;   // 3 x 2
        }
        if (!(p.dbg & 8)) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (s + WS_NS - 1 < S && !(p.dbg & 4)) issue(s + WS_NS - 1);

        v16i acc[NSUB];
#pragma unroll
        for (int j = 0; j < NSUB; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const v4i b4 = *reinterpret_cast<const v4i *>(sBias + wave * (32 * NSUB) + j * 32 + g * 8 + half * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[j][g * 4 + e] = b4[e];
            }
        const char *st = smem + (s % WS_NS) * STAGE + lane * 16;
        // fragments in batches of AB (all 12 at 2 waves/SIMD; 6 at 4 waves/SIMD, where registers are short
        // and the other waves cover the second batch's latency): the MFMAs of a batch run back to back
        constexpr int AB = NSUB == 2 ? KC : KC / 2;
#pragma unroll
        for (int k0 = 0; k0 < KC; k0 += AB) {
            v4i a[AB];
#pragma unroll
            for (int kc = 0; kc < AB; ++kc) a[kc] = *reinterpret_cast<const v4i *>(st + (k0 + kc) * 1024);
            if (p.dbg & 16) {
#pragma unroll
                for (int kc = 0; kc < AB; ++kc) acc[0][kc] ^= a[kc][0] ^ a[kc][1] ^ a[kc][2] ^ a[kc][3];
            } else {
#pragma unroll
                for (int kc = 0; kc < AB; ++kc) {
#pragma unroll
                    for (int j = 0; j < NSUB; ++j)
                        acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(w[j][k0 + kc], a[kc], acc[j], 0, 0, 0);
                }
            }
        }
        if (p.dbg & 1) {   // ablation: main loop only
            int sx = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) sx ^= acc[0][r] ^ acc[NSUB - 1][r];
            if (sx == 0x12345678) reinterpret_cast<int *>(p.out)[tid] = sx;
            continue;
        }

        // ---- epilogue for tokens (g0+s)*32 + (lane & 31)
        const long long grow = (g0 + s) * 32 + l31;
        auto epilogue = [&](auto fast) {
#pragma unroll
        for (int j = 0; j < NSUB; ++j) {
            unsigned W[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cl = wave * (32 * NSUB) + j * 32 + g * 8 + half * 4;
                int o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const double t = (double)acc[j][g * 4 + e] * sC[cl + e];
                    const int v = decltype(fast)::value ? __double2loint(t + 6755399441055744.0) : (int)__builtin_rint(t);
                    o[e] = min(max(v, OLO), OHI);
                }
                unsigned w01 = __builtin_amdgcn_perm((unsigned)o[1], (unsigned)o[0], 0x0c0c0400u);
                unsigned w23 = __builtin_amdgcn_perm((unsigned)o[3], (unsigned)o[2], 0x0c0c0400u);
                W[g] = __builtin_amdgcn_perm(w23, w01, 0x05040100u);
            }
            const int ncol0 = n0 + j * 32;                 // this sub-tile's first channel
            if (EPI == EPI_QKV && vsel[j] == 2) {
                // v^T rows are token-contiguous: 32 lanes = 32 consecutive tokens -> byte stores
                if (grow < p.M && ncol0 < p.N) {
                    int8_t *dst0 = p.vt + (long long)tb * p.H * p.dh * p.ldv + obase[j] + tt;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        int8_t *dst = dst0 + (long long)(g * 8) * p.ldv;
#pragma unroll
                        for (int e = 0; e < 4; ++e) dst[(long long)e * p.ldv] = (int8_t)(W[g] >> (8 * e));
                    }
                }
                continue;
            }
            // half-wave exchange: lanes < 32 end with channels 0..15 of the sub-tile, lanes >= 32 with 16..31
            auto s02 = __builtin_amdgcn_permlane32_swap(W[0], W[2], false, false);
            auto s13 = __builtin_amdgcn_permlane32_swap(W[1], W[3], false, false);
            const v4i v = {(int)s02[0], (int)s02[1], (int)s13[0], (int)s13[1]};
            const int gcol = ncol0 + half * 16;
            if (grow < p.M && gcol < p.N) {
                if (EPI == EPI_QKV) {
                    int8_t *dst = (vsel[j] == 0 ? p.q : p.k) + ((long long)tb * p.H * p.T + tt) * p.dh + obase[j];
                    *reinterpret_cast<v4i *>(dst) = v;
                } else {
                    *reinterpret_cast<v4i *>(reinterpret_cast<int8_t *>(p.out) + grow * p.ldc + gcol) = v;
                }
            }
        }
        };
        if (fastrq) epilogue(std::true_type{});
        else epilogue(std::false_type{});
        if (EPI == EPI_QKV) {             // next step: 32 tokens on (T >= 32: at most one image boundary)
            tt += 32;
            if (tt >= p.T) { tt -= p.T; ++tb; }
        }
    }
}
