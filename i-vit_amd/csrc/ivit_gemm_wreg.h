// ivit_gemm_wreg.h — QuantLinear + QuantAct for SHORT K (K <= 192: Swin stages 0 / 1, models/swin_quant.py:121-169,251-301
// through quant_modules.py:67-97 and quant_utils.py:213-253), weights stationary in REGISTERS.
//
// Why another GEMM.  At K = 96 a 256 x 128 output tile is 1.5 k-steps of work behind a cold operand fetch: the
// launch-per-tile kernel (ivit_gemm2.h) runs Swin-T's stage-0 qkv (802 816 x 288 x 96) at 145 us around a 56 us main loop,
// with the load / requant / store phases of a tile adding up instead of overlapping (profiles/README.md, round 4).  These
// shapes are HBM-bound (308 MB per launch, ~62 us) with almost no arithmetic per byte, so the kernel is built around the
// data stream instead of the tile:
//   * a wave keeps the weight fragments of its 32 NCT output channels for ALL of K in registers (NCT x K / 32 fragments of
//     4 registers: 36 at K = 96, 72 at K = 192) — weights are read from L2 once per workgroup, never again;
//   * the four waves of a workgroup take the four 32-token tiles of a 128-token row tile; the activation tile travels
//     global -> LDS by DMA one row tile ahead (two buffers, one workgroup barrier per row tile);
//   * v_mfma_i32_32x32x32_i8 with the weights as the A operand and the MFMA rows placed so that accumulator register v of
//     lane (token, h) is channel 16 h + v of the tile (the trick of ivit_mlp_rs.h): 16 consecutive output channels per lane,
//     stored straight from registers as one (8 bit) or two (16 bit) 16-byte global stores per tile — no staging pass;
//   * workgroups are persistent and small (2 - 4 per CU, bounded by registers): the other workgroups' loads, MFMAs and
//     stores fill a workgroup's waits, which is what the big-tile kernels try to do inside one workgroup.  (Measured and
//     not kept: parking a tile's packed outputs in registers and storing them behind the next barrier, so that the
//     vmcnt(0) guarding a DMA buffer never meets a young store — stage-0 qkv 87 -> 105 us;)
//   * the N / (32 NCT) channel groups of a row tile run as different workgroups of ONE XCD at about the same time (block
//     ids 8 apart; a workgroup keeps its channel group for life), so the activation rows reach HBM once.
// Epilogues = gemm_glds_kernel's: clamp(rne(fl64((acc + bias) c))) to 8 or 16 bits per channel (magic-number rounding
// where |z c| < 2^31 is provable for every channel of the group, v_rndne_f64 otherwise), optionally followed by the
// block's residual QuantAct (16 bit, quant_utils.py:232-245).
#pragma once
#include "ivit_gemm.h"
#include "ivit_gemm2.h"

#define GW_BM 128
#define GW_THREADS 256
// waves per SIMD the register budget is set for (= workgroups per CU: one wave of a workgroup per SIMD).  Weights NCT KS 4 +
// accumulators 16 + multipliers 32 (+ 8 identity): 4 at K = 96 without the residual, 3 with it or at NCT KS <= 12, else 2
#define GW_MINW(EPI, KS, NCT) ((NCT) * (KS) <= 9 ? ((EPI) == EPI_RQ16_CH_RES ? 3 : 4) : ((NCT) * (KS) <= 12 && (EPI) != EPI_RQ16_CH_RES ? 3 : 2))

__device__ __forceinline__ int gw_chan_of_row(int rho) { return ((rho >> 2) & 1) * 16 + (rho >> 3) * 4 + (rho & 3); }

// KS = K / 32 k-steps, NCT = 32-channel tiles per workgroup (its channel group)
template <int EPI, int KS, int NCT>
__global__ __launch_bounds__(GW_THREADS, GW_MINW(EPI, KS, NCT)) void gemm_wreg_kernel(GemmArgs p) {
    static_assert(EPI == EPI_RQ8_CH || EPI == EPI_RQ16_CH || EPI == EPI_RQ16_CH_RES, "per-channel requant epilogues");
    constexpr int K = KS * 32, NC = NCT * 32, ABUF = GW_BM * K;
    constexpr bool OUT8 = EPI == EPI_RQ8_CH;
    typedef double v2d __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(3))) char lds_c;
    typedef __attribute__((address_space(3))) v4i lds_v4i;
    __shared__ __attribute__((aligned(256))) char sA[2 * ABUF];
    __shared__ __attribute__((aligned(16))) double sC[NC];
    __shared__ __attribute__((aligned(16))) int sBias[NC];
    __shared__ int sUnsafe;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tok = lane & 31, kh = lane >> 5;

    // ---- work: XCD x = blockIdx % 8 owns the row tiles rt = 8 i + x; its L workgroups walk (row tile, channel group) pairs
    // channel-group-fastest, L is a multiple of the group count: this workgroup's channel group never changes
    const int ncg = p.N / NC, xcd = blockIdx.x & 7, L = gridDim.x >> 3, l = blockIdx.x >> 3;
    const int cg = l % ncg, chbase = cg * NC;
    const long long nrt = ((long long)p.M + GW_BM - 1) / GW_BM;
    const long long nmine = (nrt - xcd + 7) / 8;               // row tiles of this XCD
    const int8_t *A = reinterpret_cast<const int8_t *>(p.A);

    // ---- this wave's weight fragments, straight from W [N][K] (once)
    v4i wf[NCT][KS];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            wf[ct][ks] = *reinterpret_cast<const v4i *>(p.B + (size_t)(chbase + 32 * ct + gw_chan_of_row(tok)) * K + 32 * ks + 16 * kh);
    if (tid == 0) sUnsafe = 0;
    __syncthreads();
    if (tid < NC) {
        const int ch = chbase + tid;
        const double cv = p.dy_ch[ch].m * p.dy_ch[ch].r;
        const int bs = p.bias ? p.bias[ch] : 0;
        sC[tid] = cv;
        sBias[tid] = bs;
        // magic-number rounding needs |(acc + bias) * c| < 2^31; |acc| <= K * 2^14
        if (!(fabs(cv) * ((double)K * 16384.0 + fabs((double)bs)) < 2147483000.0)) sUnsafe = 1;
    }

    // activation row tile -> LDS by DMA: [k-step][token][2 x 16 B], the chunk of (token, h) at position h ^ (token >> 4 & 1)
    // (conflict-free B-fragment ds_read_b128: lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and their twins).  The
    // permutation is applied on the source side; one instruction moves 32 tokens x 32 B
    auto a_dma = [&](long long rt, int buf) __attribute__((always_inline)) {
        const int tk = lane >> 1, slot = lane & 1;
#pragma unroll
        for (int i = 0; i < (KS * 4 + 3) / 4; ++i) {
            const int id = wave + 4 * i;                       // (k-step, 32-token group): KS * 4 instructions per tile
            if (id < KS * 4) {
                const int ks = id >> 2, tg = id & 3, tokl = tg * 32 + tk, hsrc = slot ^ ((tokl >> 4) & 1);
                const long long row = min(rt * GW_BM + tokl, (long long)p.M - 1);
                const int8_t *src = A + row * K + 32 * ks + 16 * hsrc;
                const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(buf * ABUF + (ks * GW_BM + tg * 32) * 32));
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)(sA + dst), 16, 0, 0);
            }
        }
    };
    const unsigned sA_lds = (unsigned)(size_t)(lds_c *)sA;
    const unsigned fa = sA_lds + (wave * 32 + tok) * 32 + ((kh ^ ((tok >> 4) & 1)) * 16);      // + buf * ABUF + ks * GW_BM * 32
    const double cm = p.dy_main.m * p.dy_main.r, cr = p.dy_res.m * p.dy_res.r;
    const bool res_fast = fabs(cm) < RQ_FAST_CLIM && fabs(cr) < RQ_FAST_CLIM;

    // pairs p = l, l + L, ...: row tile 8 (p / ncg) + xcd (the channel group p % ncg == cg stays)
    long long pi = l;
    if (pi < nmine * ncg) a_dma(8 * (pi / ncg) + xcd, 0);
    __syncthreads();                                            // constants staged
    const bool fastrq = sUnsafe == 0;
    int buf = 0;
    auto body = [&](auto use_fast) __attribute__((always_inline)) {
        for (; pi < nmine * ncg; pi += L, buf ^= 1) {
            const long long rt = 8 * (pi / ncg) + xcd;
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // this wave's DMA pieces landed; its reads of the other buffer returned
            __syncthreads();
            if (pi + L < nmine * ncg) a_dma(8 * ((pi + L) / ncg) + xcd, buf ^ 1);
            v4i af[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) af[ks] = *(lds_v4i *)(size_t)(fa + buf * ABUF + ks * GW_BM * 32);
            const long long row = rt * GW_BM + wave * 32 + tok;
            const bool live = row < p.M;
            // identity rows of all NCT tiles requested together, ahead of the MFMAs (one exposed latency per row tile, not NCT)
            v4i res[EPI == EPI_RQ16_CH_RES ? NCT : 1][2];
            if (EPI == EPI_RQ16_CH_RES) {
                const int16_t *rp = p.residual + min(row, (long long)p.M - 1) * p.ldc + chbase + 16 * kh;
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    res[ct][0] = *reinterpret_cast<const v4i *>(rp + 32 * ct);
                    res[ct][1] = *reinterpret_cast<const v4i *>(rp + 32 * ct + 8);
                }
            }
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                v16i acc;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const v4i b4 = *reinterpret_cast<const v4i *>(sBias + 32 * ct + 16 * kh + 4 * q);
                    acc[4 * q] = b4[0]; acc[4 * q + 1] = b4[1]; acc[4 * q + 2] = b4[2]; acc[4 * q + 3] = b4[3];
                }
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[ct][ks], af[ks], acc, 0, 0, 0);
                const int ch0 = chbase + 32 * ct + 16 * kh;
                const v4i r0 = res[EPI == EPI_RQ16_CH_RES ? ct : 0][0], r1 = res[EPI == EPI_RQ16_CH_RES ? ct : 0][1];
                int o[16];
#pragma unroll
                for (int v = 0; v < 16; v += 2) {
                    const v2d c2 = *reinterpret_cast<const v2d *>(sC + 32 * ct + 16 * kh + v);
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const double t = (double)acc[v + e] * c2[e];
                        if (decltype(use_fast)::value) o[v + e] = __double2loint(t + (6755399441055744.0 + (OUT8 ? 128.0 : 0.0)));
                        else o[v + e] = min(max(rint_sat_i32(t), OUT8 ? -128 : -32768), OUT8 ? 127 : 32767) + (OUT8 ? 128 : 0);
                    }
                }
                if (OUT8) {
                    // biased to 0 .. 255: v_cvt_pk_i16_i32 + v_sat_pk_u8_i16 clamp while packing; one xor per dword takes the bias out
                    v4i w;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        unsigned p01, p23, b01, b23;
                        const int s0 = o[4 * q], s1 = o[4 * q + 1], s2 = o[4 * q + 2], s3 = o[4 * q + 3];
                        asm("v_cvt_pk_i16_i32 %0, %1, %2" : "=v"(p01) : "v"(s0), "v"(s1));
                        asm("v_cvt_pk_i16_i32 %0, %1, %2" : "=v"(p23) : "v"(s2), "v"(s3));
                        asm("v_sat_pk_u8_i16 %0, %1" : "=v"(b01) : "v"(p01));
                        asm("v_sat_pk_u8_i16 %0, %1" : "=v"(b23) : "v"(p23));
                        w[q] = (int)(__builtin_amdgcn_perm(b23, b01, 0x05040100u) ^ 0x80808080u);
                    }
                    if (live) *reinterpret_cast<v4i *>(reinterpret_cast<int8_t *>(p.out) + row * p.ldc + ch0) = w;
                } else {
                    v4i w0, w1;
#pragma unroll
                    for (int d = 0; d < 8; ++d) {
                        int t0 = min(max(o[2 * d], -32768), 32767), t1 = min(max(o[2 * d + 1], -32768), 32767);
                        if (EPI == EPI_RQ16_CH_RES) {
                            const int rw = d < 4 ? r0[d] : r1[d - 4];
                            const int ra = (int)(short)(rw & 0xffff), rb = rw >> 16;
                            // both terms are integers < 2^31: the sum is the reference's fp64 sum (quant_utils.py:238-244)
                            if (__builtin_expect(res_fast, 1)) { t0 = rq_fast(ra, cr) + rq_fast(t0, cm); t1 = rq_fast(rb, cr) + rq_fast(t1, cm); }
                            else { t0 = rq_lean_wide(ra, cr) + rq_lean_wide(t0, cm); t1 = rq_lean_wide(rb, cr) + rq_lean_wide(t1, cm); }
                        }
                        int pk;
                        asm("v_cvt_pk_i16_i32 %0, %1, %2" : "=v"(pk) : "v"(t0), "v"(t1));      // clamp to 16 bits and pack
                        if (d < 4) w0[d] = pk; else w1[d - 4] = pk;
                    }
                    if (live) {
                        int16_t *op = reinterpret_cast<int16_t *>(p.out) + row * p.ldc + ch0;
                        *reinterpret_cast<v4i *>(op) = w0;
                        *reinterpret_cast<v4i *>(op + 8) = w1;
                    }
                }
            }
        }
    };
    if (fastrq) body(std::true_type{});
    else body(std::false_type{});
}
