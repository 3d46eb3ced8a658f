// EXPERIMENT RECORD (round 6) — the scaffold the shipped i-vit_amd/csrc/ivit_gemm_ws.h grew out of, in its last experimental state: next task's
// weights requested behind the last K loop of the current one (WS "v3", 38.7 us against 35.5 for reloading at task start), half-a-first DMA with a
// second barrier, a per-SIMD MFMA baton (WS_BATON), s_setprio variants (WS_PRIO), B fragments two k-steps ahead (WS_PF), a hand-staged epilogue
// (WS_PIN = 64), fine cycle stamps (WS_TRACE / WS_FINE).  None of these is in the product; profiles/README.md (round 6, top) has the numbers.
// Compiles stand-alone (template gemm_ws_qkv_kernel<FMA>), qkv flavour only.
// ivit_gemm_ws.h — K = 384 QuantLinear (models/quantization_utils/quant_modules.py:21-80, the qkv flavour of
// models/vit_quant.py:65-74) with the WEIGHTS of a 32-channel tile resident in registers and the tokens of a whole CU in LDS:
//
//   * a workgroup (8 waves, one per CU) owns a contiguous range of 32-token tiles (<= WS_MAXT: 7 x 32 x 384 B = 86 KB of LDS,
//     loaded once by DMA, [64-column block][token][64 B] with the chunk permutation of ivit_mlp_rs.h);
//   * a wave's task is (64-channel slab = one head of q, k or v; one half of the CU's token tiles); the 2 x 12 16-byte A
//     fragments of the slab (24 KB) stay in 96 registers while the wave sweeps its token tiles two at a time (tokens are the
//     B operand, a lane = a token; each B fragment read from LDS feeds two MFMAs — with one channel tile per wave the LDS read
//     port is the bound: 4 SIMDs x 1 KB per 32-cycle MFMA = its 128 B/clk), and are replaced fragment by fragment behind the
//     last sweep's MFMAs: the weights cross L2 -> CU twice per CU and launch;
//   * NO workgroup barrier after the prologue: the two waves of a SIMD drift into anti-phase (one multiplies while the other
//     requantises and stores), which is what gemm_as_kernel's per-k-step barrier forbids (profiles/README.md round 6).
//
// v_mfma_i32_32x32x32_i8, the rows of a weight fragment placed so that accumulator register v of lane (token, h) is channel
// 16 h + v of the tile: 16 consecutive channels per lane, one 16-byte store per token tile.
#pragma once
#include "../../i-vit_amd/csrc/ivit_device.h"
#include <type_traits>

#define WS_K 384
#define WS_KS 12                                 // k-steps of 32
#define WS_MAXT 7                                // 32-token tiles of a panel in LDS
#define WS_TOK (WS_MAXT * 32)
#define WS_KBLK (WS_TOK * 64)
#define WS_SOFF (6 * WS_KBLK)                     // output row offset of each token of the panel (int)
#define WS_SBIAS (WS_SOFF + WS_TOK * 4)           // the layer's bias (int32 x N) and multipliers (double x N): no vector-memory load
#define WS_MAXN 1536                             // between two stores of the steady state (loads and stores share vmcnt)
#define WS_SCQ (WS_SBIAS + WS_MAXN * 4)
#define WS_SLOCK (WS_SCQ + WS_MAXN * 8)           // one MFMA baton per SIMD
#define WS_SMEM (WS_SLOCK + 64)
#define WS_THREADS 512
#define WS_MAGIC 6755399441055744.0
#ifndef WS_PRIO
#define WS_PRIO 0                                // 1: waves 0-3 at priority 3 (static); 2: priority 3 inside the K loop, 0 in the epilogue
#endif
#ifndef WS_PIN
#define WS_PIN 16                                // outputs per scheduling group of the epilogue (4 or 16)
#endif
#ifndef WS_FINE
#define WS_FINE 0
#endif
#ifndef WS_TRACE
#define WS_TRACE 0
#endif
#ifndef WS_PF
#define WS_PF 1                                  // k-steps the B fragments are requested ahead of their MFMAs
#endif
#ifndef WS_BATON
#define WS_BATON 1                               // the two waves of a SIMD take turns in the K loop (one multiplies, the other requantises)
#endif
#ifndef WS_ABL
#define WS_ABL 0                                 // probe builds (results invalid): 1 no epilogue arithmetic, 2 no stores, 4 no MFMAs
#endif

struct WsArgs {
    const int8_t *x;          // [M][384]
    const v4i *wf;            // swizzled weights: fragment (ct * 12 + ks) * 64 + lane
    const int32_t *bias;      // [N]
    const double *cq;         // [N]
    int8_t *q, *k, *v;        // [B*H][T][64] each
    int M, N, T, H;
    void *dummy;              // >= 1 KB: where the lanes of rows >= M store
    long long *trace;         // WS_TRACE builds: [8 waves][32 stamps] of workgroup 0
};

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

template <bool FMA>
__global__ __launch_bounds__(WS_THREADS, 2) void gemm_ws_qkv_kernel(WsArgs p) {
    extern __shared__ __attribute__((aligned(256))) char sm[];
    typedef double v2d __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(3))) char lds_c;
    typedef __attribute__((address_space(3))) v4i lds_v4i;
    typedef __attribute__((address_space(3))) int lds_i32;
    typedef __attribute__((address_space(3))) v2d lds_v2d;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned sm_lds = (unsigned)(size_t)(lds_c *)sm;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, tok = lane & 31, kh = lane >> 5, e = kh ^ ws_g(tok);
    if (WS_PRIO == 1 && wave < 4) __builtin_amdgcn_s_setprio(3);
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
    if (tid < 16) reinterpret_cast<int *>(sm + WS_SLOCK)[tid] = 0;
    const unsigned lock = sm_lds + WS_SLOCK + (wave & 3) * 4;
    auto baton_take = [&]() __attribute__((always_inline)) {
        unsigned v, one = 1, tmp;
        unsigned long long save;
        asm volatile("s_mov_b64 %2, exec\n\t"
                     "s_mov_b64 exec, 1\n"
                     ".Lbt%=:\n\t"
                     "ds_wrxchg_rtn_b32 %0, %3, %4\n\t"
                     "s_waitcnt lgkmcnt(0)\n\t"
                     "v_readfirstlane_b32 %1, %0\n\t"
                     "s_cmp_eq_u32 %1, 0\n\t"
                     "s_cbranch_scc1 .Lbd%=\n\t"
                     "s_sleep 2\n\t"
                     "s_branch .Lbt%=\n"
                     ".Lbd%=:\n\t"
                     "s_mov_b64 exec, %2"
                     : "=&v"(v), "=&s"(tmp), "=&s"(save) : "v"(lock), "v"(one) : "memory", "scc");
    };
    auto baton_give = [&]() __attribute__((always_inline)) {
        unsigned zero = 0;
        unsigned long long save;
        asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\tds_write_b32 %1, %2\n\ts_mov_b64 exec, %0" : "=&s"(save) : "v"(lock), "v"(zero) : "memory");
    };

    for (int t0 = t_beg; t0 < t_end; t0 += WS_MAXT) {
        const int n_own = min(WS_MAXT, t_end - t0);
        if (t0 != t_beg) __syncthreads();            // a later panel: every wave is done with the previous one
        // tasks: (slab, token half); half a = tiles [0, na), half b = [na, n_own); a wave's first tasks are of half a
        const int na = (n_own + 1) >> 1, ntask = n_own > 1 ? 2 * ncp : ncp;
        // ---- the panel's tokens: global -> LDS by DMA, 16 tokens x 4 chunk slots per instruction (source chunk = slot ^ g);
        // half a, the first slab's weights, half b: the first task starts when half a has landed
        auto dma = [&](int tg) __attribute__((always_inline)) {
            const int tokl = tg * 16 + (lane >> 2), c = (lane & 3) ^ ws_g(tokl);
            const long long grow = min((long long)t0 * 32 + tokl, (long long)p.M - 1);
            const int8_t *src = p.x + grow * WS_K + c * 16;
#pragma unroll
            for (int kb = 0; kb < 6; ++kb) {
                const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(kb * WS_KBLK + tg * 1024));
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + kb * 64),
                                                 (__attribute__((address_space(3))) void *)(sm + dst), 16, 0, 0);
            }
        };
        auto w_load = [&](const char *slab, int f) __attribute__((always_inline)) {      // fragment f of a slab (uniform base + lane offset)
            return *reinterpret_cast<const v4i *>(slab + lane16 + f * 1024);
        };
        auto slab_of = [&](int task) { return reinterpret_cast<const char *>(p.wf + (size_t)(task >= ncp ? task - ncp : task) * 2 * WS_KS * 64); };
        if (wave < 2 * na) dma(wave);                                   // na <= 4: at most one group per wave
        v4i W[2][WS_KS];
        {
            const char *wq = slab_of(wave < ntask ? wave : 0);
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int ks = 0; ks < WS_KS; ++ks) W[c][ks] = w_load(wq, c * WS_KS + ks);
        }
        const bool dma_b = 2 * na + wave < 2 * n_own;                   // n_own - na <= 3: at most one group per wave
        if (dma_b) dma(2 * na + wave);
        // output row offset of every token of the panel: (b * H * T + t_in_image) * 64
        if (tid < WS_TOK) {
            const int row = min(t0 * 32 + tid, p.M - 1), b = row / p.T;
            reinterpret_cast<int *>(sm + WS_SOFF)[tid] = (b * p.H * p.T + (row - b * p.T)) * 64;
        }
        if (dma_b) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");    // everything but this wave's six half-b instructions
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        stamp();

        const unsigned fa0 = sm_lds + tok * 64 + e * 16, fa1 = sm_lds + tok * 64 + (e ^ 2) * 16;
        // The K loop of a sweep: NT token tiles x 2 channel tiles, B fragments WS_PF steps ahead
        auto mfma_part = [&](auto nt_c, const int tb, const int chb, v16i(&acc)[2][2]) __attribute__((always_inline)) {
            constexpr int NT = decltype(nt_c)::value;
            v4i bf[WS_PF + 1][NT];
            const unsigned fb0 = fa0 + tb * 2048, fb1 = fa1 + tb * 2048;
            stamp();
            auto load_b = [&](int ks, int slot) __attribute__((always_inline)) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    bf[slot][t] = *(lds_v4i *)(size_t)(((ks & 1) ? fb1 : fb0) + (ks >> 1) * WS_KBLK + t * 2048);
            };
#pragma unroll
            for (int i = 0; i < WS_PF; ++i) load_b(i, i);
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const v4i b4 = *(lds_v4i *)(size_t)(sm_lds + WS_SBIAS + (chb + 32 * c + 4 * q4) * 4);
#pragma unroll
                    for (int t = 0; t < NT; ++t) { acc[c][t][4 * q4] = b4[0]; acc[c][t][4 * q4 + 1] = b4[1]; acc[c][t][4 * q4 + 2] = b4[2]; acc[c][t][4 * q4 + 3] = b4[3]; }
                }
            if (WS_BATON) baton_take();
#pragma unroll
            for (int ks = 0; ks < WS_KS; ++ks) {
                __builtin_amdgcn_sched_barrier(0);
                if (WS_FINE && (ks & 3) == 0 && ks) stamp();
                if (ks + WS_PF < WS_KS) load_b(ks + WS_PF, (ks + WS_PF) % (WS_PF + 1));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        if (!(WS_ABL & 4)) acc[c][t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(W[c][ks], bf[ks % (WS_PF + 1)][t], acc[c][t], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (WS_BATON) baton_give();
            stamp();
        };
        // requant to 8 bits: fma(z, c, magic + 128) leaves Q + 128 in the low dword; the two packs saturate to [0, 255] =
        // clamp(Q, -128, 127) + 128; the xor takes the bias off again.  One (channel tile, token tile) at a time: sixteen
        // channels of a token per lane, one 16-byte store
        auto epi_part = [&](auto nt_c, const int tb, const int chb, int8_t *obase, v16i(&acc)[2][2]) __attribute__((always_inline)) {
            constexpr int NT = decltype(nt_c)::value;
            int toff[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) toff[t] = *(lds_i32 *)(size_t)(sm_lds + WS_SOFF + ((tb + t) * 32 + tok) * 4);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                v2d cqv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) cqv[j] = *(lds_v2d *)(size_t)(sm_lds + WS_SCQ + (chb + 32 * c + 2 * j) * 8);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    v4i o4;
                    if (WS_PIN == 64) {
                        // staged by hand: eight conversions, eight FMAs, the packs — an instruction's operands are several
                        // instructions old (the compiler's own order puts each pack right behind the FMAs it reads)
#pragma unroll
                        for (int h8 = 0; h8 < 2; ++h8) {
                            double d[8];
#pragma unroll
                            for (int i = 0; i < 8; ++i) d[i] = (double)acc[c][t][8 * h8 + i];
                            asm volatile("" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]));
#pragma unroll
                            for (int i = 0; i < 8; ++i) {
                                const double m = cqv[4 * h8 + (i >> 1)][i & 1];
                                d[i] = FMA ? __builtin_fma(d[i], m, WS_MAGIC + 128.0) : (d[i] * m + (WS_MAGIC + 128.0));
                            }
                            asm volatile("" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]));
                            unsigned pk[4], sb[4];
#pragma unroll
                            for (int i = 0; i < 4; ++i) asm volatile("v_cvt_pk_i16_i32 %0, %1, %2" : "=v"(pk[i]) : "v"(__double2loint(d[2 * i])), "v"(__double2loint(d[2 * i + 1])));
#pragma unroll
                            for (int i = 0; i < 4; ++i) asm volatile("v_sat_pk_u8_i16 %0, %1" : "=v"(sb[i]) : "v"(pk[i]));
#pragma unroll
                            for (int i = 0; i < 2; ++i) o4[2 * h8 + i] = (int)(__builtin_amdgcn_perm(sb[2 * i + 1], sb[2 * i], 0x05040100u) ^ 0x80808080u);
                        }
                        asm volatile("" : "+v"(o4));
                    } else {
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        int o[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const double m = cqv[2 * q4 + (i >> 1)][i & 1];
                            const double tq = FMA ? __builtin_fma((double)acc[c][t][4 * q4 + i], m, WS_MAGIC + 128.0)
                                                  : ((double)acc[c][t][4 * q4 + i] * m + (WS_MAGIC + 128.0));
                            o[i] = (WS_ABL & 1) ? acc[c][t][4 * q4 + i] : __double2loint(tq);
                        }
                        unsigned p01, p23, b01, b23;
                        asm("v_cvt_pk_i16_i32 %0, %1, %2" : "=v"(p01) : "v"(o[0]), "v"(o[1]));
                        asm("v_cvt_pk_i16_i32 %0, %1, %2" : "=v"(p23) : "v"(o[2]), "v"(o[3]));
                        asm("v_sat_pk_u8_i16 %0, %1" : "=v"(b01) : "v"(p01));
                        asm("v_sat_pk_u8_i16 %0, %1" : "=v"(b23) : "v"(p23));
                        o4[q4] = (int)(__builtin_amdgcn_perm(b23, b01, 0x05040100u) ^ 0x80808080u);
                        if (WS_PIN == 4) asm volatile("" : "+v"(o4[q4]));
                    }
                    if (WS_PIN == 16) asm volatile("" : "+v"(o4));       // pinned per tile: left alone, the optimiser converts every accumulator first
                    }
                    const int row = (t0 + tb + t) * 32 + tok;
                    if (WS_ABL & 2) asm volatile("" ::"v"(o4));
                    else *reinterpret_cast<v4i *>(row < p.M ? obase + toff[t] + 32 * c : (int8_t *)p.dummy + lane16) = o4;
                }
                if (WS_FINE) stamp();
            }
        };
        // One task.  Its last K loop is followed by the NEXT task's weights into the same registers (one unconditional block per
        // iteration: inside the sweep-shape branches the compiler copies all 96 registers at the back edge), then by its epilogue
        auto do_task = [&](const int task) __attribute__((always_inline)) {
            const int half = task >= ncp, cp = task - half * ncp;
            const int te = half ? n_own : na;
            int tb = half ? na : 0;
            const int chb = 64 * cp + 16 * kh;
            const int which = cp / ncp3;
            int8_t *obase = (which == 0 ? p.q : which == 1 ? p.k : p.v) + (size_t)(cp - which * ncp3) * p.T * 64 + 16 * kh;
            v16i acc[2][2];
            if (te - tb > 2) {
                mfma_part(std::integral_constant<int, 2>{}, tb, chb, acc);
                epi_part(std::integral_constant<int, 2>{}, tb, chb, obase, acc);
                tb += 2;
            }
            const bool two = te - tb == 2;
            if (two) mfma_part(std::integral_constant<int, 2>{}, tb, chb, acc);
            else mfma_part(std::integral_constant<int, 1>{}, tb, chb, acc);
            {
                const char *wn = slab_of(task + 8 < ntask ? task + 8 : task);
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int ks = 0; ks < WS_KS; ++ks) W[c][ks] = w_load(wn, c * WS_KS + ks);
            }
            if (two) epi_part(std::integral_constant<int, 2>{}, tb, chb, obase, acc);
            else epi_part(std::integral_constant<int, 1>{}, tb, chb, obase, acc);
        };
        if (wave < ntask) do_task(wave);
        // half b: every wave's DMA has landed by now (its own: vmcnt below — the weights requested inside the first task retire first)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int task = wave + 8; task < ntask; task += 8) do_task(task);
        stamp();
    }
}
