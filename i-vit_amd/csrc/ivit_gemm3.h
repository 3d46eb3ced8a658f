// ivit_gemm3.h — persistent, software-pipelined int8 GEMMs for the QuantLinear layers: the production kernels of the
// DeiT / ViT path (planned entry points of include/ivit.h; dispatch in ivit_hip.hip, launch_gemm3).
//
//   C = A (M x K int8) * W^T (W: N x K int8) + bias, fused requant epilogues (a1 + a3 [+ a3 identity] [+ qkv scatter]).
//
//   * gemm_as_kernel (K % 384 == 0, M >= 256, N % 32 == 0): one workgroup per CU (8 waves, 256 registers each) walks
//     (256-token panel, 128-channel tile) units.  K == 384: the panel stays in LDS (A-stationary), only the weight tile
//     streams; K = n * 384: n rounds over the same LDS slots, A streams like the weights.  Described at the kernel.
//   * gemm_ps_kernel (everything else the planned path accepts: K % 64 == 0, K >= 320): two workgroups per CU walk
//     128 x 128 tiles, A and W through a 3-deep ring, LDS-staged output tile.  The first persistent design, kept as
//     the general-shape kernel.
//
// Common to both:
//   * persistent: the (unit, k-step) sequence of a workgroup is ONE stream; operand slices arrive by global_load_lds
//     into an LDS ring that runs two k-steps ahead ACROSS unit boundaries — only the first unit waits for a cold load;
//   * two accumulator sets: while the MFMAs of unit i+1 fill one, the requant arithmetic of unit i drains the other in
//     the same instruction stream;
//   * the bias is the accumulators' initial value; the per-channel multipliers c = m * 2^-e come precomputed from the
//     linear plan (ivit_linear_plan_create), which also PROVES per channel, from sum_k |W[n,k]|, that
//     rne((acc + bias) * c) may be taken as the low dword of fma(double(z), c, 1.5 * 2^52): one v_cvt_f64_i32 + one
//     v_fma_f64 per output element (FMA = true), or v_mul_f64 + v_add_f64 when only the product bound holds;
//   * exact `s_waitcnt vmcnt(n)`: vector-memory instructions retire in issue order, so "slice s has landed" is a
//     count of the instructions issued after it; stores and residual loads never drain the ring.
//
// Arithmetic (bit-exact restatement of quant_utils.py:229-231, see ivit_gemm2.h): swapped MFMA operands (weights =
// "A"), so a lane holds one token and 4 consecutive channels per register quad.
#pragma once
#include "ivit_gemm2.h"

// ablation mask, compile time only (a run-time test would split every k-step into several basic blocks and
// forbid the MFMA / VALU interleaving the kernels are built around): build with -DG3_DBG=<mask> into a scratch
// library and point IVIT_LIB at it.  1 = no operand DMA after the first unit, 2 = no LDS fragment reads / MFMAs,
// 4 = no epilogue arithmetic, 8 = no output stores.  Results are invalid, timing only.
#ifndef G3_DBG
#define G3_DBG 0
#endif
#define G3_NS 3
#define G3_STG_LD 136
#define G3_CONST_BYTES 1536
#define G3_MAGIC 6755399441055744.0
#define G3_MAXNK_ASTAT 6                 // panel slices that fit: K <= 384

template <bool ASTAT> struct G3Cfg {
    static constexpr int BM = ASTAT ? 256 : 128;
    static constexpr int NT = BM * 2;                                  // threads: one wave per 64 x 64 sub-tile
    static constexpr int PANEL = ASTAT ? G3_MAXNK_ASTAT * 16384 : 0;    // stationary A slices [nk][256 x 64]
    static constexpr int STAGE = ASTAT ? 8192 : 16384;                  // ring stage: W slice (+ A slice when streaming)
    static constexpr int RING = G3_NS * STAGE;
    static constexpr int STG = BM * G3_STG_LD;
    static constexpr int SMEM = PANEL + RING + STG + 2 * G3_CONST_BYTES;
};

struct G3Tile {
    int row0, col0;   // first token / channel of the tile
    int par;          // constants buffer of this tile
};

// ---- per-tile constants -> LDS by DMA: c[128] (wave 0), bias[0..63] (wave 1), bias[64..127] (wave 2)
__device__ __forceinline__ void g3_issue_consts(const GemmArgs &p, int col0, char *cst, int wave, int lane) {
    if (wave == 0) {
        int ch = min(col0 + 2 * lane, p.N - 2);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(p.cq + ch),
                                         (__attribute__((address_space(3))) void *)cst, 16, 0, 0);
    } else if (wave < 3) {
        int ch = min(col0 + (wave - 1) * 64 + lane, p.N - 1);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(p.bias + ch),
                                         (__attribute__((address_space(3))) void *)(cst + 1024 + (wave - 1) * 256), 4, 0, 0);
    }
}

// ---- requant of one 32 x 32 accumulator sub-tile (bias already inside) -> staged tile
// OUT8: int8 staging [token][128 channels]; else int16 staging [token][2 x 32 channels] (one 64-channel half)
template <bool OUT8, bool FMA>
__device__ __forceinline__ void g3_requant_subtile(const v16i &acc, const char *cst, char *stg, int ml, int nl0,
                                                   int st_off) {
    const double *cbase = reinterpret_cast<const double *>(cst);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const double *cp = cbase + nl0 + g * 8;
        int o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int z = acc[g * 4 + e];
            const double t = FMA ? __builtin_fma((double)z, cp[e], G3_MAGIC) : ((double)z * cp[e] + G3_MAGIC);
            const int v = __double2loint(t);
            o[e] = OUT8 ? min(max(v, -128), 127) : min(max(v, -32768), 32767);
        }
        if (OUT8) {
            unsigned w01 = __builtin_amdgcn_perm((unsigned)o[1], (unsigned)o[0], 0x0c0c0400u);
            unsigned w23 = __builtin_amdgcn_perm((unsigned)o[3], (unsigned)o[2], 0x0c0c0400u);
            *reinterpret_cast<unsigned *>(stg + ml * G3_STG_LD + st_off + g * 8) = __builtin_amdgcn_perm(w23, w01, 0x05040100u);
        } else {
            v2i w = {(int)__builtin_amdgcn_perm((unsigned)o[1], (unsigned)o[0], 0x05040100u),
                     (int)__builtin_amdgcn_perm((unsigned)o[3], (unsigned)o[2], 0x05040100u)};
            *reinterpret_cast<v2i *>(stg + ml * G3_STG_LD + st_off + g * 16) = w;
        }
    }
}

__device__ __forceinline__ int g3_div(int x, int d, float rcp) {
    int q = (int)((float)x * rcp);
    const int r = x - q * d;
    if (r < 0) --q;
    else if (r >= d) ++q;
    return q;
}

__device__ __forceinline__ v4i g3_stg_read16(const char *sp) {
    v2i lo = *reinterpret_cast<const v2i *>(sp), hi = *reinterpret_cast<const v2i *>(sp + 8);
    return v4i{lo[0], lo[1], hi[0], hi[1]};
}

// plain 16-byte global load the compiler's wait-count model does not see (it would drain the DMA ring with
// vmcnt(0) at the first use); the value is complete after the NEXT k-step's counted wait (issue order below)
__device__ __forceinline__ v4i g3_load16_async(const void *ptr) {
    v4i v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(ptr) : "memory");
    return v;
}

// the same through a buffer resource: lanes whose offset is out of range (>= num_records) read zeros / store nothing,
// so tile edges need neither a branch nor a scratch line
#define GA_OOB 0x80000000u
__device__ __forceinline__ v4i ga_bufload16_async(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, int imm) {
    v4i v;
    if (imm == 0) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(v) : "v"(voff), "s"(rsrc) : "memory");
    else asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:16" : "=v"(v) : "v"(voff), "s"(rsrc) : "memory");
    return v;
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ga_rsrc(const void *ptr) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(ptr), 0, 0x7ffffff0, 0x00020000);
}

// DMA of one 64-column slice of a 256-row A panel (ASTAT): 16 KB, two 16-byte pieces per thread, into the
// XOR-swizzled [row][64] image the fragment reads expect (same image as g2_issue's)
__device__ __forceinline__ void g3_issue_a256(const int8_t *A, int lda, int M, int row0, int k0, char *dst, int tid) {
    const int wave = tid >> 6;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int id = tid + i * 512, row = id >> 2, pos = id & 3;
        const int c = pos ^ ((row >> 2) & 3);
        const int grow = min(row0 + row, M - 1);
        const int8_t *src = A + (long long)grow * lda + k0 + c * 16;
        const unsigned loff = __builtin_amdgcn_readfirstlane((unsigned)(i * 8192 + wave * 1024));
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)(dst + loff), 16, 0, 0);
    }
}
// one 64-column slice of a 128-channel weight tile: 8 KB, one piece per thread of a 512-thread workgroup
__device__ __forceinline__ void g3_issue_w128(const int8_t *B, int ldb, int N, int col0, int k0, char *dst, int tid) {
    const int wave = tid >> 6;
    const int row = tid >> 2, pos = tid & 3;
    const int c = pos ^ ((row >> 2) & 3);
    const int grow = min(col0 + row, N - 1);
    const int8_t *src = B + (long long)grow * ldb + k0 + c * 16;
    const unsigned loff = __builtin_amdgcn_readfirstlane((unsigned)(wave * 1024));
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                     (__attribute__((address_space(3))) void *)(dst + loff), 16, 0, 0);
}

template <int EPI, bool FMA, bool ASTAT>
__global__ __launch_bounds__(G3Cfg<ASTAT>::NT, 2) void gemm_ps_kernel(GemmArgs p) {
    using Cf = G3Cfg<ASTAT>;
    constexpr int NT = Cf::NT, BM = Cf::BM, BMSH = ASTAT ? 8 : 7;
    __shared__ __attribute__((aligned(16))) char smem[Cf::SMEM];
    char *const ring = smem + Cf::PANEL;
    char *const stg = ring + Cf::RING;
    char *const cst0 = stg + Cf::STG;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, half = lane >> 5;
    constexpr bool OUT8 = (EPI == EPI_RQ8_CH || EPI == EPI_QKV);
    constexpr bool RES = (EPI == EPI_RQ16_CH_RES);

    // ---- this workgroup's units.  Unit u = (row block u / tiles_n, channel tile u % tiles_n).
    // ASTAT: one contiguous range [u_first, u_end), step 1 (consecutive units share the stationary panel);
    //        ranges are handed out so that the workgroups of one XCD (b % 8) hold neighbouring ranges.
    // streaming: the unit space is cut into one contiguous range per XCD; the workgroups of an XCD take the
    //        units of its range round-robin (units that share an A panel run at the same time on one L2).
    const int nwg = gridDim.x, bid = blockIdx.x;
    const long long nunits = (long long)((p.M + BM - 1) >> BMSH) * p.tiles_n;
    int u_step, u_end, u_load;
    if (ASTAT) {
        const int r = (nwg & 7) == 0 ? (bid & 7) * (nwg >> 3) + (bid >> 3) : bid;
        u_step = 1;
        u_load = (int)(nunits * r / nwg);
        u_end = (int)(nunits * (r + 1) / nwg);
    } else {
        const int parts = nwg < 8 ? nwg : 8;
        const int part = bid % parts, li = bid / parts;
        u_step = (nwg - part + parts - 1) / parts;
        u_end = (int)(nunits * (part + 1) / parts);
        u_load = (int)(nunits * part / parts) + li;
    }
    if (u_load >= u_end) return;
    const int u_first = u_load;
    const int nk = p.K >> 6;

    const int8_t *A = reinterpret_cast<const int8_t *>(p.A);
    const int8_t *B = p.B;

    // ---- load cursor (runs two k-steps ahead of the MFMAs, across unit boundaries)
    int l_kt = 0, l_slot = 0, l_par = 0;
    int l_tm = u_load / p.tiles_n, l_tn = u_load - l_tm * p.tiles_n;
    bool l_need_a = true;          // ASTAT: this unit brings its panel's slices along
    int infl = 0;                  // DMA instructions per thread of the most recent issue (0: stream exhausted)
    auto issue_next = [&]() __attribute__((always_inline)) {
        if (u_load >= u_end) { infl = 0; return; }
        const int row0 = l_tm << BMSH, col0 = l_tn << 7;
        if (l_kt == 0) g3_issue_consts(p, col0, cst0 + l_par * G3_CONST_BYTES, wave, lane);
        const bool skip = (G3_DBG & 1) && u_load != u_first;     // ablation: no operand traffic after the first unit
        if (ASTAT) {
            if (l_need_a) {
                if (!skip) g3_issue_a256(A, p.lda, p.M, row0, l_kt * G2_BK, smem + l_kt * 16384, tid);
                infl = 3;
            } else {
                infl = 1;
            }
            if (!skip) g3_issue_w128(B, p.ldb, p.N, col0, l_kt * G2_BK, ring + l_slot * Cf::STAGE, tid);
        } else {
            if (!skip) {
                const int8_t *abase[2] = {A + (long long)min(row0 + (tid >> 2), p.M - 1) * p.lda,
                                          A + (long long)min(row0 + ((tid + G2Cfg<128>::THREADS) >> 2), p.M - 1) * p.lda};
                g2_issue<128>(abase, B, p.ldb, p.N, col0, l_kt * G2_BK, ring + l_slot * Cf::STAGE, tid);
            }
            infl = 4;
        }
        if (skip) infl = 0;
        l_slot = (l_slot == G3_NS - 1) ? 0 : l_slot + 1;
        if (++l_kt == nk) {
            l_kt = 0;
            u_load += u_step;
            l_par ^= 1;
            if (ASTAT) {
                l_need_a = false;
                if (++l_tn == p.tiles_n) { l_tn = 0; ++l_tm; l_need_a = true; }
            } else {
                l_tm = u_load / p.tiles_n;
                l_tn = u_load - l_tm * p.tiles_n;
            }
        }
    };
    issue_next();
    issue_next();

    int c_slot = 0;                                         // ring slot of the k-step being multiplied
    const int ml0 = wm * 64 + (lane & 31);                  // token of this lane inside the unit (+32 for i = 1)

    const double cm = p.dy_main.m * p.dy_main.r, cr = p.dy_res.m * p.dy_res.r;
    const float rcpT = 1.0f / (float)(p.T > 0 ? p.T : 1), rcpD = 1.0f / (float)(p.D > 0 ? p.D : 1),
                rcpdh = 1.0f / (float)(p.dh > 0 ? p.dh : 1);

    // ------------------------------------------------------------------------------------------------
    // one unit: K loop of `cur` into accC (HAS_CUR) with the epilogue of `prev` out of accP (HAS_PREV)
    // spread over the first k-steps.  All branches inside are on template parameters or peeled step
    // numbers: the MFMA + epilogue part of a k-step is ONE basic block, so the scheduler interleaves them.
    auto tile_body = [&](auto has_cur_t, auto has_prev_t, v16i(&accC)[2][2], v16i(&accP)[2][2], const G3Tile cur,
                         const G3Tile prev) __attribute__((always_inline)) {
        constexpr bool HAS_CUR = decltype(has_cur_t)::value, HAS_PREV = decltype(has_prev_t)::value;
        const char *pcst = cst0 + prev.par * G3_CONST_BYTES;
        const char *ccst = cst0 + cur.par * G3_CONST_BYTES;
        v4i hold[4];      // 16-bit epilogues: finished 16-byte output pieces waiting for their store slot
        v4i resv[4];      // residual pieces in flight
        (void)hold; (void)resv;

        // ---- epilogue pieces: thread -> (row, 16-byte piece) of the staged tile, 4 pieces per thread
        auto out8_store = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (EPI == EPI_RQ8_CH) {
                    const int id = tid + i * NT, row = id >> 3, c = id & 7;
                    const int grow = prev.row0 + row, gcol = prev.col0 + c * 16;
                    const v4i v = g3_stg_read16(stg + row * G3_STG_LD + c * 16);
                    if (grow < p.M && gcol < p.N && !(G3_DBG & 8))
                        *reinterpret_cast<v4i *>(reinterpret_cast<int8_t *>(p.out) + (long long)grow * p.ldc + gcol) = v;
                } else {
                    // rows-fastest: a wave covers 64 consecutive tokens of one 16-channel piece
                    const int id = tid + i * NT, row = id & (BM - 1), c = id >> BMSH;
                    const int grow = prev.row0 + row, gcol = prev.col0 + c * 16;
                    const v4i v = g3_stg_read16(stg + row * G3_STG_LD + c * 16);
                    if (grow < p.M && gcol < p.N) {
                        // float-reciprocal quotients with one correction step (operands < 2^23)
                        const int which = g3_div(gcol, p.D, rcpD), within = gcol - which * p.D;
                        const int head = g3_div(within, p.dh, rcpdh), d0 = within - head * p.dh;
                        const int b = g3_div(grow, p.T, rcpT), t = grow - b * p.T;
                        const long long bh = (long long)b * p.H + head;
                        if (which < 2 || p.ldv == 0) {          // ldv == 0: v row-major like q and k (round 6)
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
        };
        // 16-bit halves: piece id -> row, 64-channel segment (wn) and 8-channel group inside the half
        auto h16_addr = [&](int i, int j, int &row, int &sc, long long &goff, bool &ok) __attribute__((always_inline)) {
            const int id = tid + i * NT;
            row = id >> 3;
            const int c = id & 7;
            sc = c * 16;                                                  // byte offset in the staged row
            const int grow = prev.row0 + row, gcol = prev.col0 + (c >> 2) * 64 + j * 32 + (c & 3) * 8;
            ok = grow < p.M && gcol < p.N;
            goff = (long long)grow * p.ldc + gcol;
        };
        auto res_issue = [&](int j) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int row, sc; long long goff; bool ok;
                h16_addr(i, j, row, sc, goff, ok);
                const long long safe = ok ? goff : 0;
                resv[i] = g3_load16_async(p.residual + safe);
            }
        };
        auto h16_finish = [&](int j) __attribute__((always_inline)) {     // staged half -> (residual requant-add) -> hold[]
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int row, sc; long long goff; bool ok;
                h16_addr(i, j, row, sc, goff, ok);
                v4i v = g3_stg_read16(stg + row * G3_STG_LD + sc);
                if (RES) {
                    const v4i rs = resv[i];
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const int t0 = (int)(short)(v[w] & 0xffff), t1 = v[w] >> 16;
                        const int r0 = (int)(short)(rs[w] & 0xffff), r1 = rs[w] >> 16;
                        int o0 = rq_fast(r0, cr) + rq_fast(t0, cm);
                        int o1 = rq_fast(r1, cr) + rq_fast(t1, cm);
                        o0 = min(max(o0, -32768), 32767);
                        o1 = min(max(o1, -32768), 32767);
                        v[w] = (o0 & 0xffff) | (o1 << 16);
                    }
                }
                hold[i] = v;
            }
        };
        auto h16_store = [&](int j) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int row, sc; long long goff; bool ok;
                h16_addr(i, j, row, sc, goff, ok);
                if (ok) *reinterpret_cast<v4i *>(reinterpret_cast<int16_t *>(p.out) + goff) = hold[i];
            }
        };
        // vector-memory work of epilogue step CH, issued BEFORE the step's operand DMA
        auto epi_pre = [&](auto ch_t) __attribute__((always_inline)) {
            constexpr int CH = decltype(ch_t)::value;
            if (!HAS_PREV) return;
            if (OUT8) {
                if (CH == 4) out8_store();
            } else {
                if (CH == 0 && RES) res_issue(0);
                if (CH == 2) { h16_store(0); if (RES) res_issue(1); }
                if (CH == 4) h16_store(1);
            }
        };
        // arithmetic of epilogue step CH (interleaves with the step's MFMAs)
        auto epi_main = [&](auto ch_t) __attribute__((always_inline)) {
            constexpr int CH = decltype(ch_t)::value;
            if (!HAS_PREV) return;
            if constexpr (OUT8) {
                if constexpr (CH < 4) {
                    constexpr int i = CH & 1, j = CH >> 1;
                    g3_requant_subtile<true, FMA>(accP[i][j], pcst, stg, ml0 + i * 32, wn * 64 + j * 32 + half * 4,
                                                  wn * 64 + j * 32 + half * 4);
                }
            } else {
                if constexpr (CH == 0 || CH == 2) {
                    constexpr int j = CH >> 1;
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        g3_requant_subtile<false, FMA>(accP[i][j], pcst, stg, ml0 + i * 32, wn * 64 + j * 32 + half * 4,
                                                       wn * 64 + half * 8);
                }
                if (CH == 1) h16_finish(0);
                if (CH == 3) h16_finish(1);
            }
        };

        // ---- one k-step ---------------------------------------------------------------------------
        auto kstep = [&](auto ch_t, const int kt) __attribute__((always_inline)) {
            constexpr int CH = decltype(ch_t)::value;      // epilogue step (6 = none)
            // operands of THIS step complete; the newest step's `infl` DMA instructions stay in flight
            if (!HAS_CUR || infl == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            else if (!ASTAT) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
            else if (infl == 3) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
            // lgkmcnt(0): a raw s_barrier does not wait for this wave's own LDS reads / staging writes
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            epi_pre(ch_t);
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (HAS_CUR) {
                issue_next();
                asm volatile("" ::: "memory");
            }
            if (HAS_CUR && !(G3_DBG & 2)) {
                const char *sB = ring + c_slot * Cf::STAGE + (ASTAT ? 0 : 8192);
                const char *sA = ASTAT ? smem + kt * 16384 : ring + c_slot * Cf::STAGE;
                v4i a[2], b[2];
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const int chunk = kk * 2 + half;
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        a[i] = *reinterpret_cast<const v4i *>(sA + lds_off(wm * 64 + i * 32 + (lane & 31), chunk));
                        b[i] = *reinterpret_cast<const v4i *>(sB + lds_off(wn * 64 + i * 32 + (lane & 31), chunk));
                    }
                    if (CH == 0 && kk == 0) {
                        // first MFMA of the unit: C operand = bias (lane: channels 32j + 8g + 4*half + e)
                        const int *bp = reinterpret_cast<const int *>(ccst + 1024) + wn * 64 + half * 4;
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            v16i init;
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                const v4i bv = *reinterpret_cast<const v4i *>(bp + j * 32 + g * 8);
#pragma unroll
                                for (int e = 0; e < 4; ++e) init[g * 4 + e] = bv[e];
                            }
#pragma unroll
                            for (int i = 0; i < 2; ++i)
                                accC[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(b[j], a[i], init, 0, 0, 0);
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int j = 0; j < 2; ++j)
                                accC[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(b[j], a[i], accC[i][j], 0, 0, 0);
                    }
                }
            }
            if (HAS_CUR) c_slot = (c_slot == G3_NS - 1) ? 0 : c_slot + 1;
            if (!(G3_DBG & 4)) epi_main(ch_t);
        };
        kstep(std::integral_constant<int, 0>{}, 0);
        kstep(std::integral_constant<int, 1>{}, 1);
        kstep(std::integral_constant<int, 2>{}, 2);
        kstep(std::integral_constant<int, 3>{}, 3);
        kstep(std::integral_constant<int, 4>{}, 4);
        if (HAS_CUR) {
            for (int kt = 5; kt < nk; ++kt) kstep(std::integral_constant<int, 6>{}, kt);
        }
    };

    // ---- the unit stream, two accumulator sets alternating
    v16i acc0[2][2], acc1[2][2];
    int u_cur = u_first;
    auto locate = [&](int u, G3Tile &t) __attribute__((always_inline)) {
        const int tm = u / p.tiles_n;
        t.row0 = tm << BMSH;
        t.col0 = (u - tm * p.tiles_n) << 7;
    };
    G3Tile cur{0, 0, 0}, prev{0, 0, 0};
    locate(u_cur, cur);
    auto advance = [&]() __attribute__((always_inline)) {
        prev = cur;
        u_cur += u_step;
        locate(u_cur, cur);
        cur.par ^= 1;
        return u_cur < u_end;
    };
    const std::true_type T{};
    const std::false_type F{};
    tile_body(T, F, acc0, acc1, cur, prev);
    for (;;) {
        if (!advance()) { tile_body(F, T, acc1, acc0, cur, prev); break; }
        tile_body(T, T, acc1, acc0, cur, prev);
        if (!advance()) { tile_body(F, T, acc0, acc1, cur, prev); break; }
        tile_body(T, T, acc0, acc1, cur, prev);
    }
}

// =====================================================================================================
// gemm_as_kernel — K = n * 384: every QuantLinear of the D = 384 / 768 models.
//
// LDS (148.5 KB): A panel 3 x 32 KB (256 tokens x 128-column slice) | weight ring 3 x 16 KB (128 channels x 128
// columns; slot == k-step index) | constants 3 x 1.5 KB (unit i-1's multipliers are in use during unit i, whose first
// step already requests unit i+1's).  A unit is 3 k-steps of 128 columns (n rounds of them for K = n * 384); a k-step is
// four SECTIONS of 4 MFMAs per wave (8 waves as 4 (tokens) x 2 (channels), 64 x 64 per wave).
//
// What the timeline trace (G3_TRACE, tools/gemm3_trace.py) and the ablation builds said, and what the kernel does
// about it (profiles/README.md has the numbers):
//   * a global_load_lds takes ~0.85 us from issue to landing: the ring runs two k-steps ahead, across unit boundaries;
//   * a section that reads fragments or multipliers from the LDS and uses them at once stalls ~200 cycles with at most
//     one MFMA in flight: everything a section consumes is requested one section earlier, and the step's barrier sits
//     between sections 2 and 3, so that section 3 can already request the next step's first fragments;
//   * eight waves requesting their DMA pieces at once queue in the CU's one address path (~300 cycles at the DMA
//     instructions, MFMAs unissued): the waves take turns, one section apart (SIMD mates in different sections);
//   * pure VALU work has no place of its own in the compiler's schedule — it drifted next to its users, leaving
//     MFMA MFMA MFMA MFMA, then all the requant work with the matrix pipe idle: the epilogue is cut into ~8-instruction
//     CHUNKS, one behind each MFMA, pinned by empty volatile statements and scheduling fences;
//   * epilogue without LDS: a lane's 4 packed dwords of a 32 x 32 sub-tile (4-channel runs at 8g + 4*half) become 16
//     consecutive channels of one token with two v_permlane32_swap — straight 16-byte buffer stores (range-checked:
//     tile edges need no branch); the residual is loaded in the same layout, requant-add in registers;
//   * 256 registers hold two accumulator sets (128), double-buffered fragments (32), the residual pieces (32, int16 +
//     residual flavour only) and little else: per-lane addresses are recomputed from an opaque copy of the thread id
//     at each use instead of living across the unit loop.
//   * K > 384, int16 epilogues: the previous unit's epilogue is spread over the unit's first TWO rounds (16 residual
//     registers live instead of 32) — with all 32 the compiler kept loop scalars in VGPRs and spilled them, and every
//     spill reload carries an s_waitcnt vmcnt(0) that drains the DMA ring;
//   * the counted wait is `s_waitcnt vmcnt(C)` with C = the smallest count the steady state produces (4, or 8 with A
//     streaming) whenever the exact count n >= C — only stricter, never wrong — and falls back to smaller immediates at
//     a workgroup's first and last units (ga_wait_vm_fast; the exact jump table stays behind G3_WAIT_TABLE).
// Where it stands (fc1, 50432 x 1536 x 384): 45.6 us = 1.30 POP/s.  The matrix pipe sustains 47 cycles per MFMA on
// random int8 data and the fp64 requant costs ~19 SIMD-cycles per 64 outputs that do NOT hide behind MFMAs of the
// other wave (tools/ubench/overlap.hip: 46.6 cycles per MFMA alone, 72 with this kernel's 1.33 outputs per MFMA):
// ~2300 cycles per k-step are the instruction mix's own floor, ~2800 are measured.
#define GA_NK 3                       // k-steps of 128 columns per round (K = 384 per round)
#define GA_BK 128
#define GA_ASLICE 32768               // 256 tokens x 128 B
#define GA_PANEL (GA_NK * GA_ASLICE)
#define GA_WSTAGE 16384               // 128 channels x 128 B
#define GA_RING (GA_NK * GA_WSTAGE)
#define GA_SMEM (GA_PANEL + GA_RING + 3 * G3_CONST_BYTES)
// timeline instrumentation (compile time, -DG3_TRACE=1 into a scratch library): wave w of workgroup 0 stamps
// s_memtime at six points of every k-step of three units into LDS and dumps them behind the plan's store
// scratch (read back with ivit_debug_plan_scratch).  Points 0..3: start of section 0..3; 4 / 5: before / after the
// counted wait (the barrier follows 5).  The stamps themselves cost ~80 cycles each.
#ifndef G3_TRACE
#define G3_TRACE 0
#endif
// 1: waves 4..7 run at s_setprio 1.  Zero-sum (timeline trace): whichever SIMD mate has the priority finishes its
// step ~500 cycles earlier and waits that much longer at the barrier; kept as a switch for experiments.
#ifndef G3_PRIO
#define G3_PRIO 0
#endif
// 1: counted waits through the jump table (exact); 0: two inline levels (ga_wait_vm_fast)
#ifndef G3_WAIT_TABLE
#define G3_WAIT_TABLE 0
#endif
#define GA_TRACE_BYTES (G3_TRACE ? 8 * 3 * 6 * 4 * 8 : 0)

// wait until at most n (uniform, SGPR) vector-memory instructions of this wave are outstanding, and for all LDS
// traffic.  s_waitcnt only takes an immediate, so the wait is an indexed jump into a table of 48 eight-byte
// {s_waitcnt vmcnt(k) lgkmcnt(0); s_branch end} entries: five scalar instructions and one jump.  (A C++ switch
// over the same cases compiled into a tree of ~40 scalar branches — 900 cycles per k-step, measured.)
#define GA_W1(k) "s_waitcnt vmcnt(" #k ") lgkmcnt(0)\n\ts_branch .Lgaw%=\n\t"
#define GA_W8(a, b, c, d, e, f, g, h) GA_W1(a) GA_W1(b) GA_W1(c) GA_W1(d) GA_W1(e) GA_W1(f) GA_W1(g) GA_W1(h)
__device__ __forceinline__ void ga_wait_vm(int n) {
    n = __builtin_amdgcn_readfirstlane(n);                         // scalar clamp (s_max / s_min)
    n = n < 0 ? 0 : (n > 47 ? 47 : n);
    const int off = __builtin_amdgcn_readfirstlane(n * 8 + 12);   // table starts 12 bytes after the s_getpc result
    asm volatile(
        "s_getpc_b64 vcc\n\t"
        "s_add_u32 vcc_lo, vcc_lo, %0\n\t"
        "s_addc_u32 vcc_hi, vcc_hi, 0\n\t"
        "s_setpc_b64 vcc\n\t"
        GA_W8(0, 1, 2, 3, 4, 5, 6, 7) GA_W8(8, 9, 10, 11, 12, 13, 14, 15) GA_W8(16, 17, 18, 19, 20, 21, 22, 23)
        GA_W8(24, 25, 26, 27, 28, 29, 30, 31) GA_W8(32, 33, 34, 35, 36, 37, 38, 39) GA_W8(40, 41, 42, 43, 44, 45, 46, 47)
        ".Lgaw%=:\n\t"
        :: "s"(off) : "vcc", "scc", "memory");
}

// The same with two inline levels instead of the table: s_waitcnt vmcnt(C) is right whenever n >= C (it only waits for a
// few MORE of the older instructions than necessary), which is the steady state when C is the smallest count the stream
// produces between a slice's request and its use; n >= C2 covers a workgroup's first unit, anything less waits for all.
// Costs a wait, two compares and a short forward branch; the table costs two far jumps (~120 cycles on the timeline).
template <int C, int C2>
__device__ __forceinline__ void ga_wait_vm_fast(int n) {
    n = __builtin_amdgcn_readfirstlane(n);
    asm volatile(
        "s_waitcnt vmcnt(%1) lgkmcnt(0)\n\t"
        "s_cmp_ge_i32 %0, %1\n\t"
        "s_cbranch_scc1 .Lgawf%=\n\t"
        "s_waitcnt vmcnt(%2)\n\t"
        "s_cmp_ge_i32 %0, %2\n\t"
        "s_cbranch_scc1 .Lgawf%=\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        ".Lgawf%=:\n\t"
        :: "s"(n), "n"(C), "n"(C2) : "scc", "memory");
}

// 16-byte LDS read at an explicit LDS byte address (one base register + immediate offset)
__device__ __forceinline__ v4i ga_lds_read16(unsigned lds_addr) {
    typedef __attribute__((address_space(3))) const v4i lds_v4i_t;
    return *(lds_v4i_t *)(size_t)(lds_addr);
}

// swap the upper-half lanes of `a` with the lower-half lanes of `b`
__device__ __forceinline__ void ga_swap32(int &a, int &b) {
    typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
    const v2u_t r = __builtin_amdgcn_permlane32_swap((unsigned)a, (unsigned)b, false, false);
    a = (int)r[0];
    b = (int)r[1];
}

// requant of one quad: four independent cvt -> fma (or mul, add) chains issued abreast.  Left to the compiler the four
// chains are emitted one after the other through ONE temporary register pair — a dependent cvt/fma/med3 sequence per
// element, ~2x the issue-bound time (measured on the timeline trace).
template <bool FMA>
__device__ __forceinline__ void ga_rq4(const int (&z)[4], double c0, double c1, double c2, double c3, int (&o)[4]) {
    double t0, t1, t2, t3;
    const double mg = G3_MAGIC;
    if (FMA) {
        asm volatile("v_cvt_f64_i32 %0, %4\n\tv_cvt_f64_i32 %1, %5\n\tv_cvt_f64_i32 %2, %6\n\tv_cvt_f64_i32 %3, %7\n\t"
            "v_fma_f64 %0, %0, %8, %12\n\tv_fma_f64 %1, %1, %9, %12\n\tv_fma_f64 %2, %2, %10, %12\n\tv_fma_f64 %3, %3, %11, %12"
            : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
            : "v"(z[0]), "v"(z[1]), "v"(z[2]), "v"(z[3]), "v"(c0), "v"(c1), "v"(c2), "v"(c3), "v"(mg));
    } else {
        asm volatile("v_cvt_f64_i32 %0, %4\n\tv_cvt_f64_i32 %1, %5\n\tv_cvt_f64_i32 %2, %6\n\tv_cvt_f64_i32 %3, %7\n\t"
            "v_mul_f64 %0, %0, %8\n\tv_mul_f64 %1, %1, %9\n\tv_mul_f64 %2, %2, %10\n\tv_mul_f64 %3, %3, %11\n\t"
            "v_add_f64 %0, %0, %12\n\tv_add_f64 %1, %1, %12\n\tv_add_f64 %2, %2, %12\n\tv_add_f64 %3, %3, %12"
            : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
            : "v"(z[0]), "v"(z[1]), "v"(z[2]), "v"(z[3]), "v"(c0), "v"(c1), "v"(c2), "v"(c3), "v"(mg));
    }
    o[0] = __double2loint(t0);
    o[1] = __double2loint(t1);
    o[2] = __double2loint(t2);
    o[3] = __double2loint(t3);
}

// Pure VALU work has no place of its own in the instruction stream: the compiler puts it next to its users, across
// scheduling fences.  These empty volatile statements (ordered among themselves and with the fences) tie values to a
// chunk: inputs pinned at the chunk's start, results at its end.
#define GA_PIN1(a) asm volatile("" : "+v"(a))
#define GA_PIN2(a, b) asm volatile("" : "+v"(a), "+v"(b))
#define GA_PIN4(a, b, c, d) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))

struct GaUnit {
    int row0, col0;   // first token / channel
    int cb;           // constants buffer (0..2)
    int need_a;       // first unit of its panel inside this workgroup: brings the panel's A slices along
    int valid;
};

// one 1 KB piece of an operand slice by DMA: uniform 64-bit base + per-lane 32-bit offset (the saddr + voffset form:
// no per-lane 64-bit address arithmetic), LDS destination = uniform base (M0) + lane * 16
__device__ __forceinline__ void ga_dma16(const int8_t *sbase, unsigned voff, unsigned lds_uniform) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(sbase + (size_t)voff),
                                     (__attribute__((address_space(3))) void *)(size_t)lds_uniform, 16, 0, 0);
}

template <int EPI, bool MULTI, bool FMA>
__global__ __launch_bounds__(512, 2) void gemm_as_kernel(GemmArgs p) {
    __shared__ __attribute__((aligned(128))) char smem[GA_SMEM + GA_TRACE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, half = lane >> 5;
    constexpr bool OUT8 = (EPI == EPI_RQ8_CH || EPI == EPI_QKV);
    constexpr bool RES = (EPI == EPI_RQ16_CH_RES);
    if (G3_PRIO && wave >= 4) __builtin_amdgcn_s_setprio(1);
    const unsigned smem_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
    const unsigned ring_lds = smem_lds + GA_PANEL, cst_lds = ring_lds + GA_RING;

    // ---- this workgroup's units; unit u = (256-token panel u / tiles_n, 128-channel tile u % tiles_n).
    // K = 384 (A stationary): one CONTIGUOUS range per workgroup (consecutive units share the panel in LDS); ranges are
    // handed out so that the workgroups of one XCD (b % 8) hold neighbouring ranges.
    // K > 384 (A streams): the unit space is cut into one contiguous range per XCD and the workgroups of an XCD take
    // its units ROUND-ROBIN, so that the channel tiles of one panel run at the same time on CUs that share an L2
    // (a contiguous range per workgroup re-read every panel from beyond the L2: FETCH_SIZE 244 MB vs ~120 MB for fc2).
    const int nwg = gridDim.x, bid = blockIdx.x;
    const long long nunits = (long long)((p.M + 255) >> 8) * p.tiles_n;
    int u_first, u_end, u_step;
    if (!MULTI) {
        const int rr = (nwg & 7) == 0 ? (bid & 7) * (nwg >> 3) + (bid >> 3) : bid;
        u_first = (int)(nunits * rr / nwg);
        u_end = (int)(nunits * (rr + 1) / nwg);
        u_step = 1;
    } else {
        const int parts = nwg < 8 ? nwg : 8, part = bid % parts;
        u_step = (nwg - part + parts - 1) / parts;
        u_first = (int)(nunits * part / parts) + bid / parts;
        u_end = (int)(nunits * (part + 1) / parts);
    }
    if (u_first >= u_end) return;

    const int8_t *A = reinterpret_cast<const int8_t *>(p.A);
    const int8_t *B = p.B;

    // ---- vector-memory bookkeeping (all uniform): `issued` counts this wave's VMEM instructions,
    // mark[k] = count right after the slices of k-step k (ring slot k) were requested
    int issued = 0, mark[GA_NK] = {0, 0, 0}, mark_res[2] = {0, 0}, mark_cst = 0;   // mark_res[h]: after sub-tiles 2h, 2h+1

    // ---- operand DMA.  A k-step is 128 columns: every row slice is one whole 128-byte line, fetched by 8 lanes
    // (64-column slices made every line travel L2 -> L1 twice, once per half).  Piece id -> (row = id >> 3, position
    // id & 7); the LDS image [row][128 B] is lane-linear and XOR-swizzled through the SOURCE chunk
    // (chunk = position ^ ((row >> 1) & 7): conflict-free ds_read_b128 for the MFMA fragments).  Per-lane byte offsets
    // (relative to A / B) are recomputed once per panel / per unit; a slice then costs a scalar M0 write and the
    // instruction itself.
    int a_row0 = 0, w_col0 = 0;     // panel / channel tile the next slice requests belong to (uniform)
    int late_ok = 0;                // 0 until the first barrier of the stream: the prologue requested all three slots
    const int dma_q = (wave + (wave >> 2) * 2 + 3) & 3;     // this wave's section of a refill window
    auto tid_once = [&]() __attribute__((always_inline)) { int t = tid; asm volatile("" : "+v"(t)); return t; };
    // piece id = tid + 512 i: row = (tid >> 3) + 64 i, the same swizzled source chunk for every i.  Per-lane offsets are
    // recomputed for every slice from an opaque copy of tid (a handful of VALU operations per k-step) instead of
    // living in registers across the unit loop, which runs within a few registers of the 256 budget.
    auto set_panel = [&](int row0) __attribute__((always_inline)) { a_row0 = row0; };
    auto set_wtile = [&](int col0) __attribute__((always_inline)) { w_col0 = col0; };
    const unsigned dma_lane0 = wave * 1024;
    // K = nrounds * 384: a unit is `nrounds` rounds of 3 k-steps over the same 3 + 3 LDS slots.  With one round the A
    // slices are stationary (requested only by the first unit of a panel); with more they stream like the weights.
    const int nrounds = MULTI ? p.K / (GA_BK * GA_NK) : 1;
    auto issue_slice = [&](auto s_t, const GaUnit &u, const int round) __attribute__((always_inline)) {
        constexpr int S = decltype(s_t)::value;
        const int kb = round * (GA_BK * GA_NK) + S * GA_BK;
        if (!((G3_DBG & 1) && u.row0 + u.col0 != 0)) {
            const int t = tid_once(), r = t >> 3, ck = ((t & 7) ^ ((t >> 4) & 7)) * 16;
            // scalar slice bases, opaque so that the address stays (SGPR base) + (32-bit VGPR offset)
            const int8_t *sa = A + kb, *sb = B + kb;
            asm volatile("" : "+s"(sa), "+s"(sb));
            if (MULTI || u.need_a) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    ga_dma16(sa, (unsigned)min(a_row0 + r + i * 64, p.M - 1) * (unsigned)p.lda + ck,
                             smem_lds + S * GA_ASLICE + i * 8192 + dma_lane0);
                issued += 4;
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
                ga_dma16(sb, (unsigned)min(w_col0 + r + i * 64, p.N - 1) * (unsigned)p.ldb + ck,
                         ring_lds + S * GA_WSTAGE + i * 8192 + dma_lane0);
            issued += 2;
        }
        mark[S] = issued;
    };
    auto issue_consts = [&](int col0, int cb) __attribute__((always_inline)) {
        g3_issue_consts(p, col0, smem + GA_PANEL + GA_RING + cb * G3_CONST_BYTES, wave, tid_once() & 63);
        if (wave < 3) issued += 1;
        mark_cst = issued;
    };

    // ---- MFMA fragment / constant addresses.  The LDS image is 148 KB but a DS instruction's immediate offset
    // stops at 64 KB: left alone, the compiler keeps one address register per (k-step, operand).  Opaque per-lane
    // bases reach every fragment with an immediate.
    // Group q of a slice differs from group 0 only in chunk bits: address(q) = address(0) ^ (q << 5) (the image is
    // 128-byte aligned), one VALU op per fragment pair instead of a resident register per (operand, q).
    unsigned fa0, fw0;
    {
        const int ra = wm * 64 + (lane & 31), rw = wn * 64 + (lane & 31);
        fa0 = smem_lds + ra * 128 + ((half ^ ((ra >> 1) & 7)) << 4);
        fw0 = ring_lds + rw * 128 + ((half ^ ((rw >> 1) & 7)) << 4);
        asm volatile("" : "+v"(fa0), "+v"(fw0));
    }
    unsigned pc_lds = cst_lds + (wn * 64 + half * 4) * 8;          // this lane's first multiplier, buffer 0
    asm volatile("" : "+v"(pc_lds));

    const double cm = p.dy_main.m * p.dy_main.r, cr = p.dy_res.m * p.dy_res.r;
    const float rcpT = 1.0f / (float)(p.T > 0 ? p.T : 1);
    int tr_unit = -2;     // trace: index of the current unit relative to the first traced one
    auto trace = [&](int kt, int pt) __attribute__((always_inline)) {
        if (G3_TRACE && bid == 0 && lane == 0 && tr_unit >= 0 && tr_unit < 3) {
            unsigned long long *tb = reinterpret_cast<unsigned long long *>(smem + GA_SMEM);
            tb[(wave * 3 + tr_unit) * 24 + kt * 8 + pt] = __builtin_readcyclecounter();
        }
    };
    v4i resv[8];          // residual pieces in flight / waiting for their sub-tile's epilogue: [(j*2 + i)*2 + piece]
    (void)resv;
    typedef double v2d __attribute__((ext_vector_type(2)));
    v2d cqv[2][2];        // multipliers of the next epilogue work: [quad slot][c01 | c23], requested at the END of the
                          // section before (after their previous contents were consumed)
    v4i fa[2][2], fb[2][2];   // MFMA fragments of section G: [G & 1][i | j]
    // accumulators of a unit start at the bias (lane: channels 32j + 8g + 4*half + e)
    auto bias_init = [&](v16i(&acc)[2][2], int cb) __attribute__((always_inline)) {
        // this lane's first bias word: cst + 1024 + (wn * 64 + half * 4) * 4, from the multiplier address
        const unsigned bads = ((pc_lds - cst_lds) >> 1) + cst_lds + 1024 + (unsigned)cb * G3_CONST_BYTES;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const v4i bv = ga_lds_read16(bads + j * 128 + g * 32);
#pragma unroll
                for (int e = 0; e < 4; ++e) { acc[0][j][g * 4 + e] = bv[e]; acc[1][j][g * 4 + e] = bv[e]; }
            }
    };

    // token row / channel column of this lane's 16-channel run in sub-tile (i, j) of a unit
    auto sub_row = [&](const GaUnit &t, int i) __attribute__((always_inline)) { return t.row0 + wm * 64 + i * 32 + (tid_once() & 31); };
    auto sub_col = [&](const GaUnit &t, int j) __attribute__((always_inline)) { return t.col0 + wn * 64 + j * 32 + ((tid_once() >> 1) & 16); };

    // ------------------------------------------------------------------------------------------------
    // one unit: K loop of `cur` into accC (HAS_CUR) with the epilogue of `prev` out of accP (HAS_PREV)
    auto tile_body = [&](auto has_cur_t, auto has_prev_t, auto eh_t, v16i(&accC)[2][2], v16i(&accP)[2][2], const GaUnit cur,
                         const GaUnit prev, const GaUnit next, const int round) __attribute__((always_inline)) {
        constexpr bool HAS_CUR = decltype(has_cur_t)::value, HAS_PREV = decltype(has_prev_t)::value;
        // EH: which half of `prev`'s int16 epilogue this body carries — -1: all four sub-tiles (K = 384, one round per
        // unit); 0 / 1: sub-tiles 0, 1 / 2, 3 (K > 384: the epilogue is spread over the unit's first two rounds, so only
        // 16 residual registers are live at a time — with all 32 the K > 384 flavour spilled loop scalars)
        constexpr int EH = decltype(eh_t)::value;
        constexpr int KB16 = EH < 0 ? 0 : 4 * EH, NK16 = EH < 0 ? 8 : 4;

        // ---- epilogue of `prev`, in pieces small enough to be dealt out between the MFMA groups of a pair.
        // Sub-tile C = (i = C & 1, j = C >> 1).  int8: four quads (requant + pack) and a finish (half-wave exchange +
        // one 16-byte store); int16: two pieces (two quads + exchange + residual requant-add + one store each).
        // The work is cut into CHUNKS of a few VALU instructions, one per MFMA slot (slot = 4 G + m, 48 per round):
        // the section issues MFMA, chunk, MFMA, chunk ... with scheduling fences in between — the pattern
        // tools/ubench/overlap.hip measures (a wave that issues its MFMAs back to back and its VALU work afterwards
        // leaves the matrix pipe idle for the whole VALU stretch: both waves of a SIMD run the same schedule).
        //   int8 (slots 12 C + 3 g + ph of sub-tile C, quad g): ph 0 requant (cvt/fma) + multipliers of the next quad,
        //        ph 1 clamp + pack, ph 2 (g == 3) half-wave exchange + 16-byte store
        //   int16 (slots 14 + 4 k + ph of piece k = 2 C + h2, quads h2 and h2 + 2): ph 0 requant a, ph 1 pack a +
        //        requant b, ph 2 pack b + exchange + residual dwords 0, 1, ph 3 residual dwords 2, 3 + store
        int dpk[4];        // packed dwords of the int8 sub-tile in flight
        int oq[4];         // requantised quad between its two chunks
        int w16[2][2];     // packed int16 pairs of the piece in flight
        v4i v16;           // the piece after the half-wave exchange
        (void)dpk; (void)oq; (void)w16; (void)v16;
        // multipliers of quad (C, g) of unit t -> cqv[slot]
        auto cq_quad = [&](int slot, const GaUnit &t, int C, int g) __attribute__((always_inline)) {
            const unsigned cads = pc_lds + (unsigned)t.cb * G3_CONST_BYTES + (C >> 1) * 256 + g * 64;
            cqv[slot][0] = __builtin_bit_cast(v2d, ga_lds_read16(cads));
            cqv[slot][1] = __builtin_bit_cast(v2d, ga_lds_read16(cads + 16));
        };
        auto rq_asm = [&](auto c_t, int g, const v2d (&c)[2], int (&o)[4]) __attribute__((always_inline)) {
            constexpr int C = decltype(c_t)::value, i = C & 1, j = C >> 1;
            const int z[4] = {accP[i][j][g * 4], accP[i][j][g * 4 + 1], accP[i][j][g * 4 + 2], accP[i][j][g * 4 + 3]};
            ga_rq4<FMA>(z, c[0][0], c[0][1], c[1][0], c[1][1], o);
        };
        auto epi8_finish = [&](auto c_t) __attribute__((always_inline)) {
            constexpr int C = decltype(c_t)::value, i = C & 1, j = C >> 1;
            const int grow = sub_row(prev, i), gcol = sub_col(prev, j);
            const bool ok = grow < p.M && gcol < p.N;
            // lower half-wave: channels 0..15 of the token = {h0.d0, h1.d0, h0.d1, h1.d1}; upper: 16..31
            ga_swap32(dpk[0], dpk[2]);
            ga_swap32(dpk[1], dpk[3]);
            const v4i v = {dpk[0], dpk[2], dpk[1], dpk[3]};
            if (G3_DBG & 8) return;
            typedef unsigned v4u __attribute__((ext_vector_type(4)));
            if constexpr (EPI == EPI_RQ8_CH) {
                const unsigned off = ok ? (unsigned)grow * (unsigned)p.ldc + (unsigned)gcol : GA_OOB;
                __builtin_amdgcn_raw_buffer_store_b128((v4u)v, ga_rsrc(p.out), off, 0, 0);
                issued += 1;
            } else {
                // q / k: [b, head, t, dh] — 16 channels of one head; v^T: [b, head, dh, ldv] byte scatter.
                // The 128-column unit lies inside one of q / k / v and the 32-column group inside one head
                // (D % 128 == 0, dh % 32 == 0: checked by the host), so which / head are uniform.
                const int ucol = prev.col0 + wn * 64 + j * 32;
                const int which = ucol / p.D, within = ucol - which * p.D;
                const int head = within / p.dh, d0 = within - head * p.dh + ((tid_once() >> 1) & 16);
                const int gr = ok ? grow : 0;
                const int b = g3_div(gr, p.T, rcpT), t = gr - b * p.T;
                const unsigned bh = (unsigned)(b * p.H + head);
                if (which < 2 || p.ldv == 0) {                  // ldv == 0: v row-major like q and k — one 16-byte store per lane
                    const unsigned off = ok ? (bh * (unsigned)p.T + (unsigned)t) * (unsigned)p.dh + (unsigned)d0 : GA_OOB;
                    __builtin_amdgcn_raw_buffer_store_b128((v4u)v, ga_rsrc(which == 0 ? p.q : (which == 1 ? p.k : p.vt)), off, 0, 0);
                    issued += 1;
                } else {                                        // v^T: sixteen byte stores per lane (the layout ivit_attn_pv_requant reads)
                    const unsigned off = ok ? (bh * (unsigned)p.dh + (unsigned)d0) * (unsigned)p.ldv + (unsigned)t : GA_OOB;
                    const __amdgpu_buffer_rsrc_t rs = ga_rsrc(p.vt);
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        __builtin_amdgcn_raw_buffer_store_b8((unsigned char)(v[e >> 2] >> (8 * (e & 3))), rs, off, e * p.ldv, 0);
                    issued += 16;
                }
            }
        };
        auto chunk8 = [&](auto s_t) __attribute__((always_inline)) {
            constexpr int S = decltype(s_t)::value, C = S / 12, r = S % 12, g = r / 3, ph = r % 3, qi = C * 4 + g;
            const std::integral_constant<int, C> c_t{};
            if constexpr (ph == 0) {
                if constexpr (qi + 1 < 16) cq_quad((qi + 1) & 1, prev, (qi + 1) >> 2, (qi + 1) & 3);
                rq_asm(c_t, g, cqv[qi & 1], oq);
            } else if constexpr (ph == 1) {
                GA_PIN4(oq[0], oq[1], oq[2], oq[3]);
#pragma unroll
                for (int e = 0; e < 4; ++e) oq[e] = min(max(oq[e], -128), 127);
                const unsigned w01 = __builtin_amdgcn_perm((unsigned)oq[1], (unsigned)oq[0], 0x0c0c0400u);
                const unsigned w23 = __builtin_amdgcn_perm((unsigned)oq[3], (unsigned)oq[2], 0x0c0c0400u);
                dpk[g] = (int)__builtin_amdgcn_perm(w23, w01, 0x05040100u);
                GA_PIN1(dpk[g]);
            } else if constexpr (g == 3) {
                GA_PIN4(dpk[0], dpk[1], dpk[2], dpk[3]);
                epi8_finish(c_t);
            }
        };
        auto pack16 = [&](int (&o)[4], int (&w)[2]) __attribute__((always_inline)) {
            typedef short v2s __attribute__((ext_vector_type(2)));
            w[0] = __builtin_bit_cast(int, (v2s)__builtin_amdgcn_cvt_pk_i16(o[0], o[1]));      // saturating
            w[1] = __builtin_bit_cast(int, (v2s)__builtin_amdgcn_cvt_pk_i16(o[2], o[3]));
        };
        auto res_math = [&](auto c_t, int h2, int q) __attribute__((always_inline)) {
            constexpr int C = decltype(c_t)::value;
            constexpr int RB = (EH < 0 ? C : (C & 1)) * 2;
            const v4i rs = h2 ? resv[RB + 1] : resv[RB];
            const int t0 = (int)(short)(v16[q] & 0xffff), t1 = v16[q] >> 16;
            const int r0 = (int)(short)(rs[q] & 0xffff), r1 = rs[q] >> 16;
            int o0 = rq_fast(r0, cr) + rq_fast(t0, cm);
            int o1 = rq_fast(r1, cr) + rq_fast(t1, cm);
            typedef short v2s __attribute__((ext_vector_type(2)));
            v16[q] = __builtin_bit_cast(int, (v2s)__builtin_amdgcn_cvt_pk_i16(o0, o1));
        };
        constexpr int S16 = 14;      // first slot of the int16 pieces (their residual lands with the step-0 wait)
        auto chunk16 = [&](auto s_t) __attribute__((always_inline)) {
            constexpr int S = decltype(s_t)::value;
            if constexpr (S == S16 - 2) cq_quad(0, prev, KB16 >> 1, 0);
            if constexpr (S >= S16 && S < S16 + 4 * NK16) {
                constexpr int k = KB16 + ((S - S16) >> 2), ph = (S - S16) & 3, C = k >> 1, h2 = k & 1, i = C & 1, j = C >> 1;
                const std::integral_constant<int, C> c_t{};
                if constexpr (ph == 0) {
                    cq_quad(1, prev, C, h2 + 2);
                    rq_asm(c_t, h2, cqv[0], oq);
                } else if constexpr (ph == 1) {
                    GA_PIN4(oq[0], oq[1], oq[2], oq[3]);
                    pack16(oq, w16[0]);
                    GA_PIN2(w16[0][0], w16[0][1]);
                    if constexpr (k + 1 < KB16 + NK16) cq_quad(0, prev, (k + 1) >> 1, (k + 1) & 1);
                    rq_asm(c_t, h2 + 2, cqv[1], oq);
                } else if constexpr (ph == 2) {
                    GA_PIN4(oq[0], oq[1], oq[2], oq[3]);
                    pack16(oq, w16[1]);
                    // lower half-wave: channels 8*h2 .. +7 = {h0.g(h2), h1.g(h2)}; upper half-wave: 16 + the same
                    ga_swap32(w16[0][0], w16[1][0]);
                    ga_swap32(w16[0][1], w16[1][1]);
                    v16 = v4i{w16[0][0], w16[0][1], w16[1][0], w16[1][1]};
                    if constexpr (RES) { res_math(c_t, h2, 0); res_math(c_t, h2, 1); }
                    GA_PIN4(v16[0], v16[1], v16[2], v16[3]);
                } else {
                    GA_PIN4(v16[0], v16[1], v16[2], v16[3]);
                    if constexpr (RES) { res_math(c_t, h2, 2); res_math(c_t, h2, 3); }
                    const int grow = sub_row(prev, i), gcol = sub_col(prev, j);
                    const bool ok = grow < p.M && gcol < p.N;
                    const unsigned off = ok ? ((unsigned)grow * (unsigned)p.ldc + (unsigned)gcol) * 2 + h2 * 16 : GA_OOB;
                    typedef unsigned v4u __attribute__((ext_vector_type(4)));
                    __builtin_amdgcn_raw_buffer_store_b128((v4u)v16, ga_rsrc(p.out), off, 0, 0);
                    issued += 1;
                }
            }
        };
        // residual of sub-tile C of unit `t`, in the epilogue's register layout; requested a pair before its epilogue,
        // inside the same straight-line body (a value an asm load is still filling must not cross a loop edge: the
        // register allocator may copy it)
        auto res_request = [&](auto c_t, const GaUnit &t) __attribute__((always_inline)) {
            constexpr int C = decltype(c_t)::value;
            const int grow = sub_row(t, C & 1), gcol = sub_col(t, C >> 1);
            const bool ok = grow < p.M && gcol < p.N;
            const unsigned off = ok ? ((unsigned)grow * (unsigned)p.ldc + (unsigned)gcol) * 2 : GA_OOB;
            const __amdgpu_buffer_rsrc_t rs = ga_rsrc(p.residual);
            constexpr int RB = (EH < 0 ? C : (C & 1)) * 2;
            resv[RB] = ga_bufload16_async(rs, off, 0);
            resv[RB + 1] = ga_bufload16_async(rs, off, 16);
            issued += 2;
            mark_res[EH < 0 ? (C >> 1) : 0] = issued;
        };

        // fragments of (k-step KT, 32-column half kk) and the 4 MFMAs that consume them
        auto frag_load = [&](auto kt_t, auto q_t, v4i (&a)[2], v4i (&b)[2]) __attribute__((always_inline)) {
            constexpr int KT = decltype(kt_t)::value, q = decltype(q_t)::value;
            const unsigned sA = (fa0 ^ (q << 5)) + KT * GA_ASLICE;
            const unsigned sB = (fw0 ^ (q << 5)) + KT * GA_WSTAGE;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = ga_lds_read16(sA + i * 4096);
                b[i] = ga_lds_read16(sB + i * 4096);
            }
        };
        // ---- one k-step (128 columns) = four SECTIONS of 4 MFMAs (one 32-column group each).  Everything a section
        // consumes from the LDS was requested one section earlier, so no section waits for a read it has just issued:
        //   section (PP, Q), G = 4 PP + Q:  request the fragments of the next section and the multipliers of section
        //   G + 1's epilogue work | 4 MFMAs on the fragments requested by the previous section | epilogue work G of
        //   `prev` (multipliers requested by the previous section).
        // The step's barrier sits between Q = 2 and Q = 3: before it every wave has waited for its own DMA pieces of
        // the NEXT step's slices and for its last fragment reads of THIS step's; after it Q = 3 requests the next
        // step's first fragments and refills this step's ring slot (slice PP of the next round / unit).
        // Epilogue schedule of `prev` over the 12 sections of round 0 — int8: sub-tile G / 3, quads {0, 1} | {2} |
        // {3} + finish; int16 + residual: residual requests in sections 0, 1, piece G - 3 in sections 3..10.
        auto section = [&](auto pp_t, auto q_t) __attribute__((always_inline)) {
            constexpr int PP = decltype(pp_t)::value, Q = decltype(q_t)::value, G = PP * 4 + Q;
            constexpr int KTN = Q == 3 ? (PP + 1) % GA_NK : PP, QN = (Q + 1) & 3;
            if (G3_TRACE && HAS_CUR) { if (G == 0) ++tr_unit; trace(PP, Q); }
            if (HAS_CUR) {
                // Refill of ring slot SL (free once everybody passed step SL's barrier with its reads done) with slice SL
                // of the next round / unit.  The waves take turns: wave w requests its pieces in section dma_q of the
                // window [Q = 3 of step SL, Q = 2 of step SL + 1] — requested by all eight waves at once, the pieces queue
                // in the CU's one address path and every wave sits ~300 cycles at its DMA instructions with its MFMAs
                // unissued (timeline trace).  SIMD mates (w, w + 4) get different sections.
                constexpr int SL = Q == 3 ? PP : (PP + 2) % GA_NK;
                const bool last_round = !MULTI || round + 1 == nrounds;
                if (Q == 3 && PP == 0) {
                    if (last_round) {
                        set_wtile(next.col0);
                        if (MULTI || next.need_a) set_panel(next.row0);
                    }
                    late_ok = 1;
                }
                if (dma_q == Q) {
                    if (Q != 3 && PP == 0) {
                        // slot 2, freed by the previous round's (unit's) last barrier: slice 2 of THIS round
                        if (late_ok) issue_slice(std::integral_constant<int, SL>{}, cur, round);
                    } else {
                        const GaUnit &lu = last_round ? next : cur;
                        if (lu.valid) issue_slice(std::integral_constant<int, SL>{}, lu, last_round ? 0 : round + 1);
                    }
                }
                if (Q == 3 && PP == 0 && last_round && next.valid) issue_consts(next.col0, next.cb);
            }
            __builtin_amdgcn_sched_barrier(0);
            auto slot = [&](auto m_t) __attribute__((always_inline)) {
                constexpr int m = decltype(m_t)::value;
                if (HAS_CUR && !(G3_DBG & 2))
                    accC[m & 1][m >> 1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fb[G & 1][m >> 1], fa[G & 1][m & 1], accC[m & 1][m >> 1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (m == 0) {
                    // the next section's fragments, requested behind this section's first MFMA (whose operand wait then
                    // does not cover them) and three MFMAs ahead of their use
                    if (HAS_CUR) frag_load(std::integral_constant<int, KTN>{}, std::integral_constant<int, QN>{}, fa[(G + 1) & 1], fb[(G + 1) & 1]);
                    if constexpr (RES && HAS_PREV && G < (EH < 0 ? 2 : 1)) {
                        res_request(std::integral_constant<int, (EH < 0 ? 2 * G : 2 * EH)>{}, prev);
                        res_request(std::integral_constant<int, (EH < 0 ? 2 * G : 2 * EH) + 1>{}, prev);
                    }
                }
                if constexpr (HAS_PREV) {
                    if (!(G3_DBG & 4)) {
                        if constexpr (OUT8) chunk8(std::integral_constant<int, G * 4 + m>{});
                        else chunk16(std::integral_constant<int, G * 4 + m>{});
                    } else if constexpr ((G * 4 + m) % 12 == 11) {
                        // ablation: no requant work, one raw store per sub-tile keeps the MFMAs alive
                        constexpr int C = (G * 4 + m) / 12;
                        typedef unsigned v4u __attribute__((ext_vector_type(4)));
                        const v4u v = {(unsigned)accP[C & 1][C >> 1][0], (unsigned)accP[C & 1][C >> 1][5],
                                       (unsigned)accP[C & 1][C >> 1][10], (unsigned)accP[C & 1][C >> 1][15]};
                        __builtin_amdgcn_raw_buffer_store_b128(v, ga_rsrc(p.dummy), (unsigned)(tid_once() & 63) * 16, 0, 0);
                        issued += 1;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            slot(std::integral_constant<int, 0>{});
            slot(std::integral_constant<int, 1>{});
            slot(std::integral_constant<int, 2>{});
            slot(std::integral_constant<int, 3>{});
            if constexpr (HAS_CUR && G == 11) {
                // for the next body: the first quad's multipliers; the next unit's bias into the drained accumulators
                // (harmless before the last round)
                if (OUT8) cq_quad(0, cur, 0, 0);
                // (harmless before the last round; not while sub-tiles 2, 3 of `prev` still wait in accP: EH == 0)
                if constexpr (EH != 0) bias_init(accP, next.cb);
            }
            if (Q == 2) {
                // the next step's slices (requested two steps ago), the residual pieces of the next four sections, at
                // step 2 the next unit's constants; lgkmcnt(0): a raw s_barrier does not wait for this wave's own LDS reads
                int n = 1 << 20;
                if (HAS_CUR) n = issued - mark[(PP + 1) % GA_NK];
                if (HAS_CUR && PP == 2) n = min(n, issued - mark_cst);
                if (RES && HAS_PREV && PP < (EH < 0 ? 2 : 1)) n = min(n, issued - mark_res[PP < 2 ? PP : 0]);
                if (G3_TRACE && HAS_CUR) { __builtin_amdgcn_sched_barrier(0); trace(PP, 4); }
                if (HAS_CUR || (RES && HAS_PREV && PP < (EH < 0 ? 2 : 1))) {
                    if (G3_WAIT_TABLE) ga_wait_vm(n);
                    else ga_wait_vm_fast<MULTI ? 8 : 4, MULTI ? 6 : 2>(n);
                }
                if constexpr (RES && HAS_PREV && PP < (EH < 0 ? 2 : 1))      // tie the residual registers to the wait
                    asm volatile("" : "+v"(resv[PP * 4]), "+v"(resv[PP * 4 + 1]), "+v"(resv[PP * 4 + 2]), "+v"(resv[PP * 4 + 3]));
                if (G3_TRACE && HAS_CUR) trace(PP, 5);
                if (HAS_CUR) __builtin_amdgcn_s_barrier();
            }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        };
        auto kstep = [&](auto pp_t) __attribute__((always_inline)) {
            section(pp_t, std::integral_constant<int, 0>{});
            section(pp_t, std::integral_constant<int, 1>{});
            section(pp_t, std::integral_constant<int, 2>{});
            section(pp_t, std::integral_constant<int, 3>{});
        };
        kstep(std::integral_constant<int, 0>{});
        kstep(std::integral_constant<int, 1>{});
        kstep(std::integral_constant<int, 2>{});
    };

    // ---- the unit stream, two accumulator sets alternating
    auto locate = [&](int u, int cb) __attribute__((always_inline)) {
        GaUnit t;
        const int tm = u / p.tiles_n, tn = u - tm * p.tiles_n;
        t.row0 = tm << 8;
        t.col0 = tn << 7;
        t.cb = cb;
        t.need_a = (u == u_first || tn == 0) ? 1 : 0;
        t.valid = u < u_end ? 1 : 0;
        return t;
    };
    int u_cur = u_first;
    GaUnit cur = locate(u_cur, 0), prev = cur, next = locate(u_cur + u_step, 1);
    v16i acc0[2][2], acc1[2][2];
    // prologue: constants and all three slices of the first unit's round 0; its bias and first fragments
    set_panel(cur.row0);
    set_wtile(cur.col0);
    issue_consts(cur.col0, 0);
    issue_slice(std::integral_constant<int, 0>{}, cur, 0);
    issue_slice(std::integral_constant<int, 1>{}, cur, 0);
    issue_slice(std::integral_constant<int, 2>{}, cur, 0);
    ga_wait_vm(issued - mark[0]);       // constants and slice 0 (requested before slices 1, 2)
    __builtin_amdgcn_s_barrier();
    bias_init(acc0, 0);
    {
        const unsigned sA = fa0, sB = fw0;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            fa[0][i] = ga_lds_read16(sA + i * 4096);
            fb[0][i] = ga_lds_read16(sB + i * 4096);
        }
    }
    auto advance = [&]() __attribute__((always_inline)) {
        prev = cur;
        cur = next;
        u_cur += u_step;
        next = locate(u_cur + u_step, cur.cb == 2 ? 0 : cur.cb + 1);
        return cur.valid != 0;
    };
    const std::true_type T{};
    const std::false_type F{};
    // round 0 of a unit carries the previous unit's epilogue; rounds 1.. (K > 384) only multiply
    const std::integral_constant<int, -1> EA{};
    const std::integral_constant<int, 0> E0{};
    const std::integral_constant<int, 1> E1{};
    constexpr bool SPLIT = MULTI && !OUT8;        // int16 epilogue over the unit's first two rounds (K >= 768)
    // round 0 of a unit carries the previous unit's epilogue (the first two rounds if SPLIT); later rounds only multiply
#define GA_UNIT(HP, aC, aP)                                                                         \
    if constexpr (SPLIT) {                                                                          \
        tile_body(T, HP, E0, aC, aP, cur, prev, next, 0);                                           \
        tile_body(T, HP, E1, aC, aP, cur, prev, next, 1);                                           \
        for (int r = 2; r < nrounds; ++r) tile_body(T, F, EA, aC, aP, cur, prev, next, r);          \
    } else {                                                                                        \
        tile_body(T, HP, EA, aC, aP, cur, prev, next, 0);                                           \
        if (MULTI) for (int r = 1; r < nrounds; ++r) tile_body(T, F, EA, aC, aP, cur, prev, next, r); \
    }
#define GA_DRAIN(aC, aP)                                                                            \
    if constexpr (SPLIT) {                                                                          \
        tile_body(F, T, E0, aC, aP, cur, prev, next, 0);                                            \
        tile_body(F, T, E1, aC, aP, cur, prev, next, 0);                                            \
    } else {                                                                                        \
        tile_body(F, T, EA, aC, aP, cur, prev, next, 0);                                            \
    }
    GA_UNIT(F, acc0, acc1)
    for (;;) {
        if (!advance()) { GA_DRAIN(acc1, acc0) break; }
        GA_UNIT(T, acc1, acc0)
        if (!advance()) { GA_DRAIN(acc0, acc1) break; }
        GA_UNIT(T, acc0, acc1)
    }
#undef GA_UNIT
#undef GA_DRAIN
    if (G3_TRACE && bid == 0) {
        __syncthreads();
        unsigned long long *dst = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(p.dummy) + 1024);
        const unsigned long long *tb = reinterpret_cast<const unsigned long long *>(smem + GA_SMEM);
        for (int k = tid; k < GA_TRACE_BYTES / 8; k += 512) dst[k] = tb[k];
    }
}
