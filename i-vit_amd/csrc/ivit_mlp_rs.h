// ivit_mlp_rs.h — the fused Mlp of ivit_mlp.h (same arguments, same plan constants, same integers) with the eight waves split
// by ROLE instead of running every phase in lock-step (models/layers_quant.py:144-153, vit_quant.py:141-142):
//
//   waves 0-3, "producers", one per SIMD: fc1 + qact_gelu (8 bit) of unit u + 1
//   waves 4-7, "consumers", one per SIMD: ShiftGELU (+ qact1), fc2 + qact2 (16 bit) + qact4 with the identity branch of unit u
//
// so that on every SIMD one wave's MFMAs run beside the other wave's VALU / LDS / global work (a wave's own VALU does not
// overlap its own MFMAs on this chip, another wave's does: profiles/README.md round 3, "MFMA / VALU co-residence").  In
// mlp384_kernel all eight waves are in the same phase: the matrix pipe idles through ShiftGELU and both epilogues.
//
// v_mfma_i32_32x32x32_i8 throughout (a lone wave issues it at the full rate; a lone wave reaches half the rate with 16x16x64),
// weights as the A operand (rows = channels), 32 tokens as the B operand (a lane = a token).  The rows of a weight fragment are
// placed at plan time so that MFMA row q*8 + h*4 + i is channel h*16 + q*4 + i of the 32-channel tile: accumulator register v of
// lane (token, h) is then channel 16 h + v — sixteen CONSECUTIVE channels per lane: one ds_write_b128 per token tile into the
// hidden tile, two 16-byte global stores per token tile in fc2's epilogue.
//
// One hidden tile (<= 80 tokens x 1536 B = 120 KB; two do not fit and smaller units pay the 1.18 MB weight sweep more often),
// so a producer may only overwrite the 128-channel slice r of the hidden tile once every consumer is past it in fc2's K loop of
// the PREVIOUS unit: the producer parks up to RS_HD requantised rounds in registers (4 dwords per token tile and round) and
// writes round r - RS_HD behind the K loop of round r.  Hand-over by monotonically increasing LDS counters (no workgroup barrier
// after the prologue): F_H hidden tile of unit u complete (4 producers), F_G ShiftGELU of unit u complete (4 consumers),
// F_R[r] consumers past slice r in fc2 of unit u, F_A producers done with the activation tile, F_D next activation tile landed.
// An LDS instruction stream of one wave executes in order, so "data accesses, then ds_add" / "ds_read counter, then data
// accesses" need no fences beyond keeping the compiler from reordering them (asm volatile + memory clobber).
//
// LDS images: [64-column block][token][64 B]; the four 16-byte chunks of a token's 64 B are permuted by
// g(token) = ((token >> 1) & 3) ^ gray((token >> 3) & 3): conflict-free both for the B-fragment ds_read_b128 (lane groups
// {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and their upper-half twins: MI355X_MICROARCH.md, LDS) and for the producers'
// ds_write_b128 (eight consecutive tokens per group, 32 banks).
#pragma once
#include "ivit_mlp.h"
#include "ivit_layernorm.h"

#define RS_T 80                                  // token rows of a unit in LDS (5 tiles of 16)
#define RS_KBLK (RS_T * 64)
#define RS_SH 0                                  // hidden tile [24][80][64 B]
#define RS_SA (MLP_KS2 * RS_KBLK)                // activation tile [6][80][64 B]
#define RS_STAB (RS_SA + MLP_KS1 * RS_KBLK)      // one ShiftGELU table line per consumer half-wave
#define RS_SFLAG (RS_STAB + 32 * 256)            // two ShiftGELU table lines per half-wave (sixteen half-waves)
#define RS_SDUMMY (RS_SFLAG + 128)               // 64 x 16 B: where the lanes of non-existent token rows store
#define RS_SMAX (RS_SDUMMY + 1024)                // row maxima of the hidden tile (int32 per token, 96 slots)
#define RS_SMEM (RS_SMAX + 384)
#define RS_THREADS 512
#define RS_F_H 0
#define RS_F_G 1
#define RS_F_A 2
#define RS_F_D 3
#define RS_F_R 4                                 // .. 15
#define RS_F_L 16                                // LNH = 2: consumers done with the LayerNorm of units 1 ..
#ifndef RS_WR
#define RS_WR 9                                  // producer: slots of the weight-fragment ring (RS_WR - 1 k-steps in flight); divides 12 * RS_HD
#endif
#ifndef RS_HD
#define RS_HD 6                                  // producer: requantised rounds parked in registers; divides 12
#endif
#ifndef RS_WD2
#define RS_WD2 3                                 // consumer: k-steps of weight fragments in flight (3 fragments each)
#endif
#ifndef RS_TRACE
#define RS_TRACE 0
#endif
#ifndef RS_ABL
#define RS_ABL 0                                 // probe builds, timing only (results invalid): 1 no ShiftGELU body, 2 no epilogue arithmetic,
#endif                                           // 4 no producer requant arithmetic
#ifndef RS_HELP
#define RS_HELP 1                                // the consumers multiply rounds 6..11 of the FIRST unit's fc1 (they have nothing else to do yet)
#endif
#ifndef RS_GSPLIT
#define RS_GSPLIT 1                              // ShiftGELU of a unit on all sixteen half-waves (0: the consumers' eight)
#endif
#ifndef RS_DBG_ROLE
#define RS_DBG_ROLE 3                            // probe builds: 1 = producers only, 2 = consumers only (resource usage per role)
#endif
#ifndef RS_DBG_NT
#define RS_DBG_NT 3                              // probe builds: 1 = two-tile bodies only, 2 = three-tile bodies only
#endif

__device__ __forceinline__ int rs_chan_of_row(int rho) { return ((rho >> 2) & 1) * 16 + (rho >> 3) * 4 + (rho & 3); }
__device__ __forceinline__ int rs_g(int tok) { return ((tok >> 1) & 3) ^ ((tok >> 3) & 3) ^ ((tok >> 4) & 1); }

// fc1 weights [1536][384] -> fragments of 64 lanes x 16 B in consumption order: fragment (r * 12 + ks) * 4 + w is what
// producer w multiplies in k-step ks (32 columns) of round r (channels 128 r + 32 w ...): one contiguous 4 KB window per step
__global__ __launch_bounds__(256) void rs_swizzle_w1_kernel(const int8_t *__restrict__ w, v4i *__restrict__ wf) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < 144 * 4 * 64; i += gridDim.x * 256) {
        const int l = i & 63, f = i >> 6, pw = f & 3, s = f >> 2, r = s / 12, ks = s - r * 12;
        const int ch = 128 * r + 32 * pw + rs_chan_of_row(l & 31);
        wf[i] = *reinterpret_cast<const v4i *>(w + (size_t)ch * MLP_C + 32 * ks + 16 * (l >> 5));
    }
}
// fc2 weights [384][1536]: fragment (ks * 4 + j) * 3 + ct = consumer j, channel tile ct (channels 96 j + 32 ct ...), k-step ks
__global__ __launch_bounds__(256) void rs_swizzle_w2_kernel(const int8_t *__restrict__ w, v4i *__restrict__ wf) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < 48 * 12 * 64; i += gridDim.x * 256) {
        const int l = i & 63, f = i >> 6, ct = f % 3, j = (f / 3) & 3, ks = f / 12;
        const int ch = 96 * j + 32 * ct + rs_chan_of_row(l & 31);
        wf[i] = *reinterpret_cast<const v4i *>(w + (size_t)ch * MLP_HD + 32 * ks + 16 * (l >> 5));
    }
}

// Both hand-over primitives are single asm blocks: straight-line code for the register allocator (as C++ the spin loop was
// unrolled nine times and every `if (lane == 0)` split a basic block: 1.4 K spilled registers in the producers).
__device__ __forceinline__ void rs_signal(unsigned flag_addr) {            // lane 0 adds 1 (flag_addr is wave-uniform)
    unsigned long long save;
    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\tds_add_u32 %1, %2\n\ts_mov_b64 exec, %0"
                 : "=&s"(save) : "v"(flag_addr), "v"(1u) : "memory");
}
// spin until the LDS counter reaches `target` (the counters only grow).  A hand-over that never comes is a bug: trap loudly —
// after 2^28 polls of >= 64 cycles each (~10 s: far beyond any stall a debugger, a profiler or throttled clocks produce, short
// enough that a real deadlock ends the launch instead of wedging the queue)
__device__ __forceinline__ void rs_wait(unsigned flag_addr, unsigned target) {
    unsigned v, cnt, tmp;
    asm volatile("s_mov_b32 %1, 0\n"
                 ".Lrsw%=:\n\t"
                 "ds_read_b32 %0, %3\n\t"
                 "s_waitcnt lgkmcnt(0)\n\t"
                 "v_readfirstlane_b32 %2, %0\n\t"
                 "s_sub_i32 %2, %2, %4\n\t"
                 "s_cmp_ge_i32 %2, 0\n\t"
                 "s_cbranch_scc1 .Lrsd%=\n\t"
                 "s_sleep 1\n\t"
                 "s_add_u32 %1, %1, 1\n\t"
                 "s_cmp_lt_u32 %1, 0x10000000\n\t"
                 "s_cbranch_scc1 .Lrsw%=\n\t"
                 "s_trap 2\n"
                 ".Lrsd%=:"
                 : "=&v"(v), "=&s"(cnt), "=&s"(tmp) : "v"(flag_addr), "s"(target) : "memory", "scc");
}

// LNH: 0 = the activations are 8-bit (p.x); 1 = norm2 of every row of the workgroup first, by all eight waves; 2 = only the first unit's rows
// first, the rest by the four consumer waves beside the producers' fc1 of the first unit (the consumers then do not help with that fc1)
template <bool FMA, int LNH = 0>
__global__ __launch_bounds__(RS_THREADS, 2) void mlp384rs_kernel(MlpArgs p) {
    extern __shared__ __attribute__((aligned(256))) char sm[];
    typedef double v2d __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(3))) char lds_c;
    typedef __attribute__((address_space(3))) v4i lds_v4i;
    typedef __attribute__((address_space(3))) unsigned lds_u32;
    typedef __attribute__((address_space(3))) int lds_i32;
    typedef __attribute__((address_space(3))) v2i lds_v2i;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned sm_lds = (unsigned)(size_t)(lds_c *)sm;
    const unsigned fl = sm_lds + RS_SFLAG;

    // ---- this workgroup's units (the schedule of mlp384_kernel)
    const long long ntiles = (p.M + 15) >> 4;
    const long long t_beg = ntiles * blockIdx.x / gridDim.x, t_end = ntiles * (blockIdx.x + 1) / gridDim.x;
    const int n_own = (int)(t_end - t_beg);
    const long long nfix = (ntiles + MLP_TT - 2) / (MLP_TT - 1);
    const int nu = p.balanced ? (n_own + MLP_TT - 1) / MLP_TT : (int)((nfix - (long long)blockIdx.x + gridDim.x - 1) / gridDim.x);
    if (nu <= 0) return;
    auto unit_tile0 = [&](int i) -> long long {
        if (p.balanced) return t_beg + (long long)n_own * i / nu;
        return min(((long long)blockIdx.x + (long long)i * gridDim.x) * (MLP_TT - 1), ntiles);
    };
    auto unit_ntt = [&](int i) -> int {
        if (i >= nu) return 0;
        if (p.balanced) return (int)(unit_tile0(i + 1) - unit_tile0(i));
        return (int)min((long long)(MLP_TT - 1), ntiles - unit_tile0(i));
    };
    auto stamp = [&](int u, int pt) __attribute__((always_inline)) {
        if (RS_TRACE) {
            if (blockIdx.x == 0 && u < 4 && (threadIdx.x & 63) == 0) p.trace[(u * 8 + wave) * 8 + pt] = __builtin_readcyclecounter();
        }
    };
    // cumulative hand-over targets: in the first unit the consumers produce too (RS_HELP)
    constexpr bool HELP = RS_HELP && LNH != 2;
    constexpr unsigned NP0 = HELP ? 8 : 4, NG = RS_GSPLIT ? 8 : 4;
    auto fh_target = [&](int u) -> unsigned { return NP0 + 4u * (unsigned)u; };      // F_H / F_A after unit u

    if (threadIdx.x < 32) reinterpret_cast<unsigned *>(sm + RS_SFLAG)[threadIdx.x] = 0;
    if (threadIdx.x < 96) reinterpret_cast<int *>(sm + RS_SMAX)[threadIdx.x] = (int)0x80000000;
    // ---- norm2 + qact3 (vit_quant.py:139-140) of the rows this workgroup will multiply (LnGroup<384, 2>: layernorm_reg_kernel's arithmetic,
    // 8 rows per wave and pass), into the 8-bit scratch p.x; the units fetch their activation tiles from it (L2) as before.  The LayerNorm's
    // constants sit in the ShiftGELU table-line slots, which nobody touches before the first ShiftGELU (LNH = 2: before F_L)
    typedef LnGroup<MLP_C, 2> LG;
    double *cC = reinterpret_cast<double *>(sm + RS_STAB);
    float *cB = reinterpret_cast<float *>(sm + RS_STAB + MLP_C * 8), *cSc = cB + MLP_C, *cY = cSc + MLP_C;
    static_assert(MLP_C * 20 <= 32 * 256, "the LayerNorm constants fit the table-line slots");
    bool ln_fast = false;
    // rows [r_beg, r_end) by `nw` waves, this one being number `w`
    auto ln_rows = [&](long long r_beg, long long r_end, int w, int nw) __attribute__((always_inline)) {
        const int lane = threadIdx.x & 63, j = lane & 7, k = j >> 1, hh = j & 1;
        const float ys = rcp_rn(p.ln_s);
        int8_t *a8 = const_cast<int8_t *>(p.x);
        for (long long r0 = r_beg + w * 8; r0 < r_end; r0 += nw * 8) {
            const long long row_raw = r0 + (lane >> 3);
            const bool live = row_raw < r_end;
            const long long row = live ? row_raw : r_end - 1;
            const int16_t *xp = p.residual + row * MLP_C + 8 * k + 4 * hh;
            float xv[LG::NSTEP][LG::EPC];
#pragma unroll
            for (int i = 0; i < LG::NSTEP; ++i) {
                const LnRaw<4>::T t = *reinterpret_cast<const LnRaw<4>::T *>(xp + 32 * i);
#pragma unroll
                for (int c = 0; c < 4; ++c) xv[i][c] = requotient_m((float)t[c], p.ln_s, ys);
            }
            LG::run(xv, j, k, 8 * k + 4 * hh, ln_fast, live, cC, cB, cSc, cY, p.ln_bias_int, p.ln_sc, p.ln_dy, a8 + row * MLP_C + 8 * k + 4 * hh);
        }
    };
    // unit range [u0, u1) of this workgroup: contiguous rows in the balanced schedule, one 64-token unit at a time otherwise
    auto ln_units = [&](int u0, int u1, int w, int nw) __attribute__((always_inline)) {
        if (u0 >= u1) return;
        if (p.balanced) {
            ln_rows(unit_tile0(u0) * 16, min(unit_tile0(u1) * 16, p.M), w, nw);
        } else {
            for (int u = u0; u < u1; ++u) ln_rows(unit_tile0(u) * 16, min((unit_tile0(u) + unit_ntt(u)) * 16, p.M), w, nw);
        }
    };
    if constexpr (LNH != 0) {
        ln_fast = ln_stage_constants<MLP_C, RS_THREADS>(p.ln_bias_int, p.ln_sc, p.ln_dy, cC, cB, cSc, cY);
        ln_units(0, LNH == 2 ? 1 : nu, wave, 8);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's bytes are in the L2 before anybody's DMA asks for them
    }
    __syncthreads();

    // activation tile of a unit: global -> LDS by DMA, 16 tokens x 4 chunk slots per instruction; the source chunk of a
    // slot is slot ^ g(token) (the permutation is applied on the source side; the LDS side of a DMA is lane-linear)
    auto a_dma = [&](long long tile0, int ntt) __attribute__((always_inline)) {
        const int lane = threadIdx.x & 63;
        for (int tg = 0; tg < ntt; ++tg) {
            const int tokl = tg * 16 + (lane >> 2), c = (lane & 3) ^ rs_g(tokl);
            const long long grow = min(tile0 * 16 + tokl, p.M - 1);
            const int8_t *src = p.x + grow * MLP_C + c * 16;
#pragma unroll
            for (int kb = 0; kb < MLP_KS1; ++kb) {
                const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(RS_SA + kb * RS_KBLK + tg * 1024));
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + kb * 64),
                                                 (__attribute__((address_space(3))) void *)(sm + dst), 16, 0, 0);
            }
        }
    };

    // ------------------------------------------------------------------------------------------------------------------------
    // fc1 + qact_gelu of unit u, iterations [it0, it1) of RS_HD rounds each, as producer `pw` (channels 128 r + 32 pw ... of
    // round r).  wf: this wave's weight ring, holding the first RS_WR - 1 k-steps of iteration it0 on entry; on exit it holds
    // the first k-steps of iteration 0 (the next unit's).  Ends with the row maxima folded into RS_SMAX and F_H signalled.
    auto produce = [&](auto nt_c, const int pw, v4i(&wf)[RS_WR], const int u, const int it0, const int it1, const long long next_tile0,
                       const int next_ntt, const bool loader) __attribute__((always_inline)) {
        constexpr int NT = decltype(nt_c)::value;            // 32-token tiles multiplied: 2, or 3 for a five-tile unit
        constexpr int SPI = RS_HD * 12;                      // k-steps per iteration of the round loop
        static_assert(SPI % RS_WR == 0 && 12 % RS_HD == 0, "ring slots must be static across iterations");
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, tok = lane & 31, kh = lane >> 5, e = kh ^ rs_g(tok);
        // LDS addresses as integers (per-lane base + immediates): through `sm + ...` every access costs an address register
        const unsigned fa0 = sm_lds + RS_SA + tok * 64 + e * 16, fa1 = sm_lds + RS_SA + tok * 64 + (e ^ 2) * 16;
        const unsigned hwo = sm_lds + RS_SH + (pw >> 1) * RS_KBLK + tok * 64 + ((((pw & 1) * 2) ^ e) * 16);
        const unsigned dummy = sm_lds + RS_SDUMMY + lane * 16;
        const v4i *w1 = p.w1f + pw * 64 + lane;
        v4i bf[2][NT], hold[RS_HD][NT];
        int mx[NT];                                    // running maximum of this lane's Q + 128 per token tile, before the clamp
#pragma unroll
        for (int t = 0; t < NT; ++t) mx[t] = 0;
        auto load_b = [&](int ks, int slot) __attribute__((always_inline)) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
                bf[slot][t] = *(lds_v4i *)(size_t)(((ks & 1) ? fa1 : fa0) + (ks >> 1) * RS_KBLK + t * 2048);
        };
        auto flush = [&](int slot, int round) __attribute__((always_inline)) {
            rs_wait(fl + 4 * (RS_F_R + round), 4u * (unsigned)u);
            const unsigned hb = 2 * round * RS_KBLK + hwo;
#pragma unroll
            for (int t = 0; t < NT; ++t) {          // rows 80..95 of a five-tile unit do not exist: those lanes write a dummy slot
                if (t < 2) *(lds_v4i *)(size_t)(hb + t * 2048) = hold[slot][t];
                else *(lds_v4i *)(size_t)(tok < 16 ? hb + t * 2048 : dummy) = hold[slot][t];
            }
        };
        v16i bias;
        auto load_bias = [&](int r) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const v4i b4 = *reinterpret_cast<const v4i *>(p.b1 + 128 * r + 32 * pw + 16 * kh + 4 * q);
                bias[4 * q] = b4[0]; bias[4 * q + 1] = b4[1]; bias[4 * q + 2] = b4[2]; bias[4 * q + 3] = b4[3];
            }
        };
        load_bias(it0 * RS_HD);
        rs_wait(fl + 4 * RS_F_D, (unsigned)u + 1);
        stamp(u, 0);
        load_b(0, 0);
        for (int it = it0; it < it1; ++it) {
            const v4i *wq = w1 + (size_t)it * SPI * 256;
            const v4i *wn = (it + 1 < it1) ? wq + (size_t)SPI * 256 : w1;     // the k-steps behind this iteration's
#pragma unroll
            for (int rr = 0; rr < RS_HD; ++rr) {
                const int r = it * RS_HD + rr, chb = 128 * r + 32 * pw + 16 * kh;
                v16i acc[NT];
                v2d cq[8];                             // this round's multipliers: requested four k-steps before the requant
#pragma unroll
                for (int ks = 0; ks < 12; ++ks) {
                    const int s = rr * 12 + ks, nx = s + RS_WR - 1;
                    __builtin_amdgcn_sched_barrier(0);
                    wf[nx % RS_WR] = nx < SPI ? wq[(size_t)nx * 256] : wn[(size_t)(nx - SPI) * 256];
                    load_b((ks + 1) % 12, (ks + 1) & 1);
                    if (ks == 8) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) cq[i] = *reinterpret_cast<const v2d *>(p.cq1 + chb + 2 * i);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[s % RS_WR], bf[ks & 1][t], ks == 0 ? bias : acc[t], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (it > it0) flush(rr, r - RS_HD);
                if (r + 1 < 12) load_bias(r + 1);       // consumed by the next round's first MFMAs, behind this requant
                // requant: accumulator register v is channel chb + v.  fma(z, c, magic + 128) leaves Q + 128 in the low dword;
                // v_cvt_pk_i16_i32 and v_sat_pk_u8_i16 saturate to [0, 255] = clamp(Q, -128, 127) + 128 while packing: the hidden
                // tile holds BIASED bytes, which is what ShiftGELU's table is indexed by
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        int o[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const double c = cq[2 * q + (i >> 1)][i & 1];
                            const double tq = FMA ? __builtin_fma((double)acc[t][4 * q + i], c, MLP_MAGIC + 128.0)
                                                  : ((double)acc[t][4 * q + i] * c + (MLP_MAGIC + 128.0));
                            o[i] = (RS_ABL & 4) ? acc[t][4 * q + i] : __double2loint(tq);
                        }
                        mx[t] = max(max(mx[t], o[0]), o[1]);
                        mx[t] = max(max(mx[t], o[2]), o[3]);
                        asm volatile("" : "+v"(mx[t]));
                        unsigned p01, p23, b01, b23;
                        asm("v_cvt_pk_i16_i32 %0, %1, %2" : "=v"(p01) : "v"(o[0]), "v"(o[1]));
                        asm("v_cvt_pk_i16_i32 %0, %1, %2" : "=v"(p23) : "v"(o[2]), "v"(o[3]));
                        asm("v_sat_pk_u8_i16 %0, %1" : "=v"(b01) : "v"(p01));
                        asm("v_sat_pk_u8_i16 %0, %1" : "=v"(b23) : "v"(p23));
                        int hq = (int)__builtin_amdgcn_perm(b23, b01, 0x05040100u);
                        asm volatile("" : "+v"(hq));       // pinned here: left alone, the optimiser sinks the whole requant
                        hold[rr][t][q] = hq;               // of all RS_HD rounds to the flush that first reads it (and spills)
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        stamp(u, 1);
        // the activation tile is dead for this wave (the look-ahead read of k-step 0 above is discarded)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        rs_signal(fl + 4 * RS_F_A);
        if (loader && next_ntt > 0) {
            rs_wait(fl + 4 * RS_F_A, fh_target(u));
            if (LNH == 2) rs_wait(fl + 4 * RS_F_L, 4u);      // the consumers' LayerNorm of the later units' rows is in the L2
            a_dma(next_tile0, next_ntt);
        }
#pragma unroll
        for (int rr = 0; rr < RS_HD; ++rr) flush(rr, (it1 - 1) * RS_HD + rr);
        // ShiftGELU's row maximum (quant_modules.py:420-424 via the table's row index): folded across the lanes and the
        // producers by ds_max; the previous unit's maxima were read and reset before F_R[11] let us by
#pragma unroll
        for (int t = 0; t < NT; ++t)
            asm volatile("ds_max_i32 %0, %1" ::"v"(sm_lds + RS_SMAX + (t * 32 + tok) * 4), "v"(min(mx[t], 255)) : "memory");
        rs_signal(fl + 4 * RS_F_H);
        if (loader && next_ntt > 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            rs_signal(fl + 4 * RS_F_D);
        }
        stamp(u, 2);
    };

    // ------------------------------------------------------------------------------------------------------------------------
    // ShiftGELU (+ qact1) of a unit in place, half a wavefront per token: half-wave `hwid` of NHW takes tokens hwid + NHW i.
    // The row maximum comes from the producers (RS_SMAX, reset here); per token: its 256-byte table line global -> one of this
    // half-wave's two LDS slots, the row's 12 dwords per lane, 48 byte gathers, write-back.  Gathers are issued three dwords
    // (12 gathers) ahead of the merge that consumes them (lgkmcnt counts 15 at most), their addresses are one SDWA each.
    auto gelu = [&](auto nt_c, auto nhw_c, const int nvalid) __attribute__((always_inline)) {
        constexpr int NT = decltype(nt_c)::value, NHW = decltype(nhw_c)::value;
        constexpr int NTK = (NT == 3 ? 80 : 64) / NHW;                                   // tokens per half-wave
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, l32 = lane & 31;
        const int hwid = (NHW == 16 ? wave * 2 : (wave - 4) * 2) + (lane >> 5);
        const unsigned rowa = sm_lds + RS_SH + hwid * 64 + (l32 & 15) * 4 + (l32 >> 4) * RS_KBLK;    // token hwid, K blocks l32 >> 4 (+ 2 m)
        const unsigned rowb = rowa + 12 * RS_KBLK;
        const unsigned maxa = sm_lds + RS_SMAX + hwid * 4;
        v2i line[NTK];
#pragma unroll
        for (int i = 0; i < NTK; ++i)
            if (i * NHW < nvalid && !(RS_ABL & 1)) {     // tokens NHW i .. NHW i + NHW - 1, one per half-wave: all valid or none
                const int qb = *(lds_i32 *)(size_t)(maxa + i * NHW * 4);          // max(Q) + 128
                line[i] = reinterpret_cast<const v2i *>(p.tab + (size_t)qb * 256)[l32];
            }
#pragma unroll
        for (int i = 0; i < NTK; ++i)
            if (i * NHW < nvalid && !(RS_ABL & 1)) {
                if (l32 == 0) *(lds_i32 *)(size_t)(maxa + i * NHW * 4) = (int)0x80000000;
                const unsigned base = sm_lds + RS_STAB + (hwid * 2 + (i & 1)) * 256;
                *(lds_v2i *)(size_t)(base + l32 * 8) = line[i];
                unsigned w[12], g[12][4];
#pragma unroll
                for (int m = 0; m < 12; ++m) w[m] = *(lds_u32 *)(size_t)((m < 6 ? rowa : rowb) + i * NHW * 64 + (m % 6) * 2 * RS_KBLK);
#pragma unroll
                for (int c = 0; c <= 4; ++c) {
                    if (c < 4) {
#pragma unroll
                        for (int m = 3 * c; m < 3 * c + 3; ++m) {
                            unsigned a0, a1, a2, a3;
                            asm("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "=v"(a0) : "v"(w[m]), "v"(base));
                            asm("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(a1) : "v"(w[m]), "v"(base));
                            asm("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(a2) : "v"(w[m]), "v"(base));
                            asm("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(a3) : "v"(w[m]), "v"(base));
                            asm volatile("ds_read_u8 %0, %1" : "=v"(g[m][0]) : "v"(a0) : "memory");
                            asm volatile("ds_read_u8 %0, %1" : "=v"(g[m][1]) : "v"(a1) : "memory");
                            asm volatile("ds_read_u8_d16_hi %0, %1" : "=v"(g[m][2]) : "v"(a2) : "memory");     // byte << 16, low half zeroed (SRAM-ECC d16 semantics)
                            asm volatile("ds_read_u8_d16_hi %0, %1" : "=v"(g[m][3]) : "v"(a3) : "memory");
                        }
                    }
                    if (c > 0) {
                        if (c < 4) asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");
                        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                        for (int m = 3 * c - 3; m < 3 * c; ++m) {
                            unsigned o, t13;
                            asm volatile("v_or_b32 %0, %1, %2" : "=v"(t13) : "v"(g[m][1]), "v"(g[m][3]));      // behind the wait
                            asm volatile("v_or3_b32 %0, %1, %2, %3" : "=v"(o) : "v"(g[m][0]), "v"(g[m][2]), "v"(t13 << 8));
                            *(lds_u32 *)(size_t)((m < 6 ? rowa : rowb) + i * NHW * 64 + (m % 6) * 2 * RS_KBLK) = o;
                        }
                    }
                }
            }
        // the producers fold maxima into all 32 NT rows of the unit, the rows >= nvalid from stale LDS; the sweep above only reads
        // and resets the valid ones.  Reset the rest of this half-wave's slots too, so that a short unit could never hand stale
        // maxima to a larger unit after it (the host only makes short LAST units today)
#pragma unroll
        for (int i = 0; i < NTK; ++i)
            if (i * NHW >= nvalid && l32 == 0) *(lds_i32 *)(size_t)(maxa + i * NHW * 4) = (int)0x80000000;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        rs_signal(fl + 4 * RS_F_G);
    };

    if (wave < 4) {
        // =========================================================================================== producers
        if (wave == 0) {
            a_dma(unit_tile0(0), unit_ntt(0));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            rs_signal(fl + 4 * RS_F_D);
        }
        v4i wf[RS_WR];
        {
            const v4i *w1 = p.w1f + wave * 64 + (threadIdx.x & 63);
#pragma unroll
            for (int s = 0; s < RS_WR - 1; ++s) wf[s] = w1[(size_t)s * 256];
        }
        for (int u = 0; u < nu; ++u) {
            const int ntt = unit_ntt(u), it1 = (HELP && u == 0) ? 12 / RS_HD / 2 : 12 / RS_HD;
            if (!(RS_DBG_ROLE & 1)) continue;
            if (ntt == MLP_TT && (RS_DBG_NT & 2)) {
                produce(std::integral_constant<int, 3>{}, wave, wf, u, 0, it1, unit_tile0(u + 1), unit_ntt(u + 1), wave == 0);
                if (RS_GSPLIT) { rs_wait(fl + 4 * RS_F_H, fh_target(u)); if (LNH == 2) rs_wait(fl + 4 * RS_F_L, 4u); gelu(std::integral_constant<int, 3>{}, std::integral_constant<int, 16>{}, ntt * 16); }
            } else if (RS_DBG_NT & 1) {
                produce(std::integral_constant<int, 2>{}, wave, wf, u, 0, it1, unit_tile0(u + 1), unit_ntt(u + 1), wave == 0);
                if (RS_GSPLIT) { rs_wait(fl + 4 * RS_F_H, fh_target(u)); if (LNH == 2) rs_wait(fl + 4 * RS_F_L, 4u); gelu(std::integral_constant<int, 2>{}, std::integral_constant<int, 16>{}, ntt * 16); }
            }
        }
    } else {
        // =========================================================================================== consumers
        const int j = wave - 4;
        if constexpr (LNH == 2) {
            ln_units(1, nu, j, 4);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            rs_signal(fl + 4 * RS_F_L);
            rs_wait(fl + 4 * RS_F_L, 4u);                    // every consumer is done with the constants in the table-line slots
        }
        if (HELP && (RS_DBG_ROLE & 1)) {
            // the first unit's fc1, second half of the rounds: nothing else for a consumer to do until a hidden tile exists
            static_assert(!RS_HELP || 12 / RS_HD == 2, "the first unit is split by iterations");
            v4i wfh[RS_WR];
            const v4i *w1 = p.w1f + (size_t)RS_HD * 12 * 256 + j * 64 + (threadIdx.x & 63);
#pragma unroll
            for (int s = 0; s < RS_WR - 1; ++s) wfh[s] = w1[(size_t)s * 256];
            if (unit_ntt(0) == MLP_TT && (RS_DBG_NT & 2)) produce(std::integral_constant<int, 3>{}, j, wfh, 0, 1, 2, 0, 0, false);
            else if (RS_DBG_NT & 1) produce(std::integral_constant<int, 2>{}, j, wfh, 0, 1, 2, 0, 0, false);
        }
        auto c_unit = [&](auto nt_c, const int u, const long long tile0, const int ntt) __attribute__((always_inline)) {
            constexpr int NT = decltype(nt_c)::value;
            int tid = threadIdx.x;
            asm volatile("" : "+v"(tid));
            const int lane = tid & 63, tok = lane & 31, kh = lane >> 5, e = kh ^ rs_g(tok);
            const unsigned fh0 = sm_lds + RS_SH + tok * 64 + e * 16, fh1 = sm_lds + RS_SH + tok * 64 + (e ^ 2) * 16;
            const unsigned fh2 = fh0 + 12 * RS_KBLK, fh3 = fh1 + 12 * RS_KBLK;       // the DS offset field holds 16 bits
            const int nvalid = ntt * 16;
            const v4i *w2 = p.w2f + (size_t)(j * 3) * 64 + lane;
            v4i wf[RS_WD2 + 1][3], bf[2][NT];
            auto load_w = [&](int ks, int slot) __attribute__((always_inline)) {
#pragma unroll
                for (int ct = 0; ct < 3; ++ct) wf[slot][ct] = w2[(size_t)(ks * 12 + ct) * 64];
            };
            auto load_b = [&](int ks, int slot) __attribute__((always_inline)) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    bf[slot][t] = *(lds_v4i *)(size_t)((ks < 24 ? ((ks & 1) ? fh1 : fh0) : ((ks & 1) ? fh3 : fh2)) + ((ks >> 1) % 12) * RS_KBLK + t * 2048);
            };
#pragma unroll
            for (int s = 0; s < RS_WD2; ++s) load_w(s, s);
            rs_wait(fl + 4 * RS_F_H, fh_target(u));
            stamp(u, 3);
            if (RS_GSPLIT) gelu(nt_c, std::integral_constant<int, 16>{}, nvalid);
            else gelu(nt_c, std::integral_constant<int, 8>{}, nvalid);
            rs_wait(fl + 4 * RS_F_G, NG * (unsigned)(u + 1));
            stamp(u, 4);
            // ---- fc2: output channels 96 j + 32 ct + 16 kh + v of token t * 32 + tok in acc[ct][t][v]
            v16i acc[3][NT];
            {
                v16i bias[3];
#pragma unroll
                for (int ct = 0; ct < 3; ++ct)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const v4i b4 = *reinterpret_cast<const v4i *>(p.b2 + 96 * j + 32 * ct + 16 * kh + 4 * q);
                        bias[ct][4 * q] = b4[0]; bias[ct][4 * q + 1] = b4[1]; bias[ct][4 * q + 2] = b4[2]; bias[ct][4 * q + 3] = b4[3];
                    }
                load_b(0, 0);
#pragma unroll
                for (int ks = 0; ks < 48; ++ks) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (ks + RS_WD2 < 48) load_w(ks + RS_WD2, (ks + RS_WD2) % (RS_WD2 + 1));
                    if (ks + 1 < 48) load_b(ks + 1, (ks + 1) & 1);
                    if ((ks & 3) == 3) rs_signal(fl + 4 * (RS_F_R + (ks >> 2)));     // every read of slice ks >> 2 is issued
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int ct = 0; ct < 3; ++ct)
#pragma unroll
                        for (int t = 0; t < NT; ++t)
                            acc[ct][t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[ks % (RS_WD2 + 1)][ct], bf[ks & 1][t],
                                                                                ks == 0 ? bias[ct] : acc[ct][t], 0, 0, 0);
                }
            }
            stamp(u, 5);
            // ---- qact2 (16 bit) + qact4 with the identity branch: 16 consecutive channels of a token per lane.  The identity rows
            // of tile n + 2 and the multipliers of the next channel tile are requested while tile n is requantised
            const long long tok0 = tile0 * 16;
            v4i idr[3][2];
            v2d c2[2][8];
            auto load_id = [&](int n) __attribute__((always_inline)) {
                const int ct = n / NT, t = n - ct * NT;
                const long long row = min(tok0 + t * 32 + tok, p.M - 1);
                const int16_t *rp = p.residual + row * MLP_C + 96 * j + 32 * ct + 16 * kh;
                idr[n % 3][0] = *reinterpret_cast<const v4i *>(rp);
                idr[n % 3][1] = *reinterpret_cast<const v4i *>(rp + 8);
            };
            auto load_c2 = [&](int ct) __attribute__((always_inline)) {
#pragma unroll
                for (int i = 0; i < 8; ++i) c2[ct & 1][i] = *reinterpret_cast<const v2d *>(p.cq2 + 96 * j + 32 * ct + 16 * kh + 2 * i);
            };
            load_c2(0);
            load_id(0);
            load_id(1);
#pragma unroll
            for (int n = 0; n < 3 * NT; ++n) {
                const int ct = n / NT, t = n - ct * NT, ch0 = 96 * j + 32 * ct + 16 * kh;
                __builtin_amdgcn_sched_barrier(0);
                if (n + 2 < 3 * NT) load_id(n + 2);
                if (t == 0 && ct + 1 < 3) load_c2(ct + 1);
                __builtin_amdgcn_sched_barrier(0);
                const int tl = t * 32 + tok;
                const long long row = tok0 + tl;
                v4i o0, o1;
#pragma unroll
                for (int d = 0; d < 8; ++d) {
                    const unsigned rw = (unsigned)(d < 4 ? idr[n % 3][0][d] : idr[n % 3][1][d - 4]);
                    int o[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int v = 2 * d + h;
                        if (RS_ABL & 2) { o[h] = acc[ct][t][v] ^ (int)rw ^ (int)c2[ct & 1][v >> 1][v & 1]; continue; }
                        const int t16 = min(max(mlp_rq<FMA>(acc[ct][t][v], c2[ct & 1][v >> 1][v & 1]), -32768), 32767);
                        const int r = h ? ((int)rw >> 16) : (int)(short)(rw & 0xffffu);
                        o[h] = rq_fast(r, p.cr) + rq_fast(t16, p.cm);         // both terms < 2^31 / 2: the sum is the reference's fp64 sum
                    }
                    int pk;
                    asm("v_cvt_pk_i16_i32 %0, %1, %2" : "=v"(pk) : "v"(o[0]), "v"(o[1]));      // clamp to 16 bits and pack in one
                    if (d < 4) o0[d] = pk; else o1[d - 4] = pk;
                }
                if (tl < nvalid && row < p.M) {
                    *reinterpret_cast<v4i *>(p.out + row * MLP_C + ch0) = o0;
                    *reinterpret_cast<v4i *>(p.out + row * MLP_C + ch0 + 8) = o1;
                }
            }
            stamp(u, 6);
        };
        for (int u = 0; u < nu; ++u) {
            const int ntt = unit_ntt(u);
            if (!(RS_DBG_ROLE & 2)) continue;
            if (ntt == MLP_TT && (RS_DBG_NT & 2)) c_unit(std::integral_constant<int, 3>{}, u, unit_tile0(u), ntt);
            else if (RS_DBG_NT & 1) c_unit(std::integral_constant<int, 2>{}, u, unit_tile0(u), ntt);
        }
    }
}
