// ivit_layernorm_pipe.h — EXPERIMENT (round 6), not part of the library: a persistent, software-pipelined form of the
// register-resident I-LayerNorm (i-vit_amd/csrc/ivit_layernorm.h).  Built to test the hypothesis "HBM time and VALU time add
// because every wave loads, then every wave computes"; measured with tools/ubench/ln_pipe_probe.hip, byte-identical to the
// shipped kernel at every grid size and SLOWER: 50 432 x 384 — one-shot 19.5 us; persistent 24.9 (1 block / CU), 21.1 (2),
// 21.4 (3), 20.5 (4); with hipcc's own loop-top vmcnt(0) instead of the counted wait 26.8 / 21.5 / 21.6 / 20.5.  What the
// row-count sweep of the one-shot kernel shows instead (profiles/README.md, round 6): its marginal rate is 12.4 us per
// 50 432 rows (= the VALU issue time of ~1 250 wave-instructions per 8-row group), the other ~7 us are the head (constants,
// first rows: 5.3 us for ONE block) and a second round of 296 blocks at low occupancy; waves that start together stay in
// phase through the latency-bound sections (Newton iteration, cross-lane sums), which a persistent grid makes worse, not better.
#pragma once
#include "../../i-vit_amd/csrc/ivit_layernorm.h"

// Pipelined form for whole activation tensors.  The one-shot kernel above holds a row group per wave and the grid is ~1.2
// rounds of resident blocks at DeiT-S b256 (1 576 blocks on 1 280 slots): every wave of the chip loads at once, then every wave
// computes, then a thin second round — HBM time and VALU time ADD (measured 20 us = ~9 + ~9 + tail).  Here the grid is
// LNP_BPC blocks per CU, each wave walks the row groups g, g + W, g + 2 W, ... and requests group g + W (as raw int16, 6 / S
// registers per step) before it starts the arithmetic of group g, so that in the steady state a SIMD's waves compute while
// the next rows are in flight; the per-channel constants are staged once per block instead of once per 32 rows.
#ifndef IVIT_OPT_LN_PIPE
#define IVIT_OPT_LN_PIPE 1
#endif
#ifndef LNP_BPC
#define LNP_BPC 2
#endif
#ifndef LNP_MINW
#define LNP_MINW 2
#endif
template <int CC, int S>
__global__ __launch_bounds__(256, LNP_MINW) void layernorm_pipe_kernel(const int16_t *__restrict__ x, long long rows,
                                                                      long long row_stride, float s,
                                                                      const float *__restrict__ bias_int,
                                                                      const float *__restrict__ sc,
                                                                      const ivit_dyadic *__restrict__ dy,
                                                                      int8_t *__restrict__ out) {
    static_assert(S != 1, "the 4-lanes-per-row form is a probe");
    typedef LnGroup<CC, S> G;
    typedef typename LnRaw<G::EPC>::T raw_t;
    constexpr int LPR = G::LPR, EPC = G::EPC, NSTEP = G::NSTEP, RPW = G::RPW;
    __shared__ __attribute__((aligned(16))) double cC[CC];
    __shared__ __attribute__((aligned(16))) float cB[CC], cSc[CC], cY[CC];
    const int tid = threadIdx.x;
    const bool fastrq = ln_stage_constants<CC, 256>(bias_int, sc, dy, cC, cB, cSc, cY);
    const int lane = tid & 63, j = lane % LPR, k = j / S, hh = j % S, cb0 = 8 * k + EPC * hh, rl = lane / LPR;
    const float ys = rcp_rn(s);
    const long long ngroups = (rows + RPW - 1) / RPW, gstride = (long long)gridDim.x * 4;
    long long g = (long long)blockIdx.x * 4 + (tid >> 6);
    if (g >= ngroups) return;                                 // wave-uniform; no barrier below
    auto row_of = [&](long long gg) { const long long r = gg * RPW + rl; return r < rows ? r : rows - 1; };
    // The row loads are asm statements the compiler does not count, with the waits written here: left to hipcc, the loop-top
    // wait for the prefetched rows is vmcnt(0) — it drains the previous group's twelve stores as well (the counter retires in
    // order and the stores are younger), which puts a store round trip on every group's critical path.
    raw_t raw[NSTEP];
    auto request = [&](long long gg) {
        const int16_t *xp = x + row_of(gg) * row_stride + cb0;
#pragma unroll
        for (int i = 0; i < NSTEP; ++i) {
            if constexpr (EPC == 8) asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(raw[i]) : "v"(xp), "i"(64 * i) : "memory");
            else if constexpr (EPC == 4) asm volatile("global_load_dwordx2 %0, %1, off offset:%2" : "=v"(raw[i]) : "v"(xp), "i"(64 * i) : "memory");
            else asm volatile("global_load_dword %0, %1, off offset:%2" : "=v"(raw[i]) : "v"(xp), "i"(64 * i) : "memory");
        }
    };
    static_assert(64 * (NSTEP - 1) < 4096, "13-bit signed offset of global_load");
    request(g);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (; g < ngroups; g += gstride) {
        // everything older than this group's NSTEP stores has landed: the rows requested one group ago
        static_assert(NSTEP == 3 || NSTEP == 4 || NSTEP == 6 || NSTEP == 8 || NSTEP == 12 || NSTEP == 16 || NSTEP == 24 || NSTEP == 32 || NSTEP == 48, "wait ladder");
#pragma unroll
        for (int i = 0; i < NSTEP; ++i) {
            if (i == 0) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(raw[0]) : "i"(NSTEP < 63 ? NSTEP : 63));
            else asm volatile("" : "+v"(raw[i]));
        }
        float xv[NSTEP][EPC];
#pragma unroll
        for (int i = 0; i < NSTEP; ++i)
#pragma unroll
            for (int e = 0; e < EPC; ++e) xv[i][e] = requotient_m((float)raw[i][e], s, ys);
        if (g + gstride < ngroups) request(g + gstride);        // the next group of this wave, in flight under this group's arithmetic
        // a dead lane group (past the last row) recomputes the last row and stores the same bytes again: every wave issues
        // exactly NSTEP stores per group, so the wait for the next group's loads is a constant vmcnt(NSTEP)
        G::run(xv, j, k, cb0, fastrq, true, cC, cB, cSc, cY, bias_int, sc, dy, out + row_of(g) * CC + cb0);
    }
}

