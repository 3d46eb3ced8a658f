// ivit_layernorm_plan.h — EXPERIMENT (round 6), not part of the library: the register-resident I-LayerNorm with the per-channel
// constants PRECOMPUTED (a "LayerNorm plan") and staged by LDS-DMA behind the row loads, waited for only in front of the
// output pass.  Hypothesis: the one-shot kernel's staging (global loads -> an fp64 division per channel -> LDS -> barrier) sits
// in front of its first row request and is a large part of the ~7 us a launch spends beside its marginal rate.  Measured
// (tools/ubench/ln_plan_probe.hip, 50 432 x 384, byte-identical): 19.6 -> 19.1 us (256 rows: 5.3 -> 4.8) — the staging is worth
// 0.5 us, not the 7; 78 registers (6 waves per SIMD) instead of 82 (5) changed nothing either.  Not worth a plan object in the
// C-ABI; the head and tail of the launch are the lock-step start of all waves and the second round of blocks
// (profiles/README.md, round 6).
#pragma once
#include "../../i-vit_amd/csrc/ivit_layernorm.h"

// ---- planned form (ivit_layernorm_plan_create): the per-channel constants come PRECOMPUTED as one blob in the layout of the
// block's LDS copy (c double[CC] | bias float[CC] | sc float[CC] | 1/sc float[CC] = 20 CC bytes) and travel by LDS-DMA, issued
// before the row loads and waited for only in front of the output pass: the staging round trip of the one-shot kernel (global
// loads -> an fp64 division per channel -> LDS -> barrier, ~1.5 us before the first row is even requested) leaves the
// head of the launch.  `fast` is the plan's verdict on the two-operation 8-bit requant (every channel inside the bound).
__global__ __launch_bounds__(256) void layernorm_plan_build_kernel(const float *__restrict__ bias_int, const float *__restrict__ sc,
                                                                   const ivit_dyadic *__restrict__ dy, int C, char *__restrict__ blob,
                                                                   int *__restrict__ wide_flag) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float scv = sc[c], bv = bias_int[c];
    const double cv = dy[c].m * dy[c].r;
    reinterpret_cast<double *>(blob)[c] = cv;
    reinterpret_cast<float *>(blob + 8 * (size_t)C)[c] = bv;
    reinterpret_cast<float *>(blob + 12 * (size_t)C)[c] = scv;
    reinterpret_cast<float *>(blob + 16 * (size_t)C)[c] = rcp_rn(scv);
    if (!(fabs(cv) * (1.2e12 + 1.01 * fabs((double)bv)) < 2147483000.0)) atomicOr(wide_flag, 1);
}

template <int CC, int S>
__global__ __launch_bounds__(LNR_THREADS(S), LNR_MIN_WAVES(CC, S)) void layernorm_plan_kernel(const int16_t *__restrict__ x, long long rows,
                                                                      long long row_stride, float s,
                                                                      const char *__restrict__ blob, int fast,
                                                                      int8_t *__restrict__ out) {
    static_assert(S != 1, "the 4-lanes-per-row form is a probe");
    typedef LnGroup<CC, S> G;
    constexpr int LPR = G::LPR, EPC = G::EPC, NSTEP = G::NSTEP, RPW = G::RPW, RPB = (LNR_THREADS(S) / 64) * RPW;
    constexpr int THREADS = LNR_THREADS(S), NCH = 20 * CC / 16;            // 16-byte pieces of the blob
    __shared__ __attribute__((aligned(16))) char cst[20 * CC];
    const int tid = threadIdx.x;
#pragma unroll
    for (int c0 = 0; c0 < NCH; c0 += THREADS) {
        const int c = c0 + tid;
        const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)((c0 + (tid & ~63)) * 16));     // this wave's 1 KB window
        if (c < NCH)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(blob + (size_t)c * 16),
                                             (__attribute__((address_space(3))) void *)(cst + base), 16, 0, 0);
    }
    const int lane = tid & 63, j = lane % LPR, k = j / S, hh = j % S;
    const long long row_raw = (long long)blockIdx.x * RPB + (tid >> 6) * RPW + lane / LPR;
    const bool live = row_raw < rows;
    const long long row = live ? row_raw : rows - 1;          // a dead lane group recomputes the last row, stores nothing
    const int16_t *xp = x + row * row_stride + 8 * k + EPC * hh;
    const float ys = rcp_rn(s);
    float xv[NSTEP][EPC];
#pragma unroll
    for (int i = 0; i < NSTEP; ++i) {
        const typename LnRaw<EPC>::T t = *reinterpret_cast<const typename LnRaw<EPC>::T *>(xp + 32 * i);
#pragma unroll
        for (int e = 0; e < EPC; ++e) xv[i][e] = requotient_m((float)t[e], s, ys);
    }
    G::run(xv, j, k, 8 * k + EPC * hh, fast != 0, live, reinterpret_cast<const double *>(cst), reinterpret_cast<const float *>(cst + 8 * CC),
           reinterpret_cast<const float *>(cst + 12 * CC), reinterpret_cast<const float *>(cst + 16 * CC), nullptr, nullptr, nullptr,
           out + row * CC + 8 * k + EPC * hh, std::true_type{});
}

