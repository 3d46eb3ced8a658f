// attn_probe.hip — the fused attention kernels alone (csrc/ivit_attention.h; tools/experiments/ivit_attention_stream.h) as a small shared library, so that a
// kernel edit is a 10-second rebuild instead of the whole libivit_hip.so: tools/attn_probe.py feeds it the same operands as the
// library and compares the two outputs byte for byte.
// variant: 0 = two-level tables (LUT = 1), 1 = row tables (LUT = 2), 2 = row tables, packed-byte streaming kernel (T = 577 only);
// ldv = 0: v row-major.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Xclang -target-feature -Xclang -packed-fp32-ops -fPIC -shared \
//        tools/ubench/attn_probe.hip -o tools/ubench/libattn_probe.so
#include "../experiments/ivit_attention_stream.h"
#include <stdio.h>

template <typename K>
static void launch_k(K kern, const AttnArgs &a, int BH, int threads, size_t lds) {
    (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    kern<<<BH, threads, lds, 0>>>(a);
}

extern "C" int attn_probe(const int8_t *q, const int8_t *k, const int8_t *vt, double qk_m, double qk_r, float s_softmax,
                          const uint16_t *aq, const float *et, const uint8_t *cls, int nc, int t_count, int dmin,
                          const float *rowtab, double pv_m, double pv_r, int8_t *ctx, int B, int H, int T, int ldv, int variant, int reps,
                          float *us) {
    AttnArgs a;
    a.q = q; a.k = k; a.vt = vt; a.ctx = ctx; a.T = T; a.H = H; a.ldv = ldv;
    a.s_softmax = s_softmax; a.dy_qk = ivit_dyadic{qk_m, qk_r}; a.dy_pv = ivit_dyadic{pv_m, pv_r};
    a.aq = aq; a.et = et; a.cls = cls; a.nc = nc; a.t_count = t_count; a.dmin = dmin; a.rowtab = rowtab;
    if (T != 197 && T != 577) return 1;
    if (variant && !rowtab) return 5;
    const size_t lut1 = (size_t)((t_count + 3) & ~3) * 4 + (size_t)nc * 512 + 256;
    auto launch = [&]() {
        const int BH = B * H;
        if (T == 197) {
            if (variant == 0) launch_k(attn_fused_kernel<4, true, 197, 1, false>, a, BH, ATT_WAVES * 64, AttCfg<4>::SMEM + lut1);
            else if (ldv == 0) launch_k(attn_fused_kernel<4, true, 197, 2, true>, a, BH, ATT_WAVES * 64, AttCfg<4>::SMEM + ATT_ROWLINE_BYTES);
            else launch_k(attn_fused_kernel<4, true, 197, 2, false>, a, BH, ATT_WAVES * 64, AttCfg<4>::SMEM + ATT_ROWLINE_BYTES);
        } else {
            if (variant == 0) launch_k(attn_fused_kernel<10, true, 577, 1, false>, a, BH, ATT_WAVES * 64, AttCfg<10>::SMEM + lut1);
            else if (variant == 1 && ldv == 0) launch_k(attn_fused_kernel<10, true, 577, 2, true>, a, BH, ATT_WAVES * 64, AttCfg<10>::SMEM + ATT_ROWLINE_BYTES);
            else if (variant == 1) launch_k(attn_fused_kernel<10, true, 577, 2, false>, a, BH, ATT_WAVES * 64, AttCfg<10>::SMEM + ATT_ROWLINE_BYTES);
            else if (ldv == 0) launch_k(attn_stream_kernel<10, 577, true>, a, BH, ATS_WAVES * 64, AtsCfg<10>::SMEM);
            else launch_k(attn_stream_kernel<10, 577, false>, a, BH, ATS_WAVES * 64, AtsCfg<10>::SMEM);
        }
    };
    launch();
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        (void)hipEventRecord(e0, 0);
        for (int i = 0; i < reps; ++i) launch();
        (void)hipEventRecord(e1, 0);
        if (hipEventSynchronize(e1) != hipSuccess) return 3;
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    *us = best * 1000.f / reps;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return hipGetLastError() == hipSuccess ? 0 : 4;
}
