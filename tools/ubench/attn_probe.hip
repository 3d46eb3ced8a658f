// attn_probe.hip — the fused attention kernel alone (ivit_attention.h) as a small shared library, so that a kernel edit is a
// 10-second rebuild instead of the whole libivit_hip.so: tools/attn_probe.py feeds it the same operands as the library and
// compares the two outputs byte for byte.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Xclang -target-feature -Xclang -packed-fp32-ops -fPIC -shared \
//        tools/ubench/attn_probe.hip -o tools/ubench/libattn_probe.so
#include "../../i-vit_amd/csrc/ivit_attention.h"
#include <stdio.h>

extern "C" int attn_probe(const int8_t *q, const int8_t *k, const int8_t *vt, double qk_m, double qk_r, float s_softmax,
                          const uint16_t *aq, const float *et, const uint8_t *cls, int nc, int t_count, int dmin,
                          const float *rowtab, double pv_m, double pv_r, int8_t *ctx, int B, int H, int T, int ldv, int reps, float *us) {
    AttnArgs a;
    a.q = q; a.k = k; a.vt = vt; a.ctx = ctx; a.T = T; a.H = H; a.ldv = ldv;
    a.s_softmax = s_softmax; a.dy_qk = ivit_dyadic{qk_m, qk_r}; a.dy_pv = ivit_dyadic{pv_m, pv_r};
    a.aq = aq; a.et = et; a.cls = cls; a.nc = nc; a.t_count = t_count; a.dmin = dmin;
#ifdef ATT_HAS_ROWTAB
    a.rowtab = rowtab;
#endif
    if (T != 197) return 1;
#ifdef ATT_HAS_ROWTAB
    const bool rows = rowtab != nullptr;
#else
    const bool rows = false;
#endif
    auto launch = [&]() {
#ifdef ATT_HAS_ROWTAB
        if (rows) {
            const size_t lds = AttCfg<4>::SMEM + ATT_ROWLINE_BYTES;
            if (ldv == 0) {      // v row-major
                (void)hipFuncSetAttribute((const void *)attn_fused_kernel<4, true, 197, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                attn_fused_kernel<4, true, 197, 2, true><<<B * H, ATT_WAVES * 64, lds, 0>>>(a);
                return;
            }
            (void)hipFuncSetAttribute((const void *)attn_fused_kernel<4, true, 197, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attn_fused_kernel<4, true, 197, 2><<<B * H, ATT_WAVES * 64, lds, 0>>>(a);
            return;
        }
#endif
        const size_t lds = AttCfg<4>::SMEM + (size_t)((t_count + 3) & ~3) * 4 + (size_t)nc * 512 + 256;
        (void)hipFuncSetAttribute((const void *)attn_fused_kernel<4, true, 197, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attn_fused_kernel<4, true, 197, true><<<B * H, ATT_WAVES * 64, lds, 0>>>(a);
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
