// Stand-alone probe: layernorm_reg_kernel<384, 2> at 50 432 rows with its residency capped by a dummy dynamic-LDS request
// (5, 4, 3, 2 blocks of 4 waves per CU): does a shorter first round + a fuller second one beat 1.23 rounds at full occupancy?
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Xclang -target-feature -Xclang -packed-fp32-ops tools/ubench/ln_occ_probe.hip -o tools/ubench/ln_occ_probe
#include "../../i-vit_amd/csrc/ivit_layernorm.h"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
int main(int argc, char **argv) {
    const long long rows = argc > 1 ? atoll(argv[1]) : 50432;
    constexpr int C = 384, S = 2;
    std::vector<int16_t> hx(rows * C);
    std::vector<float> hb(C), hs(C);
    std::vector<ivit_dyadic> hd(C);
    srand(1);
    for (auto &v : hx) v = (int16_t)((rand() % 4001) - 2000);
    for (int c = 0; c < C; ++c) {
        hb[c] = (float)((rand() % 200001) - 100000) * 1000.f;
        hs[c] = (0.5f + (rand() % 1000) / 1000.f) * 3e-9f * ((rand() & 1) ? 1.f : -1.f);
        hd[c].m = 1073741824.0 + rand();
        hd[c].r = 1.0 / 9007199254740992.0 / 4.0;
    }
    int16_t *x; float *b, *s; ivit_dyadic *d; int8_t *o;
    (void)hipMalloc(&x, rows * C * 2); (void)hipMalloc(&b, C * 4); (void)hipMalloc(&s, C * 4); (void)hipMalloc(&d, C * 16); (void)hipMalloc(&o, rows * C);
    (void)hipMemcpy(x, hx.data(), rows * C * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(b, hb.data(), C * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(s, hs.data(), C * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(d, hd.data(), C * 16, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    constexpr int rpb = (LNR_THREADS(S) / 64) * (64 / (4 * S));
    const unsigned grid = (unsigned)((rows + rpb - 1) / rpb);
    (void)hipFuncSetAttribute((const void *)layernorm_reg_kernel<C, S>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    for (int kb : {0, 20, 30, 44, 70, 0}) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            (void)hipEventRecord(e0);
            for (int i = 0; i < 20; ++i) layernorm_reg_kernel<C, S><<<grid, LNR_THREADS(S), kb * 1024>>>(x, rows, C, 0.0123f, b, s, d, o);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        printf("rows %lld, %2d KB dummy LDS (%d blocks per CU by LDS): %6.2f us per launch\n", rows, kb, (int)(160 / (7.75 + kb)), best * 1000 / 20);
    }
    return 0;
}
