// Stand-alone timing probe for layernorm_reg_kernel (ivit_layernorm.h): DeiT-S shape, random int16 rows.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off [-DLN_ABLATE=n] tools/ubench/ln_probe.hip -o tools/ubench/ln_probe
#include "../../i-vit_amd/csrc/ivit_layernorm.h"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <type_traits>
int main(int argc, char **argv) {
    const long long rows = argc > 1 ? atoll(argv[1]) : 50432;
    constexpr int C = 384;
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
    hipMalloc(&x, rows * C * 2); hipMalloc(&b, C * 4); hipMalloc(&s, C * 4); hipMalloc(&d, C * 16); hipMalloc(&o, rows * C);
    hipMemcpy(x, hx.data(), rows * C * 2, hipMemcpyHostToDevice);
    hipMemcpy(b, hb.data(), C * 4, hipMemcpyHostToDevice);
    hipMemcpy(s, hs.data(), C * 4, hipMemcpyHostToDevice);
    hipMemcpy(d, hd.data(), C * 16, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<int8_t> ref(rows * C), got(rows * C);
    auto run = [&](auto s_t) {
        constexpr int S = decltype(s_t)::value;
        constexpr int rpb = 32;
        const unsigned grid = (unsigned)((rows + rpb - 1) / rpb);
        hipMemset(o, 0, rows * C);
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            for (int i = 0; i < 20; ++i) layernorm_reg_kernel<C, S><<<grid, LNR_THREADS(S)>>>(x, rows, C, 0.0123f, b, s, d, o);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        hipMemcpy(got.data(), o, rows * C, hipMemcpyDeviceToHost);
        if (S == 1) ref = got;
        long long diff = 0;
        for (long long i = 0; i < rows * C; ++i) diff += got[i] != ref[i];
        printf("layernorm_reg<384,%d> rows %lld: %.2f us per launch (%.2f TB/s algorithmic)  differs from S=1 in %lld bytes\n", S, rows,
               best * 1000 / 20, rows * C * 3.0 / (best / 20 * 1e-3) / 1e12, diff);
    };
    // the 39 MB input stays in the 256 MB MALL between launches (as it does in the model, where the GEMM before just wrote it)
    run(std::integral_constant<int, 1>{});
    run(std::integral_constant<int, 2>{});
    run(std::integral_constant<int, 4>{});
    return 0;
}
