// Stand-alone probe: layernorm_pipe_kernel (persistent, next row group in flight under the arithmetic) against
// layernorm_reg_kernel (one row group per wave) on the same random rows, compared byte for byte, for several grid sizes.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Xclang -target-feature -Xclang -packed-fp32-ops \
//        tools/ubench/ln_pipe_probe.hip -o tools/ubench/ln_pipe_probe
#include "../experiments/ivit_layernorm_pipe.h"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#ifndef PROBE_C
#define PROBE_C 384
#endif
#ifndef PROBE_S
#define PROBE_S 2
#endif
int main(int argc, char **argv) {
    const long long rows = argc > 1 ? atoll(argv[1]) : 50432;
    constexpr int C = PROBE_C, S = PROBE_S;
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
    std::vector<int8_t> ref(rows * C), got(rows * C);
    auto timeit = [&](auto launch, const char *name, bool is_ref) {
        (void)hipMemset(o, 0, rows * C);
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            (void)hipEventRecord(e0);
            for (int i = 0; i < 20; ++i) launch();
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        (void)hipMemcpy(got.data(), o, rows * C, hipMemcpyDeviceToHost);
        if (is_ref) ref = got;
        long long diff = 0;
        for (long long i = 0; i < rows * C; ++i) diff += got[i] != ref[i];
        printf("%-34s rows %lld C %d: %6.2f us per launch (%.2f TB/s algorithmic)  %lld bytes differ\n", name, rows, C, best * 1000 / 20,
               rows * C * 3.0 / (best / 20 * 1e-3) / 1e12, diff);
    };
    constexpr int rpb = (LNR_THREADS(S) / 64) * (64 / (4 * S));
    timeit([&] { layernorm_reg_kernel<C, S><<<(unsigned)((rows + rpb - 1) / rpb), LNR_THREADS(S)>>>(x, rows, C, 0.0123f, b, s, d, o); },
           "layernorm_reg_kernel (one-shot)", true);
    const long long ngroups = (rows + (64 / (4 * S)) - 1) / (64 / (4 * S));
    const unsigned need = (unsigned)((ngroups + 3) / 4);
    for (int q = 4; q <= 24; q += (q < 12 ? 1 : 4)) {     // quarter-blocks per CU: 1, 1.25, ... 6
        unsigned grid = 256u * q / 4;
        if (grid > need) grid = need;
        char name[64];
        snprintf(name, sizeof name, "layernorm_pipe_kernel %.2f blk/CU", q / 4.0);
        timeit([&] { layernorm_pipe_kernel<C, S><<<grid, 256>>>(x, rows, C, 0.0123f, b, s, d, o); }, name, false);
    }
    return 0;
}
