// Stand-alone probe: layernorm_plan_kernel (constants precomputed, staged by LDS-DMA behind the row loads) against
// layernorm_reg_kernel on the same random rows, byte for byte, several row counts.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Xclang -target-feature -Xclang -packed-fp32-ops \
//        tools/ubench/ln_plan_probe.hip -o tools/ubench/ln_plan_probe
#include "../experiments/ivit_layernorm_plan.h"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
int main(int argc, char **argv) {
    constexpr int C = 384, S = 2;
    const long long maxrows = 100864;
    std::vector<int16_t> hx(maxrows * C);
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
    int16_t *x; float *b, *s; ivit_dyadic *d; int8_t *o; char *blob; int *flag;
    (void)hipMalloc(&x, maxrows * C * 2); (void)hipMalloc(&b, C * 4); (void)hipMalloc(&s, C * 4); (void)hipMalloc(&d, C * 16); (void)hipMalloc(&o, maxrows * C);
    (void)hipMalloc(&blob, 20 * C); (void)hipMalloc(&flag, 4);
    (void)hipMemcpy(x, hx.data(), maxrows * C * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(b, hb.data(), C * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(s, hs.data(), C * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(d, hd.data(), C * 16, hipMemcpyHostToDevice);
    (void)hipMemset(flag, 0, 4);
    layernorm_plan_build_kernel<<<(C + 255) / 256, 256>>>(b, s, d, C, blob, flag);
    int wide = 0;
    (void)hipMemcpy(&wide, flag, 4, hipMemcpyDeviceToHost);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    std::vector<int8_t> ref(maxrows * C), got(maxrows * C);
    for (long long rows : {256LL, 8192LL, 25216LL, 40960LL, 50432LL, 100864LL}) {
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
            printf("%-24s rows %6lld: %6.2f us per launch  %lld bytes differ\n", name, rows, best * 1000 / 20, diff);
        };
        constexpr int rpb = (LNR_THREADS(S) / 64) * (64 / (4 * S));
        const unsigned grid = (unsigned)((rows + rpb - 1) / rpb);
        timeit([&] { layernorm_reg_kernel<C, S><<<grid, LNR_THREADS(S)>>>(x, rows, C, 0.0123f, b, s, d, o); }, "one-shot (staged)", true);
        timeit([&] { layernorm_plan_kernel<C, S><<<grid, LNR_THREADS(S)>>>(x, rows, C, 0.0123f, blob, !wide, o); }, "planned (DMA constants)", false);
    }
    return 0;
}
