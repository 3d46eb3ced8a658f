// Stand-alone probe: layernorm_pk_kernel (hand-packed plain-form v_pk_*_f32, tools/experiments/ivit_layernorm_pk.h) against
// layernorm_reg_kernel on the same random rows, byte for byte, several row counts and both lane splits.
// Build (packed fp32 must be ENABLED for the assembler): hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize \
//        tools/ubench/ln_pk_probe.hip -o tools/ubench/ln_pk_probe
#include "../experiments/ivit_layernorm_pk.h"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
template <int C, int S>
void run_case(long long maxrows) {
    std::vector<int16_t> hx(maxrows * C);
    std::vector<float> hb(C), hs(C);
    std::vector<ivit_dyadic> hd(C);
    srand(1 + C);
    for (auto &v : hx) v = (int16_t)((rand() % 40001) - 20000);
    for (int c = 0; c < C; ++c) {
        hb[c] = (float)((rand() % 200001) - 100000) * 1000.f;
        hs[c] = (0.5f + (rand() % 1000) / 1000.f) * 3e-9f * ((rand() & 1) ? 1.f : -1.f);
        hd[c].m = 1073741824.0 + rand();
        hd[c].r = 1.0 / 9007199254740992.0 / 4.0;
    }
    int16_t *x; float *b, *s; ivit_dyadic *d; int8_t *o;
    (void)hipMalloc(&x, maxrows * C * 2); (void)hipMalloc(&b, C * 4); (void)hipMalloc(&s, C * 4); (void)hipMalloc(&d, C * 16); (void)hipMalloc(&o, maxrows * C);
    (void)hipMemcpy(x, hx.data(), maxrows * C * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(b, hb.data(), C * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(s, hs.data(), C * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(d, hd.data(), C * 16, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    std::vector<int8_t> ref(maxrows * C), got(maxrows * C);
    for (long long rows : {maxrows / 2, maxrows}) {
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
            printf("C %4d S %d %-22s rows %6lld: %6.2f us per launch  %lld bytes differ\n", C, S, name, rows, best * 1000 / 20, diff);
        };
        constexpr int rpb = (LNR_THREADS(S) / 64) * (64 / (4 * S));
        const unsigned grid = (unsigned)((rows + rpb - 1) / rpb);
        timeit([&] { layernorm_reg_kernel<C, S><<<grid, LNR_THREADS(S)>>>(x, rows, C, 0.0123f, b, s, d, o); }, "shipped (scalar fp32)", true);
        timeit([&] { layernorm_pk_kernel<C, S><<<grid, LNR_THREADS(S)>>>(x, rows, C, 0.0123f, b, s, d, o); }, "hand-packed", false);
    }
    (void)hipFree(x); (void)hipFree(b); (void)hipFree(s); (void)hipFree(d); (void)hipFree(o);
}
int main() {
    run_case<384, 2>(50432);
    run_case<768, 4>(50432);
    run_case<192, 2>(50432);
    return 0;
}
