// Stand-alone timing / timeline probe for the EXPERIMENTAL 32x32x32 Mlp kernel (ivit_mlp32.h), random operands.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DMLP_TRACE=1 tools/ubench/mlp_probe.hip -o tools/ubench/mlp_probe
#include "ivit_mlp32.h"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
int main(int argc, char **argv) {
    const long long M = argc > 1 ? atoll(argv[1]) : 50432;
    std::vector<int8_t> hx(M * MLP_C), hw1(MLP_HD * MLP_C), hw2(MLP_C * MLP_HD), htab(65536);
    std::vector<int> hb1(MLP_HD), hb2(MLP_C);
    std::vector<double> hc1(MLP_HD), hc2(MLP_C);
    std::vector<int16_t> hres(M * MLP_C);
    srand(2);
    const bool flat = getenv("MLP_FLAT") != nullptr;   // constant operands: how much of the time is data-dependent (power)
    for (auto &v : hx) v = flat ? 1 : (int8_t)(rand() % 255 - 127);
    for (auto &v : hw1) v = flat ? 1 : (int8_t)(rand() % 255 - 127);
    for (auto &v : hw2) v = flat ? 1 : (int8_t)(rand() % 255 - 127);
    for (auto &v : htab) v = (int8_t)(rand() % 255 - 127);
    for (auto &v : hres) v = (int16_t)(rand() % 60001 - 30000);
    for (int i = 0; i < MLP_HD; ++i) { hb1[i] = rand() % 6001 - 3000; hc1[i] = 4.0e-4 * (1.0 + (rand() % 1000) / 1000.0); }
    for (int i = 0; i < MLP_C; ++i) { hb2[i] = rand() % 6001 - 3000; hc2[i] = 8.0e-3 * (1.0 + (rand() % 1000) / 1000.0); }
    int8_t *x, *w1, *w2, *tab; int *b1, *b2; double *c1, *c2; int16_t *res, *out; v4i *w1f, *w2f; unsigned long long *tr;
    hipMalloc(&x, hx.size()); hipMalloc(&w1, hw1.size()); hipMalloc(&w2, hw2.size()); hipMalloc(&tab, 65536);
    hipMalloc(&b1, MLP_HD * 4); hipMalloc(&b2, MLP_C * 4); hipMalloc(&c1, MLP_HD * 8); hipMalloc(&c2, MLP_C * 8);
    hipMalloc(&res, hres.size() * 2); hipMalloc(&out, hres.size() * 2); hipMalloc(&w1f, hw1.size()); hipMalloc(&w2f, hw2.size());
    hipMalloc(&tr, (4 * MLP_WAVES * 8 + MLP_WAVES * 32) * 8); hipMemset(tr, 0, (4 * MLP_WAVES * 8 + MLP_WAVES * 32) * 8);
    hipMemcpy(x, hx.data(), hx.size(), hipMemcpyHostToDevice); hipMemcpy(w1, hw1.data(), hw1.size(), hipMemcpyHostToDevice);
    hipMemcpy(w2, hw2.data(), hw2.size(), hipMemcpyHostToDevice); hipMemcpy(tab, htab.data(), 65536, hipMemcpyHostToDevice);
    hipMemcpy(b1, hb1.data(), MLP_HD * 4, hipMemcpyHostToDevice); hipMemcpy(b2, hb2.data(), MLP_C * 4, hipMemcpyHostToDevice);
    hipMemcpy(c1, hc1.data(), MLP_HD * 8, hipMemcpyHostToDevice); hipMemcpy(c2, hc2.data(), MLP_C * 8, hipMemcpyHostToDevice);
    hipMemcpy(res, hres.data(), hres.size() * 2, hipMemcpyHostToDevice);
    mlp_swizzle_kernel<<<256, 256>>>(w1, MLP_HD, MLP_C, w1f);
    mlp_swizzle_kernel<<<256, 256>>>(w2, MLP_C, MLP_HD, w2f);
    MlpArgs a;
    a.x = x; a.w1f = w1f; a.w2f = w2f; a.b1 = b1; a.b2 = b2; a.cq1 = c1; a.cq2 = c2; a.tab = tab; a.residual = res; a.out = out;
    a.cm = 0.645; a.cr = 0.871; a.M = M; a.trace = tr;
    hipFuncSetAttribute((const void *)mlp384_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, MLP_SMEM);
    const long long nunits = ((M + 31) / 32 + 1) / 2;
    const unsigned grid = (unsigned)(nunits < 256 ? nunits : 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) mlp384_kernel<true><<<grid, MLP_THREADS, MLP_SMEM>>>(a);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("mlp384 M %lld: %.1f us per launch, %.0f TOP/s\n", M, ms * 100, 4.0 * M * MLP_C * MLP_HD / (ms / 10 * 1e-3) / 1e12);
    }
    if (MLP_TRACE) {
        std::vector<unsigned long long> h(4 * MLP_WAVES * 8);
        hipMemcpy(h.data(), tr, h.size() * 8, hipMemcpyDeviceToHost);
        const char *nm[8] = {"start", "A tile + barrier", "fc1 (K loops + epilogues)", "barrier B2", "-",
                             "GELU + barrier", "fc2 K loop", "fc2 epilogue"};
        for (int u = 0; u < 4; ++u) {
            printf("unit %d (cycles since the unit's first stamp; per wave)\n", u);
            unsigned long long t0 = ~0ull;
            for (int w = 0; w < MLP_WAVES; ++w) if (h[(u * MLP_WAVES + w) * 8] && h[(u * MLP_WAVES + w) * 8] < t0) t0 = h[(u * MLP_WAVES + w) * 8];
            for (int pt = 0; pt < 8; ++pt) {
                printf("  %-28s", nm[pt]);
                for (int w = 0; w < MLP_WAVES; ++w) printf(" %6lld", (long long)(h[(u * MLP_WAVES + w) * 8 + pt] - t0));
                printf("\n");
            }
        }
    }
    if (MLP_TRACE == 2) {
        std::vector<unsigned long long> h(MLP_WAVES * 16);
        hipMemcpy(h.data(), tr + 4 * MLP_WAVES * 8, h.size() * 8, hipMemcpyDeviceToHost);
        unsigned long long t0 = ~0ull;
        for (int w = 0; w < MLP_WAVES; ++w) if (h[w * 16] && h[w * 16] < t0) t0 = h[w * 16];
        printf("fc2 K loop, unit 0: cycle at which each wave issues step 0, 6, ..., 42, 47 (since the first wave's step 0)\n");
        for (int w = 0; w < MLP_WAVES; ++w) {
            printf("  wave %2d:", w);
            for (int k = 0; k < 9; ++k) printf(" %6lld", (long long)(h[w * 16 + k] - t0));
            printf("\n");
        }
    }
    return 0;
}
