// Stand-alone check + timing of the role-split fused Mlp (ivit_mlp_rs.h) against the shipped one (ivit_mlp.h): same random
// operands through both kernels, outputs compared bit for bit at several token counts, then timings and (RS_TRACE=1) the
// producer / consumer timeline of workgroup 0.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DRS_TRACE=1 tools/ubench/mlp_rs_probe.hip -o tools/ubench/mlp_rs_probe
#include "../../i-vit_amd/csrc/ivit_mlp_rs.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)
int main(int argc, char **argv) {
    const long long MMAX = 50432;
    std::vector<int8_t> hx(MMAX * MLP_C), hw1(MLP_HD * MLP_C), hw2(MLP_C * MLP_HD), htab(65536);
    std::vector<int> hb1(MLP_HD), hb2(MLP_C);
    std::vector<double> hc1(MLP_HD), hc2(MLP_C);
    std::vector<int16_t> hres(MMAX * MLP_C);
    srand(2);
    for (auto &v : hx) v = (int8_t)(rand() % 255 - 127);
    for (auto &v : hw1) v = (int8_t)(rand() % 255 - 127);
    for (auto &v : hw2) v = (int8_t)(rand() % 255 - 127);
    for (auto &v : htab) v = (int8_t)(rand() % 255 - 127);
    for (auto &v : hres) v = (int16_t)(rand() % 60001 - 30000);
    for (int i = 0; i < MLP_HD; ++i) { hb1[i] = rand() % 6001 - 3000; hc1[i] = 4.0e-4 * (1.0 + (rand() % 1000) / 1000.0); }
    for (int i = 0; i < MLP_C; ++i) { hb2[i] = rand() % 6001 - 3000; hc2[i] = 8.0e-3 * (1.0 + (rand() % 1000) / 1000.0); }
    int8_t *x, *w1, *w2, *tab; int *b1, *b2; double *c1, *c2; int16_t *res, *out, *out2; v4i *w1f, *w2f, *w1r, *w2r; unsigned long long *tr;
    CK(hipMalloc(&x, hx.size())); CK(hipMalloc(&w1, hw1.size())); CK(hipMalloc(&w2, hw2.size())); CK(hipMalloc(&tab, 65536));
    CK(hipMalloc(&b1, MLP_HD * 4)); CK(hipMalloc(&b2, MLP_C * 4)); CK(hipMalloc(&c1, MLP_HD * 8)); CK(hipMalloc(&c2, MLP_C * 8));
    CK(hipMalloc(&res, hres.size() * 2)); CK(hipMalloc(&out, hres.size() * 2)); CK(hipMalloc(&out2, hres.size() * 2));
    CK(hipMalloc(&w1f, hw1.size())); CK(hipMalloc(&w2f, hw2.size())); CK(hipMalloc(&w1r, hw1.size())); CK(hipMalloc(&w2r, hw2.size()));
    const size_t trn = 4 * 8 * 8 + 256;
    CK(hipMalloc(&tr, trn * 8)); CK(hipMemset(tr, 0, trn * 8));
    CK(hipMemcpy(x, hx.data(), hx.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(w1, hw1.data(), hw1.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(w2, hw2.data(), hw2.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(tab, htab.data(), 65536, hipMemcpyHostToDevice));
    CK(hipMemcpy(b1, hb1.data(), MLP_HD * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(b2, hb2.data(), MLP_C * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(c1, hc1.data(), MLP_HD * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(c2, hc2.data(), MLP_C * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(res, hres.data(), hres.size() * 2, hipMemcpyHostToDevice));
    mlp_swizzle_kernel<<<256, 256>>>(w1, MLP_HD, MLP_C, w1f);
    mlp_swizzle_kernel<<<256, 256>>>(w2, MLP_C, MLP_HD, w2f);
    rs_swizzle_w1_kernel<<<144, 256>>>(w1, w1r);
    rs_swizzle_w2_kernel<<<144, 256>>>(w2, w2r);
    CK(hipDeviceSynchronize());
    CK(hipFuncSetAttribute((const void *)mlp384_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, MLP_SMEM));
    CK(hipFuncSetAttribute((const void *)mlp384rs_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, RS_SMEM));
    auto args = [&](long long M, int16_t *o, bool rs) {
        MlpArgs a;
        a.x = x; a.w1f = rs ? w1r : w1f; a.w2f = rs ? w2r : w2f; a.b1 = b1; a.b2 = b2; a.cq1 = c1; a.cq2 = c2; a.tab = tab; a.residual = res; a.out = o;
        a.cm = 0.645; a.cr = 0.871; a.M = M; a.trace = tr;
        const long long ntiles = (M + 15) / 16, nunits = (ntiles + MLP_TT - 2) / (MLP_TT - 1);
        const long long grid = nunits < 256 ? nunits : 256;
        const long long rounds_fixed = (nunits + grid - 1) / grid, rounds_bal = (ntiles + (long long)MLP_TT * grid - 1) / ((long long)MLP_TT * grid);
        a.balanced = rounds_bal < rounds_fixed;
        return a;
    };
    auto grid_of = [&](long long M) { const long long nunits = ((M + 15) / 16 + MLP_TT - 2) / (MLP_TT - 1); return (unsigned)(nunits < 256 ? nunits : 256); };
    const long long Ms[] = {16, 17, 80, 100, 1000, 4097, 20479, 25216, 50432};
    const int nM = argc > 1 ? atoi(argv[1]) : 9;
    std::vector<int16_t> ha(MMAX * MLP_C), hb(MMAX * MLP_C);
    int bad_total = 0;
    for (int mi = 0; mi < nM; ++mi) {
        const long long M = Ms[mi];
        CK(hipMemset(out, 0x55, M * MLP_C * 2)); CK(hipMemset(out2, 0x55, M * MLP_C * 2));
        MlpArgs a = args(M, out, false), b = args(M, out2, true);
        mlp384_kernel<true><<<grid_of(M), MLP_THREADS, MLP_SMEM>>>(a);
        mlp384rs_kernel<true><<<grid_of(M), RS_THREADS, RS_SMEM>>>(b);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(ha.data(), out, M * MLP_C * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), out2, M * MLP_C * 2, hipMemcpyDeviceToHost));
        long long bad = 0, first = -1;
        for (long long i = 0; i < M * MLP_C; ++i) if (ha[i] != hb[i]) { if (first < 0) first = i; ++bad; }
        printf("M %6lld (balanced %d): %lld of %lld outputs differ", M, a.balanced, bad, M * MLP_C);
        if (bad) printf("; first at row %lld col %lld: shipped %d role-split %d", first / MLP_C, first % MLP_C, ha[first], hb[first]);
        printf("\n");
        bad_total += bad != 0;
    }
    const long long Mt[] = {50432, 25216, 20480};
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (long long M : Mt) {
        MlpArgs a = args(M, out, false), b = args(M, out2, true);
        for (int rep = 0; rep < 3; ++rep) {
            float ms0, ms1;
            hipEventRecord(e0);
            for (int i = 0; i < 10; ++i) mlp384_kernel<true><<<grid_of(M), MLP_THREADS, MLP_SMEM>>>(a);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms0, e0, e1);
            hipEventRecord(e0);
            for (int i = 0; i < 10; ++i) mlp384rs_kernel<true><<<grid_of(M), RS_THREADS, RS_SMEM>>>(b);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms1, e0, e1);
            printf("M %lld: shipped %.1f us, role-split %.1f us per launch (%.0f TOP/s)\n", M, ms0 * 100, ms1 * 100, 4.0 * M * MLP_C * MLP_HD / (ms1 / 10 * 1e-3) / 1e12);
        }
    }
    if (RS_TRACE) {
        MlpArgs b = args(50432, out2, true);
        CK(hipMemset(tr, 0, trn * 8));
        mlp384rs_kernel<true><<<grid_of(50432), RS_THREADS, RS_SMEM>>>(b);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> h(4 * 8 * 8);
        CK(hipMemcpy(h.data(), tr, h.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull;
        for (auto v : h) if (v && v < t0) t0 = v;
        const char *nm[8] = {"P: activation tile landed", "P: K loops done", "P: hidden tile written", "C: hidden tile complete", "C: ShiftGELU done (all)", "C: fc2 K loop done", "C: epilogue done", ""};
        for (int u = 0; u < 4; ++u) {
            printf("unit %d (cycles since the first stamp of workgroup 0; waves 0-3 producers, 4-7 consumers)\n", u);
            for (int pt = 0; pt < 7; ++pt) {
                printf("  %-28s", nm[pt]);
                for (int w = 0; w < 8; ++w) { const unsigned long long v = h[(u * 8 + w) * 8 + pt]; if (v) printf(" %7lld", (long long)(v - t0)); else printf("       -"); }
                printf("\n");
            }
        }
    }
    return bad_total ? 2 : 0;
}
