// Stand-alone check + timing of mlpr_kernel (ivit_mlpr.h) against mlp384_kernel (ivit_mlp.h), random operands.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/ubench/mlpr_experiment/mlpr_probe.hip -o tools/ubench/mlpr_experiment/mlpr_probe
#include "../../../i-vit_amd/csrc/ivit_mlp.h"
#include "ivit_mlpr.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
__global__ void cvt_probe(const float *in, unsigned *out, int n) {
    int i = threadIdx.x;
    if (i < n) out[i] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 0u, 0u);
}
int main(int argc, char **argv) {
    const long long M = argc > 1 ? atoll(argv[1]) : 50432;
    const int reps = argc > 2 ? atoi(argv[2]) : 3;
    // rounding of v_cvt_pk_u8_f32
    {
        float hv[16] = {0.5f, 1.5f, 2.5f, 0.49999f, 0.99999f, 1.0f, 254.5f, 255.5f, 255.7f, 300.f, -0.5f, -3.f, 127.5f, 128.5f, 3.5f, 1e9f};
        float *d; unsigned *o, ho[16];
        hipMalloc(&d, 64); hipMalloc(&o, 64); hipMemcpy(d, hv, 64, hipMemcpyHostToDevice);
        cvt_probe<<<1, 64>>>(d, o, 16); hipMemcpy(ho, o, 64, hipMemcpyDeviceToHost);
        printf("v_cvt_pk_u8_f32:");
        for (int i = 0; i < 16; ++i) printf(" %g->%u", hv[i], ho[i]);
        printf("\n");
    }
    std::vector<int8_t> hx(M * MLP_C), hw1(MLP_HD * MLP_C), hw2(MLP_C * MLP_HD), htab(65536);
    std::vector<int> hb1(MLP_HD), hb2(MLP_C);
    std::vector<double> hc1(MLP_HD), hc2(MLP_C);
    std::vector<ivit_dyadic> hd1(MLP_HD);
    std::vector<int16_t> hres(M * MLP_C);
    srand(2);
    for (auto &v : hx) v = (int8_t)(rand() % 255 - 127);
    for (auto &v : hw1) v = (int8_t)(rand() % 255 - 127);
    for (auto &v : hw2) v = (int8_t)(rand() % 255 - 127);
    for (auto &v : htab) v = (int8_t)(rand() % 255 - 127);
    for (auto &v : hres) v = (int16_t)(rand() % 60001 - 30000);
    for (int i = 0; i < MLP_HD; ++i) {
        hb1[i] = rand() % 6001 - 3000;
        const double c = 4.0e-4 * (1.0 + (rand() % 1000) / 1000.0);
        int ex; const double mant = frexp(c, &ex);
        hd1[i].m = floor(mant * 2147483648.0); hd1[i].r = ldexp(1.0, ex - 31);
        hc1[i] = hd1[i].m * hd1[i].r;
    }
    for (int i = 0; i < MLP_C; ++i) { hb2[i] = rand() % 6001 - 3000; hc2[i] = 8.0e-3 * (1.0 + (rand() % 1000) / 1000.0); }
    int8_t *x, *w1, *w2, *tab, *wf; int *b1, *b2, *bad; double *c1, *c2; int16_t *res, *out, *out2; v4i *w1f, *w2f; float *c1f;
    ivit_dyadic *dy1; unsigned long long *tr;
    hipMalloc(&x, hx.size()); hipMalloc(&w1, hw1.size()); hipMalloc(&w2, hw2.size()); hipMalloc(&tab, 65536);
    hipMalloc(&b1, MLP_HD * 4); hipMalloc(&b2, MLP_C * 4); hipMalloc(&c1, MLP_HD * 8); hipMalloc(&c2, MLP_C * 8);
    hipMalloc(&res, hres.size() * 2); hipMalloc(&out, hres.size() * 2); hipMalloc(&out2, hres.size() * 2);
    hipMalloc(&w1f, hw1.size()); hipMalloc(&w2f, hw2.size()); hipMalloc(&wf, 2 * hw1.size());
    hipMalloc(&c1f, MLP_HD * 4); hipMalloc(&dy1, MLP_HD * sizeof(ivit_dyadic)); hipMalloc(&bad, 8); hipMemset(bad, 0, 8);
    hipMalloc(&tr, 16384); hipMemset(tr, 0, 16384);
    hipMemcpy(x, hx.data(), hx.size(), hipMemcpyHostToDevice); hipMemcpy(w1, hw1.data(), hw1.size(), hipMemcpyHostToDevice);
    hipMemcpy(w2, hw2.data(), hw2.size(), hipMemcpyHostToDevice); hipMemcpy(tab, htab.data(), 65536, hipMemcpyHostToDevice);
    hipMemcpy(b1, hb1.data(), MLP_HD * 4, hipMemcpyHostToDevice); hipMemcpy(b2, hb2.data(), MLP_C * 4, hipMemcpyHostToDevice);
    hipMemcpy(c1, hc1.data(), MLP_HD * 8, hipMemcpyHostToDevice); hipMemcpy(c2, hc2.data(), MLP_C * 8, hipMemcpyHostToDevice);
    hipMemcpy(dy1, hd1.data(), MLP_HD * sizeof(ivit_dyadic), hipMemcpyHostToDevice);
    hipMemcpy(res, hres.data(), hres.size() * 2, hipMemcpyHostToDevice);
    mlp_swizzle_kernel<<<256, 256>>>(w1, MLP_HD, MLP_C, w1f);
    mlp_swizzle_kernel<<<256, 256>>>(w2, MLP_C, MLP_HD, w2f);
    mlpr_swizzle_kernel<<<256, 256>>>(w1, w2, wf);
    const float dcand[2] = {128.0f, 128.5f};
    float d1 = 0; int hbad = 1;
    for (int i = 0; i < 2 && hbad; ++i) {
        int hb2[2];
        hipMemset(bad, 0, 8);
        mlpr_rq8_plan_kernel<<<MLP_HD, 256>>>(w1, b1, dy1, MLP_HD, MLP_C, dcand[i], c1f, bad);
        hipMemcpy(hb2, bad, 8, hipMemcpyDeviceToHost);
        hbad = hb2[0];
        d1 = dcand[i];
        printf("rq8 plan with d = %g: %d channels unprovable, %d out of range, %d needed a neighbour multiplier\n", d1, hbad & 0xffff, hbad >> 16, hb2[1]);
    }
    MlpArgs a;
    a.x = x; a.w1f = w1f; a.w2f = w2f; a.b1 = b1; a.b2 = b2; a.cq1 = c1; a.cq2 = c2; a.tab = tab; a.residual = res; a.out = out;
    a.cm = 0.645; a.cr = 0.871; a.M = M; a.trace = tr; a.balanced = 1;
    hipFuncSetAttribute((const void *)mlp384_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, MLP_SMEM);
    const long long nunits = ((M + 15) / 16 + MLP_TT - 2) / (MLP_TT - 1);
    const unsigned grid = (unsigned)(nunits < 256 ? nunits : 256);
    MlprArgs q;
    q.x = x; q.wf = wf; q.c1f = c1f; q.b1 = b1; q.b2 = b2; q.cq2 = c2; q.tab = tab; q.residual = res; q.out = out2;
    q.cm = a.cm; q.cr = a.cr; q.d1 = d1; q.M = M; q.trace = tr;
    hipFuncSetAttribute((const void *)mlpr_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, MR_SMEM + (MR_TRACE ? 12288 : 0));
    const long long nwg = (M + 127) / 128;
    const unsigned grid2 = (unsigned)(nwg < 256 ? nwg : 256);
    hipMemset(out, 0x55, hres.size() * 2); hipMemset(out2, 0xaa, hres.size() * 2);
    mlp384_kernel<false><<<grid, MLP_THREADS, MLP_SMEM>>>(a);
    mlpr_kernel<<<grid2, MR_THREADS, MR_SMEM + (MR_TRACE ? 12288 : 0)>>>(q);
    hipError_t err = hipDeviceSynchronize();
    printf("launch: %s\n", hipGetErrorString(err));
    std::vector<int16_t> ho(hres.size()), ho2(hres.size());
    hipMemcpy(ho.data(), out, ho.size() * 2, hipMemcpyDeviceToHost); hipMemcpy(ho2.data(), out2, ho.size() * 2, hipMemcpyDeviceToHost);
    long long nd = 0, first = -1;
    for (size_t i = 0; i < ho.size(); ++i) if (ho[i] != ho2[i]) { if (first < 0) first = (long long)i; ++nd; }
    printf("M %lld: %lld of %zu outputs differ", M, nd, ho.size());
    if (first >= 0) printf(" (first at token %lld channel %lld: %d vs %d)", first / MLP_C, first % MLP_C, ho[first], ho2[first]);
    printf("\n");
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < reps; ++rep) {
        float ms, ms2;
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) mlp384_kernel<false><<<grid, MLP_THREADS, MLP_SMEM>>>(a);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) mlpr_kernel<<<grid2, MR_THREADS, MR_SMEM + (MR_TRACE ? 12288 : 0)>>>(q);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms2, e0, e1);
        printf("M %lld: mlp384 %.1f us (%.0f TOP/s)   mlpr %.1f us (%.0f TOP/s)\n", M, ms * 100, 4.0 * M * MLP_C * MLP_HD / (ms / 10 * 1e-3) / 1e12,
               ms2 * 100, 4.0 * M * MLP_C * MLP_HD / (ms2 / 10 * 1e-3) / 1e12);
    }
    if (MR_TRACE) {
        std::vector<unsigned long long> g(2 * MR_WAVES * 128);
        hipMemcpy(g.data(), tr, g.size() * 8, hipMemcpyDeviceToHost);
        const char *nm[6] = {"pass start", "fc1 done", "table lines", "fc2a done", "fc2b done", "pass done"};
        for (int u = 0; u < 2; ++u) {
            const unsigned long long t0 = g[(u * MR_WAVES) * 128 + 96];
            for (int w = 0; w < MR_WAVES; ++w) {
                const unsigned long long *t = &g[(u * MR_WAVES + w) * 128];
                printf("pass %d wave %d:", u, w);
                for (int pt = 0; pt < 6; ++pt) printf(" %s %lld |", nm[pt], (long long)(t[96 + pt] - t0));
                printf("\n  stage durations:");
                for (int st = 0; st + 1 < 96; ++st) printf(" %lld", (long long)(t[st + 1] - t[st]));
                printf("\n");
            }
        }
    }
    return 0;
}
