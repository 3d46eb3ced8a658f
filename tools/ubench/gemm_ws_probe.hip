// gemm_ws_probe.hip — i-vit_amd/csrc/ivit_gemm_ws.h alone as a small shared library (tools/gemm_ws_probe.py feeds it the operands of
// ivit_layernorm_requant + ivit_linear_i8_qkv_planned and compares the outputs byte for byte).  ln = 1: x16 + norm1's constants
// instead of the 8-bit activations.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared tools/ubench/gemm_ws_probe.hip -o tools/ubench/libgemm_ws_probe.so
#include "../../i-vit_amd/csrc/ivit_gemm_ws.h"
#include <stdio.h>

extern "C" int gemm_ws_probe(const int8_t *x, const int8_t *w, const int32_t *bias, const double *cq, int8_t *q, int8_t *k, int8_t *v,
                             int M, int N, int T, int H, int fma, int grid, int reps, float *us, int ln, const int16_t *x16, float ln_s,
                             const float *ln_bias_int, const float *ln_sc, const ivit_dyadic *ln_dy) {
    if (N % 192 || N > WS_MAXN) return 1;
    v4i *wf = nullptr;
    void *dummy = nullptr;
    long long *trace = nullptr;
    if (hipMalloc((void **)&wf, (size_t)N * WS_K) != hipSuccess || hipMalloc(&dummy, 4096) != hipSuccess || hipMalloc((void **)&trace, 8 * 64 * 8) != hipSuccess) return 2;
    (void)hipMemset(trace, 0, 8 * 64 * 8);
    ws_swizzle_kernel<<<64, 256>>>(w, wf, N);
    WsArgs a{x, wf, bias, cq, q, k, v, M, N, T, H, dummy, x16, ln_s, ln_bias_int, ln_sc, ln_dy, nullptr, nullptr, 0.0, 0.0, nullptr, trace};
    auto launch = [&]() {
#define WS_L(F, L) do { (void)hipFuncSetAttribute((const void *)gemm_ws_qkv_kernel<F, L>, hipFuncAttributeMaxDynamicSharedMemorySize, WS_SMEM); \
                        gemm_ws_qkv_kernel<F, L><<<grid, WS_THREADS, WS_SMEM, 0>>>(a); } while (0)
        if (fma && ln) WS_L(true, true); else if (fma) WS_L(true, false); else if (ln) WS_L(false, true); else WS_L(false, false);
    };
    launch();
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(hipGetLastError())); return 3; }
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        (void)hipEventRecord(e0, 0);
        for (int r = 0; r < reps; ++r) launch();
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    *us = best * 1000.f / reps;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (WS_TRACE) {
        long long ht[8 * 64];
        (void)hipMemcpy(ht, trace, sizeof(ht), hipMemcpyDeviceToHost);
        long long t0 = ht[0];
        for (int w8 = 1; w8 < 8; ++w8) if (ht[w8 * 64] < t0) t0 = ht[w8 * 64];
        for (int w8 = 0; w8 < 8; ++w8) {
            printf("wave %d:", w8);
            for (int i = 0; i < 64 && ht[w8 * 64 + i]; ++i) printf(" %lld", ht[w8 * 64 + i] - t0);
            printf("\n");
        }
    }
    (void)hipFree(trace); (void)hipFree(dummy); (void)hipFree(wf);
    return hipDeviceSynchronize() == hipSuccess ? 0 : 4;
}

// attn.proj + residual QuantAct (EPI_RES16): out16 [M][N]
extern "C" int gemm_ws_probe_res(const int8_t *x, const int8_t *w, const int32_t *bias, const double *cq, const int16_t *residual, int16_t *out16,
                                 double cm, double cr, int M, int N, int fma, int grid, int reps, float *us) {
    if (N % 64 || N > WS_MAXN) return 1;
    v4i *wf = nullptr;
    void *dummy = nullptr;
    if (hipMalloc((void **)&wf, (size_t)N * WS_K) != hipSuccess || hipMalloc(&dummy, 4096) != hipSuccess) return 2;
    ws_swizzle_kernel<<<64, 256>>>(w, wf, N);
    WsArgs a{x, wf, bias, cq, nullptr, nullptr, nullptr, M, N, 1, 1, dummy, nullptr, 0.f, nullptr, nullptr, nullptr, residual, out16, cm, cr, nullptr, nullptr};
    auto launch = [&]() {
        if (fma) { (void)hipFuncSetAttribute((const void *)gemm_ws_qkv_kernel<true, false, WS_EPI_RES16>, hipFuncAttributeMaxDynamicSharedMemorySize, WS_SMEM);
                   gemm_ws_qkv_kernel<true, false, WS_EPI_RES16><<<grid, WS_THREADS, WS_SMEM, 0>>>(a); }
        else { (void)hipFuncSetAttribute((const void *)gemm_ws_qkv_kernel<false, false, WS_EPI_RES16>, hipFuncAttributeMaxDynamicSharedMemorySize, WS_SMEM);
               gemm_ws_qkv_kernel<false, false, WS_EPI_RES16><<<grid, WS_THREADS, WS_SMEM, 0>>>(a); }
    };
    launch();
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(hipGetLastError())); return 3; }
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        (void)hipEventRecord(e0, 0);
        for (int r = 0; r < reps; ++r) launch();
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    *us = best * 1000.f / reps;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(dummy); (void)hipFree(wf);
    return hipDeviceSynchronize() == hipSuccess ? 0 : 4;
}
