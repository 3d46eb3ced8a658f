// NUMERICALLY WRONG AS IT STANDS: the equality check below is RED (see ivit_gemm4.h); timing experiment only, outside the build.
// Stand-alone check + timing of tok4_qkv_kernel (ivit_gemm4.h) against a plain reference kernel, random operands.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/ubench/tok4_probe.hip -o tools/ubench/tok4_probe
#include "ivit_gemm4.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
__global__ void ref_qkv(const int8_t *x, const int8_t *w, const int *bias, const double *c, int8_t *q, int8_t *k, int8_t *vt,
                        long long M, int N, int T, int H, int ldv) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= M * N) return;
    const long long tok = i / N; const int ch = (int)(i % N);
    int z = bias[ch];
    for (int kk = 0; kk < 384; ++kk) z += (int)x[tok * 384 + kk] * (int)w[(long long)ch * 384 + kk];
    double t = __builtin_rint((double)z * c[ch]);
    t = t < -128 ? -128 : (t > 127 ? 127 : t);
    const int D = N / 3, which = ch / D, within = ch % D, head = within / 64, d0 = within % 64;
    const int b = (int)(tok / T), tp = (int)(tok % T);
    if (which < 2) (which ? k : q)[((long long)(b * H + head) * T + tp) * 64 + d0] = (int8_t)t;
    else vt[((long long)(b * H + head) * 64 + d0) * ldv + tp] = (int8_t)t;
}
int main(int argc, char **argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 256, T = 197, H = 6, N = 1152, ldv = 208;
    const long long M = (long long)B * T;
    std::vector<int8_t> hx(M * 384), hw(N * 384);
    std::vector<int> hb(N); std::vector<double> hc(N);
    srand(3);
    for (auto &v : hx) v = (int8_t)(rand() % 255 - 127);
    for (auto &v : hw) v = (int8_t)(rand() % 255 - 127);
    for (int i = 0; i < N; ++i) { hb[i] = rand() % 6001 - 3000; hc[i] = ldexp(floor((0.5 + (rand() % 1000) / 2000.0) * 2147483648.0), -31 - 11); }
    int8_t *x, *w, *wf, *q, *k, *vt, *q2, *k2, *vt2; int *b; double *c; unsigned long long *tr;
    const size_t qb = (size_t)B * H * T * 64, vb = (size_t)B * H * 64 * ldv;
    hipMalloc(&x, hx.size()); hipMalloc(&w, hw.size()); hipMalloc(&wf, hw.size()); hipMalloc(&b, N * 4); hipMalloc(&c, N * 8);
    hipMalloc(&q, qb); hipMalloc(&k, qb); hipMalloc(&vt, vb); hipMalloc(&q2, qb); hipMalloc(&k2, qb); hipMalloc(&vt2, vb); hipMalloc(&tr, 8 * 64 * 8);
    hipMemset(vt, 0, vb); hipMemset(vt2, 0, vb); hipMemset(tr, 0, 8 * 64 * 8);
    hipMemcpy(x, hx.data(), hx.size(), hipMemcpyHostToDevice); hipMemcpy(w, hw.data(), hw.size(), hipMemcpyHostToDevice);
    hipMemcpy(b, hb.data(), N * 4, hipMemcpyHostToDevice); hipMemcpy(c, hc.data(), N * 8, hipMemcpyHostToDevice);
    t4_swizzle_kernel<<<256, 256>>>(w, N, wf);
    ref_qkv<<<(unsigned)((M * N + 255) / 256), 256>>>(x, w, b, c, q, k, vt, M, N, T, H, ldv);
    Tok4Args a; a.x = x; a.wf = wf; a.cq = c; a.bias = b; a.q = q2; a.k = k2; a.vt = vt2; a.M = M; a.T = T; a.H = H; a.ldv = ldv; a.trace = tr;
    hipFuncSetAttribute((const void *)tok4_qkv_kernel<36, false>, hipFuncAttributeMaxDynamicSharedMemorySize, T4_SMEM(1152));
    const long long nwg = (M + 255) / 256;
    const unsigned grid = (unsigned)(nwg < 256 ? nwg : 256);
    tok4_qkv_kernel<36, false><<<grid, T4_THREADS, T4_SMEM(1152)>>>(a);
    printf("launch: %s\n", hipGetErrorString(hipDeviceSynchronize()));
    std::vector<int8_t> h1(qb), h2(qb);
    long long nd = 0;
    hipMemcpy(h1.data(), q, qb, hipMemcpyDeviceToHost); hipMemcpy(h2.data(), q2, qb, hipMemcpyDeviceToHost);
    for (size_t i = 0; i < qb; ++i) nd += h1[i] != h2[i];
    printf("q: %lld of %zu differ\n", nd, qb); nd = 0;
    hipMemcpy(h1.data(), k, qb, hipMemcpyDeviceToHost); hipMemcpy(h2.data(), k2, qb, hipMemcpyDeviceToHost);
    for (size_t i = 0; i < qb; ++i) nd += h1[i] != h2[i];
    printf("k: %lld of %zu differ\n", nd, qb); nd = 0;
    std::vector<int8_t> g1(vb), g2(vb);
    hipMemcpy(g1.data(), vt, vb, hipMemcpyDeviceToHost); hipMemcpy(g2.data(), vt2, vb, hipMemcpyDeviceToHost);
    for (size_t i = 0; i < vb; ++i) nd += g1[i] != g2[i];
    printf("vt: %lld of %zu differ\n", nd, vb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        float ms;
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) tok4_qkv_kernel<36, false><<<grid, T4_THREADS, T4_SMEM(1152)>>>(a);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("B %d M %lld: tok4 qkv %.1f us (%.0f TOP/s)\n", B, M, ms * 50, 2.0 * M * N * 384 / (ms / 20 * 1e-3) / 1e12);
    }
    if (T4_TRACE) {
        std::vector<unsigned long long> h(8 * 64);
        hipMemcpy(h.data(), tr, h.size() * 8, hipMemcpyDeviceToHost);
        for (int wv = 0; wv < 8; wv += 7) { printf("wave %d stage durations:", wv); for (int s = 0; s + 1 < 36; ++s) printf(" %lld", (long long)(h[wv * 64 + s + 1] - h[wv * 64 + s])); printf("\n"); }
    }
    return 0;
}
