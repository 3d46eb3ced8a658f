// Stand-alone reproduction attempt of the round-3 / round-4 one-LSB LayerNorm differences (VERDICT r4 #6b): NO library, no Python —
// the victim is layernorm_reg_kernel<192, 1> exactly as ivit_layernorm.h emits it (-DIVIT_PROBE_LN192_S1=1; built WITH packed fp32
// unless PK=0 passes -Xclang -target-feature -Xclang -packed-fp32-ops), the aggressor is gemm_glds_kernel<EPI_RQ8_CH, 128> itself
// (ivit_gemm2.h) on the eight shapes of tools/op_stress.py, in the rotation of its mixed run: every stream walks (victim, 8 GEMMs)
// rotated by three per stream, all streams at once; every victim output is compared with the single-stream result.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-pass-failed -DIVIT_PROBE_LN192_S1=1 tools/ubench/ln_s1_standalone.hip -o tools/ubench/ln_s1_standalone_pk
//        ... the same + -Xclang -target-feature -Xclang -packed-fp32-ops                                                       -o tools/ubench/ln_s1_standalone_nopk
#include "../../i-vit_amd/csrc/ivit_device.h"
#include "../../i-vit_amd/csrc/ivit_elementwise.h"
#include "../../i-vit_amd/csrc/ivit_layernorm.h"
#include "../../i-vit_amd/csrc/ivit_gemm.h"
#include "../../i-vit_amd/csrc/ivit_gemm2.h"
#include "f64_aggressors.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// gemm_nt_kernel<false, EPI_RQ8_CH> taken apart (argv[3] = 200 + FLAGS), same grid / block / 34 KB static LDS:
//   1: the main loop (global -> registers -> ds_write_b128 -> ds_read_b128 -> 32x32x32 MFMA); without it the accumulators are lane ids
//   2: the double-precision requant (v_cvt_f64_i32, v_mul_f64 x 2, v_rndne_f64, clamp, v_cvt_i32_f64); without it an integer shift
//   8 / 16 / 32 / 64 (with 1): the main loop WITHOUT its global loads / its LDS staging / its MFMAs (integer adds instead) / its barriers
//   4: the staged write-out (ds_write_b8 per value, barrier, ds_read_b128, 16-byte global stores); without it direct byte stores
template <int FLAGS>
__global__ __launch_bounds__(256) void nt_like(GemmArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[GEMM_SMEM];
    char *sA = smem, *sB = smem + 8192;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int tile_m = blockIdx.x / p.tiles_n, tile_n = blockIdx.x % p.tiles_n, row0 = tile_m * GEMM_BM, col0 = tile_n * GEMM_BN;
    const int8_t *A8 = reinterpret_cast<const int8_t *>(p.A);
    v16i acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = (FLAGS & 1) ? 0 : (lane * 97 + r * 1531 + i * 7 + j * 13 + (int)blockIdx.x) % 60001 - 30000;
    if (FLAGS & 1) {
        const int nk = (p.K + GEMM_BK - 1) / GEMM_BK;
        for (int kt = 0; kt < nk; ++kt) {
            v4i ra[2], rb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                int id = tid + i * 256, row = id >> 2, c = id & 3;
                if (FLAGS & 8) { rb[i] = v4i{id, row, c, kt}; ra[i] = v4i{kt, c, id, row}; continue; }
                rb[i] = load_chunk_i8(p.B, p.ldb, col0 + row, p.N, kt * GEMM_BK + c * 16, p.K);
                ra[i] = load_chunk_i8(A8, p.lda, row0 + row, p.M, kt * GEMM_BK + c * 16, p.K);
            }
            if (!(FLAGS & 16)) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    int id = tid + i * 256, row = id >> 2, c = id & 3;
                    *reinterpret_cast<v4i *>(sB + lds_off(row, c)) = rb[i];
                    *reinterpret_cast<v4i *>(sA + lds_off(row, c)) = ra[i];
                }
            }
            if (!(FLAGS & 64)) __syncthreads();
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int chunk = kk * 2 + (lane >> 5);
                v4i a[2], b[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    if (FLAGS & 16) { a[i] = ra[i] + kk; b[i] = rb[i] - kk; continue; }
                    a[i] = *reinterpret_cast<const v4i *>(sA + lds_off(wm * 64 + i * 32 + (lane & 31), chunk));
                    b[i] = *reinterpret_cast<const v4i *>(sB + lds_off(wn * 64 + i * 32 + (lane & 31), chunk));
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        if (FLAGS & 32) { for (int r = 0; r < 16; ++r) acc[i][j][r] += a[i][r & 3] ^ b[j][(r >> 2) & 3]; }
                        else acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i], b[j], acc[i][j], 0, 0, 0);
                    }
            }
            if (!(FLAGS & 64)) __syncthreads();
        }
    }
    double dm[2] = {0, 0}, dr[2] = {0, 0};
    int bias[2] = {0, 0};
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        int col = col0 + wn * 64 + j * 32 + (lane & 31);
        if (col < p.N) { bias[j] = p.bias[col]; dm[j] = p.dy_ch[col].m; dr[j] = p.dy_ch[col].r; }
    }
    int8_t *out = reinterpret_cast<int8_t *>(p.out);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int rl = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), cl = wn * 64 + j * 32 + (lane & 31);
                int v = acc[i][j][r] + bias[j];
                int o = (FLAGS & 2) ? clamp_b<8>(rq_f64((double)v, dm[j], dr[j])) : max(-128, min(127, v >> 9));
                if (FLAGS & 4) *reinterpret_cast<int8_t *>(smem + rl * GEMM_SC8_LD + cl) = (int8_t)o;
                else if (row0 + rl < p.M && col0 + cl < p.N) out[(long long)(row0 + rl) * p.ldc + col0 + cl] = (int8_t)o;
            }
    if (!(FLAGS & 4)) return;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int id = tid + i * 256, row = id >> 3, c = id & 7, grow = row0 + row, gcol = col0 + c * 16;
        if (grow < p.M && gcol + 16 <= p.N) *reinterpret_cast<v4i *>(out + (long long)grow * p.ldc + gcol) = *reinterpret_cast<const v4i *>(smem + row * GEMM_SC8_LD + c * 16);
    }
}

static ivit_dyadic dyadic(double s_pre, double s_out) {          // m * 2^-e ~ s_pre / s_out with a 31-bit mantissa (freeze.dyadic)
    int e;
    const double f = frexp(s_pre / s_out, &e);
    ivit_dyadic d;
    d.m = floor(f * 2147483648.0 + 0.5);
    d.r = ldexp(1.0, e - 31);
    return d;
}

int main(int argc, char **argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 200, NS = argc > 2 ? atoi(argv[2]) : 8, REP = 4;
    constexpr int C = 192;
    const long long rows = 25088;
    srand(3);
    std::vector<int16_t> hx(rows * C);
    for (auto &v : hx) v = (int16_t)(rand() % 40001 - 20000);
    std::vector<float> hb(C), hs(C);
    std::vector<ivit_dyadic> hd(C);
    for (int c = 0; c < C; ++c) {
        hb[c] = (float)((rand() % 2000001 - 1000000) * 0.3);
        hs[c] = (float)(pow(10.0, -10.2 + 0.4 * (rand() % 1000) / 1000.0));
        hd[c] = dyadic((double)hs[c], 0.03);
    }
    int16_t *x; float *bi, *sc; ivit_dyadic *dy; int8_t *out, *ref;
    CK(hipMalloc(&x, hx.size() * 2)); CK(hipMalloc(&bi, C * 4)); CK(hipMalloc(&sc, C * 4)); CK(hipMalloc(&dy, C * sizeof(ivit_dyadic)));
    CK(hipMalloc(&out, rows * C)); CK(hipMalloc(&ref, rows * C));
    CK(hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(bi, hb.data(), C * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(sc, hs.data(), C * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dy, hd.data(), C * sizeof(ivit_dyadic), hipMemcpyHostToDevice));
    constexpr int S = 1, rpb = (LNR_THREADS(S) / 64) * (64 / (4 * S));
    const unsigned lgrid = (unsigned)((rows + rpb - 1) / rpb);
    // VICTIM_CO=<code object>: the victim is taken from there (tools/ubench/ln_s1_asm: the same kernel reassembled with classes of
    // its packed fp32 instructions unpacked) instead of the one compiled into this binary
    hipFunction_t vfun = nullptr;
    if (const char *co = getenv("VICTIM_CO")) {
        hipModule_t mod;
        CK(hipModuleLoad(&mod, co));
        CK(hipModuleGetFunction(&vfun, mod, "_Z20layernorm_reg_kernelILi192ELi1EEvPKsxxfPKfS3_PK11ivit_dyadicPa"));
        printf("victim from %s\n", co);
    }
    auto victim = [&](int8_t *o, hipStream_t st) {
        if (!vfun) { layernorm_reg_kernel<C, S><<<lgrid, LNR_THREADS(S), 0, st>>>(x, rows, (long long)C, 0.01f, bi, sc, dy, o); return; }
        long long a_rows = rows, a_stride = C;
        float a_s = 0.01f;
        void *args[] = {&x, &a_rows, &a_stride, &a_s, &bi, &sc, &dy, &o};
        if (hipModuleLaunchKernel(vfun, lgrid, 1, 1, LNR_THREADS(S), 1, 1, 0, st, args, nullptr) != hipSuccess) { printf("module launch failed\n"); exit(1); }
    };

    // aggressors: the launch-per-tile QuantLinear kernel on the shapes of tools/op_stress.py's Swin slice — 8-bit requant and the
    // 16-bit + identity flavour, both tile heights as the library's dispatcher would pick them
    struct Gemm { GemmArgs a; unsigned grid; int res, bm; };
    std::vector<Gemm> gemms;
    // the ninth entry is the patch-embedding shape (K = 48: not a multiple of 32, so the library runs it on gemm_nt_kernel, the
    // generic register-staged kernel of ivit_gemm.h) — also part of the mixed run that fails through the library
    const int shapes[9][4] = {{100352, 288, 96, 0}, {100352, 96, 96, 1}, {6272, 1152, 384, 0}, {6272, 384, 384, 1},
                              {25088, 576, 192, 0}, {25088, 192, 192, 1}, {1568, 2304, 768, 0}, {1568, 768, 768, 1}, {100352, 96, 48, 0}};
    for (auto &sh : shapes) {
        const int M = sh[0], N = sh[1], K = sh[2], res = sh[3];
        std::vector<int8_t> ha((size_t)M * K), hw((size_t)N * K);
        for (auto &v : ha) v = (int8_t)(rand() % 256 - 128);
        for (auto &v : hw) v = (int8_t)(rand() % 256 - 128);
        std::vector<int> hbias(N);
        std::vector<ivit_dyadic> hdy(N);
        for (int n = 0; n < N; ++n) { hbias[n] = rand() % 6001 - 3000; hdy[n] = dyadic(pow(10.0, (res ? -5.9 : -5.6) + 0.4 * (rand() % 1000) / 1000.0), res ? 2e-4 : 0.012); }
        std::vector<int16_t> hres((size_t)M * N);
        for (auto &v : hres) v = (int16_t)(rand() % 60001 - 30000);
        int8_t *A, *W; void *O; int *B; ivit_dyadic *D; int16_t *R;
        CK(hipMalloc(&A, ha.size())); CK(hipMalloc(&W, hw.size())); CK(hipMalloc(&O, (size_t)M * N * 2)); CK(hipMalloc(&B, N * 4)); CK(hipMalloc(&D, N * sizeof(ivit_dyadic)));
        CK(hipMalloc(&R, hres.size() * 2));
        CK(hipMemcpy(A, ha.data(), ha.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(W, hw.data(), hw.size(), hipMemcpyHostToDevice));
        CK(hipMemcpy(B, hbias.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(D, hdy.data(), N * sizeof(ivit_dyadic), hipMemcpyHostToDevice));
        CK(hipMemcpy(R, hres.data(), hres.size() * 2, hipMemcpyHostToDevice));
        Gemm g;
        memset(&g.a, 0, sizeof(g.a));
        g.a.A = A; g.a.B = W; g.a.M = M; g.a.N = N; g.a.K = K; g.a.lda = K; g.a.ldb = K; g.a.ldc = N; g.a.inner = 1; g.a.bias = B;
        g.a.out = O; g.a.dy_ch = D; g.a.tiles_n = (N + G2_BN - 1) / G2_BN; g.a.residual = R;
        g.a.dy_main = dyadic(2e-4, 3.1e-4); g.a.dy_res = dyadic(2.7e-4, 3.1e-4);
        const long long t256 = (long long)((M + 255) / 256) * g.a.tiles_n, t128 = (long long)((M + 127) / 128) * g.a.tiles_n;
        const long long c256 = ((t256 + 511) / 512) * 256, c128 = ((t128 + 767) / 768) * 128;
        g.bm = c128 < c256 ? 128 : 256;
        g.grid = (unsigned)(g.bm == 128 ? t128 : t256);
        if (K % 32) { g.bm = 0; g.a.tiles_n = (N + GEMM_BN - 1) / GEMM_BN; g.grid = (unsigned)(((M + GEMM_BM - 1) / GEMM_BM) * g.a.tiles_n); }
        g.res = res;
        gemms.push_back(g);
    }
    const int only = argc > 3 ? atoi(argv[3]) : -1;               // 0..8: only this aggressor (index into the shape list);
    double *sink;                                                 // 100 + k: only the one-instruction-class kernel k of f64_aggressors.h
    CK(hipMalloc(&sink, 1024 * 256 * 8));
    auto aggressor = [&](const Gemm &g0, hipStream_t s) {
        if (only >= 200) {
            const Gemm &g = gemms[8];
            switch (only - 200) {
#define NT_CASE(F) case F: nt_like<F><<<g.grid, 256, 0, s>>>(g.a); break;
                NT_CASE(0) NT_CASE(1) NT_CASE(2) NT_CASE(3) NT_CASE(4) NT_CASE(5) NT_CASE(6) NT_CASE(7)
                NT_CASE(9) NT_CASE(17) NT_CASE(33) NT_CASE(65) NT_CASE(25) NT_CASE(41) NT_CASE(49) NT_CASE(57) NT_CASE(121)
#undef NT_CASE
            }
            return;
        }
        if (only >= 100) { launch_f64_aggressor(only - 100, sink, 1024, 1500, s); return; }
        const Gemm &g = only >= 0 ? gemms[only] : g0;
        if (g.bm == 0) gemm_nt_kernel<false, EPI_RQ8_CH><<<dim3(g.grid, 1, 1), 256, 0, s>>>(g.a);
        else if (g.res) { if (g.bm == 128) gemm_glds_kernel<EPI_RQ16_CH_RES, 128><<<g.grid, 256, 0, s>>>(g.a); else gemm_glds_kernel<EPI_RQ16_CH_RES, 256><<<g.grid, 512, 0, s>>>(g.a); }
        else { if (g.bm == 128) gemm_glds_kernel<EPI_RQ8_CH, 128><<<g.grid, 256, 0, s>>>(g.a); else gemm_glds_kernel<EPI_RQ8_CH, 256><<<g.grid, 512, 0, s>>>(g.a); }
    };
    std::vector<hipStream_t> st(NS);
    for (auto &s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    std::vector<int8_t> href(rows * C), hout(rows * C);
    victim(ref, st[0]);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(href.data(), ref, rows * C, hipMemcpyDeviceToHost));
    if (vfun) {                                                  // the reassembled victim must compute what the compiled-in one does
        layernorm_reg_kernel<C, S><<<lgrid, LNR_THREADS(S), 0, st[0]>>>(x, rows, (long long)C, 0.01f, bi, sc, dy, out);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(hout.data(), out, rows * C, hipMemcpyDeviceToHost));
        printf("code-object victim %s the compiled-in victim\n", memcmp(hout.data(), href.data(), rows * C) ? "DIFFERS FROM" : "matches");
    }
    std::vector<int8_t *> outs(NS);
    for (auto &o : outs) CK(hipMalloc(&o, rows * C));
    // sanity: alone, repeatedly
    int bad_alone = 0;
    for (int r = 0; r < 20; ++r) {
        victim(out, st[0]);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(hout.data(), out, rows * C, hipMemcpyDeviceToHost));
        bad_alone += memcmp(hout.data(), href.data(), rows * C) != 0;
    }
    // the mixed run of tools/op_stress.py: every stream walks the operator list (victim + 8 GEMM launches) rotated by 3 per stream,
    // all streams at once, REP times a round; every victim output of the round is checked
    const int nops = 1 + (int)gemms.size();
    int bad = 0, launches = 0;
    long long bad_elems = 0;
    for (int r = 0; r < rounds; ++r) {
        for (int k = 0; k < REP; ++k) {
            for (auto &o : outs) CK(hipMemsetAsync(o, 0x55, rows * C, 0));
            CK(hipDeviceSynchronize());
            for (int j = 0; j < nops; ++j)
                for (int s = 0; s < NS; ++s) {
                    const int op = (j + 3 * s) % nops;
                    if (op == 0) victim(outs[s], st[s]);
                    else aggressor(gemms[op - 1], st[s]);
                }
            CK(hipDeviceSynchronize());
            for (int s = 0; s < NS; ++s) {
                CK(hipMemcpy(hout.data(), outs[s], rows * C, hipMemcpyDeviceToHost));
                ++launches;
                if (memcmp(hout.data(), href.data(), rows * C) != 0) {
                    ++bad;
                    long long n = 0, first = -1;
                    for (long long i = 0; i < rows * C; ++i) if (hout[i] != href[i]) { if (first < 0) first = i; ++n; }
                    bad_elems += n;
                    if (bad <= 5) printf("  round %d stream %d: %lld elements differ, first at row %lld col %lld: %d vs %d\n", r, s, n, first / C, first % C, hout[first], href[first]);
                }
            }
        }
    }
    if (only >= 200) printf("aggressor nt_like<%d> only (1 main loop, 2 f64 requant, 4 staged write-out; main loop without 8 global loads, 16 LDS, 32 MFMA, 64 barriers): ", only - 200);
    else if (only >= 100) printf("aggressor %s only: ", f64_aggr_names[only - 100]);
    else if (only >= 0) printf("aggressor %d only (M %d N %d K %d%s): ", only, gemms[only].a.M, gemms[only].a.N, gemms[only].a.K, gemms[only].bm ? "" : ", gemm_nt_kernel");
    printf("layernorm_reg_kernel<192, 1> beside its aggressors on %d streams: %d of %d victim launches differ from the single-stream result (%lld elements); alone: %d of 20\n",
           NS, bad, launches, bad_elems, bad_alone);
    return 0;
}
