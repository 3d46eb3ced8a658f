// In-wave MFMA / VALU overlap probe for gfx950: how much requant-style VALU work (cvt_f64_i32, fma_f64,
// med3, add per element) hides under a stream of v_mfma_i32_32x32x32_i8 issued by the SAME wave, at
// 1 / 2 / 4 waves per SIMD.  Decides whether a GEMM wave can run tile i's epilogue under tile i+1's K loop.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/overlap.hip -o tools/ubench/overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// NE: requant elements per MFMA (x4 VALU instructions each); MF: 1 = with MFMAs, 0 = VALU only
template <int NE, int MF, int WPS>
__global__ __launch_bounds__(256, WPS) void k_overlap(int *out, int n, int seed, double cc) {
    v4i a[2], b[2];
    for (int i = 0; i < 2; ++i)
        for (int e = 0; e < 4; ++e) {
            unsigned x = (threadIdx.x * 2654435761u + i * 40503u + e * 977u) * (unsigned)seed;
            a[i][e] = (int)x;
            b[i][e] = (int)(x * 2246822519u);
        }
    v16i c[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) c[i][r] = 0;
    int z[16];
    for (int i = 0; i < 16; ++i) z[i] = threadIdx.x * 31 + i * 1001;
    double cm[4] = {cc, cc * 1.25, cc * 1.5, cc * 1.75};
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (MF) c[m & 3] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[m & 1], b[(m >> 1) & 1], c[m & 3], 0, 0, 0);
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const int g = (m * NE + e) & 15;
                double t = __builtin_fma((double)z[g], cm[e & 3], 6755399441055744.0);
                int v = __double2loint(t);
                v = min(max(v, -128), 127);
                z[g] = v + 77 * g + it;
            }
            if (MF) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (NE) __builtin_amdgcn_sched_group_barrier(0x002, NE * 4, 0);
        }
    }
    int s = 0;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += c[i][r];
    for (int i = 0; i < 16; ++i) s ^= z[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NE, int MF, int WPS>
void run(int *buf) {
    const int n = 2048;
    dim3 grid(256 * WPS), block(256);   // WPS blocks of 4 waves per CU = WPS waves per SIMD
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k_overlap<NE, MF, WPS>), grid, block, 0, 0, buf, 64, 12345, 1.0e-3);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((k_overlap<NE, MF, WPS>), grid, block, 0, 0, buf, n, 12345, 1.0e-3);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    // per SIMD: WPS waves x n x 8 slots
    double slots = (double)WPS * n * 8;
    double cyc = ms * 1e-3 * 2.4e9 / slots;
    printf("waves/SIMD %d  %s  elems/MFMA %d : %8.3f ms  %6.1f cyc per (MFMA + %d elems) per SIMD", WPS,
           MF ? "mfma+valu" : "valu only", NE, ms, cyc, NE);
    if (MF) printf("  -> %6.0f TOP/s", (double)grid.x * 4 * n * 8 * 65536.0 / ms / 1e9);
    if (NE) printf("  (%.1f cyc per element)", cyc / NE);
    printf("\n");
}

// ---- LDS byte-gather rate: random ds_read_u8 over a table of TB bytes, 4 waves/SIMD
template <int TB>
__global__ __launch_bounds__(256) void k_gather(int *out, int n) {
    __shared__ unsigned char tbl[TB];
    for (int i = threadIdx.x; i < TB; i += 256) tbl[i] = (unsigned char)(i * 7 + 3);
    __syncthreads();
    unsigned idx[8];
    for (int k = 0; k < 8; ++k) idx[k] = (threadIdx.x * 2654435761u + k * 40503u) >> 8;
    unsigned s = 0;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            unsigned v = tbl[idx[k] & (TB - 1)];
            s += v;
            idx[k] = idx[k] * 1664525u + 1013904223u;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (int)s;
}
template <int TB>
void run_gather(int *buf, int bpc) {
    const int n = 2048;
    dim3 grid(256 * bpc), block(256);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k_gather<TB>), grid, block, 0, 0, buf, 16);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((k_gather<TB>), grid, block, 0, 0, buf, n);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    double gathers_per_cu = (double)bpc * 256 * n * 8;
    printf("ds_read_u8 random over %6d B, %d blocks/CU: %8.3f ms  %.2f byte-gathers per clk per CU (%.2f cyc per wave-instr per CU)\n",
           TB, bpc, ms, gathers_per_cu / (ms * 1e-3 * 2.4e9), ms * 1e-3 * 2.4e9 / (gathers_per_cu / 64));
}

int main() {
    int *buf;
    hipMalloc(&buf, 256 * 8 * 256 * 4);
    run<0, 1, 1>(buf); run<1, 1, 1>(buf); run<2, 1, 1>(buf); run<3, 1, 1>(buf); run<4, 1, 1>(buf);
    run<1, 0, 1>(buf); run<2, 0, 1>(buf); run<4, 0, 1>(buf);
    run<0, 1, 2>(buf); run<1, 1, 2>(buf); run<2, 1, 2>(buf); run<3, 1, 2>(buf); run<4, 1, 2>(buf);
    run<1, 0, 2>(buf); run<2, 0, 2>(buf); run<4, 0, 2>(buf);
    run<0, 1, 4>(buf); run<1, 1, 4>(buf); run<2, 1, 4>(buf); run<4, 1, 4>(buf);
    run<2, 0, 4>(buf); run<4, 0, 4>(buf);
    run_gather<256>(buf, 4); run_gather<65536>(buf, 2); run_gather<256>(buf, 2);
    return 0;
}
