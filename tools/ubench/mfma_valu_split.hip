// mfma_valu_split.hip — do one wave's MFMAs run beside ANOTHER wave's VALU on the same SIMD?  One 8-wave workgroup per CU (LDS-limited):
// waves 0-3 (one per SIMD) issue dense v_mfma_i32_32x32x32_i8 (4 independent accumulators), waves 4-7 a requant-like VALU stream
// (v_cvt_f64_i32, v_fma_f64, packs).  Three runs: MFMA waves alone, VALU waves alone, both; cycles per instruction from s_memtime.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_valu_split.hip -o tools/ubench/mfma_valu_split
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512) void k(int mode, int n_mfma, int n_valu, long long *cyc, int *sink) {
    extern __shared__ char sm[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    long long t0 = 0, t1 = 0;
    if (wave < 4) {
        if (!(mode & 1)) return;
        v4i a, b;
        for (int e = 0; e < 4; ++e) { a[e] = (lane * 2654435761u + e * 977u) ^ 0x5a5a5a5a; b[e] = a[e] * 40503u; }
        v16i c0 = {}, c1 = {}, c2 = {}, c3 = {};
        __syncthreads();
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < n_mfma; ++i) {
            c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(b, a, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, a, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(b, b, c3, 0, 0, 0);
        }
        t1 = __builtin_readcyclecounter();
        int s = 0;
        for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
        if (s == 0x12345) sink[0] = s;
    } else {
        if (!(mode & 2)) { if (mode & 1) __syncthreads(); return; }
        int z[8];
        for (int e = 0; e < 8; ++e) z[e] = lane * 7919 + e * 104729;
        const double c = 1.0e-4 + lane * 1e-9;
        if (mode & 1) __syncthreads();
        t0 = __builtin_readcyclecounter();
        unsigned acc = 0;
        for (int i = 0; i < n_valu; ++i) {
            // 8 outputs: 8 cvt + 8 fma + 4 cvt_pk + 4 sat_pk + 2 perm + 2 xor = 28 VALU instructions
            int o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                double t = __builtin_fma((double)z[e], c, 6755399441055744.0 + 128.0);
                o[e] = __double2loint(t);
                z[e] += o[e] | 1;
            }
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                unsigned p01, p23, b01, b23;
                asm volatile("v_cvt_pk_i16_i32 %0, %1, %2" : "=v"(p01) : "v"(o[4 * g]), "v"(o[4 * g + 1]));
                asm volatile("v_cvt_pk_i16_i32 %0, %1, %2" : "=v"(p23) : "v"(o[4 * g + 2]), "v"(o[4 * g + 3]));
                asm volatile("v_sat_pk_u8_i16 %0, %1" : "=v"(b01) : "v"(p01));
                asm volatile("v_sat_pk_u8_i16 %0, %1" : "=v"(b23) : "v"(p23));
                acc += __builtin_amdgcn_perm(b23, b01, 0x05040100u) ^ 0x80808080u;
            }
        }
        t1 = __builtin_readcyclecounter();
        if (acc == 0x12345) sink[1] = acc;
    }
    if (lane == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}

int main() {
    long long *cyc; int *sink;
    hipMalloc(&cyc, 64); hipMalloc(&sink, 64);
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    const int n_mfma = 4096, n_valu = 4096;      // 16384 MFMAs; 4096 x 36 VALU (28 + 8 bookkeeping adds/ors)
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 1; mode <= 3; ++mode) {
            hipMemset(cyc, 0, 64);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            k<<<256, 512, 100 * 1024>>>(mode, n_mfma, n_valu, cyc, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long h[8]; hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
            printf("mode %d (%s): %.3f ms | MFMA wave 0: %lld cyc = %.1f per MFMA | VALU wave 4: %lld cyc = %.1f per 8-output group (28 requant instr + 16 bookkeeping)\n", mode,
                   mode == 1 ? "MFMA waves only" : mode == 2 ? "VALU waves only" : "both", ms, h[0], (double)h[0] / (4.0 * n_mfma), h[4], (double)h[4] / n_valu);
        }
    return 0;
}
