// Sustained int8 MFMA rate probe for gfx950: register-only MFMA loops (no LDS, no memory),
// zero and random operands, 1..4 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_peak.hip -o tools/ubench/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int SHAPE>
__global__ __launch_bounds__(256) void k_mfma(int *out, int n, int seed) {
    v4i a[4], b[4];
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 4; ++e) {
            unsigned x = (threadIdx.x * 2654435761u + i * 40503u + e * 977u) * (unsigned)seed;
            a[i][e] = (int)x;
            b[i][e] = (int)(x * 2246822519u);
        }
    if (SHAPE == 32) {
        v16i c[4];
        for (int i = 0; i < 4; ++i)
            for (int r = 0; r < 16; ++r) c[i][r] = 0;
        for (int it = 0; it < n; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i], b[i], c[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(b[i], a[i], c[i], 0, 0, 0);
        }
        int s = 0;
        for (int i = 0; i < 4; ++i)
            for (int r = 0; r < 16; ++r) s += c[i][r];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    } else {
        v4i c[8];
        for (int i = 0; i < 8; ++i)
            for (int r = 0; r < 4; ++r) c[i][r] = 0;
        for (int it = 0; it < n; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[i & 3], b[i & 3], c[i], 0, 0, 0);
        }
        int s = 0;
        for (int i = 0; i < 8; ++i)
            for (int r = 0; r < 4; ++r) s += c[i][r];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    }
}

template <int SHAPE>
void run(int *buf, int blocks_per_cu, int seed) {
    const int n = 4096;
    dim3 grid(256 * blocks_per_cu), block(256);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k_mfma<SHAPE>, grid, block, 0, 0, buf, 64, seed);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k_mfma<SHAPE>, grid, block, 0, 0, buf, n, seed);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    // per wave per iteration: 8 MFMAs; 32x32x32 = 2*32*32*32 = 65536 OP, 16x16x64 = 2*16*16*64 = 32768 OP
    const double op_per_mfma = SHAPE == 32 ? 65536.0 : 32768.0;
    double ops = (double)grid.x * 4 * n * 8 * op_per_mfma;
    printf("mfma %s  %d waves/SIMD  %-6s operands: %8.3f ms  %7.1f TOP/s  (%.1f cyc/MFMA/SIMD at 2.4 GHz)\n",
           SHAPE == 32 ? "32x32x32" : "16x16x64", blocks_per_cu, seed ? "random" : "zero", ms, ops / ms / 1e9,
           ms * 1e-3 * 2.4e9 / ((double)blocks_per_cu * n * 8));
}

int main() {
    int *buf;
    hipMalloc(&buf, 256 * 8 * 256 * 4);
    for (int seed = 0; seed < 2; ++seed)
        for (int bpc = 1; bpc <= 4; bpc *= 2) {
            run<32>(buf, bpc, seed);
            run<16>(buf, bpc, seed);
        }
    // long sustained run (thermal/power state)
    for (int r = 0; r < 3; ++r) run<32>(buf, 4, 1);
    return 0;
}
