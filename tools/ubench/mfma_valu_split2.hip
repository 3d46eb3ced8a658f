// mfma_valu_split2.hip — mfma_valu_split.hip made to look like gemm_ws_qkv_kernel step by step: which ingredient takes the K loop from 32 to 38-45
// cycles per MFMA and the partner's epilogue from 5 to 9 cycles per VALU instruction?
//   MF bit 0: the MFMA wave's A operands are 24 different register quads (the slab), bit 1: its B operands come from LDS (two ds_read_b128 per
//   four MFMAs, one k-step ahead), bit 2: two k-steps ahead
//   VA bit 0: the VALU wave reads its multipliers from LDS (8 ds_read_b128 per 32 outputs), bit 1: stores 16 bytes per lane per 16 outputs
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_valu_split2.hip -o tools/ubench/mfma_valu_split2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef double v2d __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) v4i lds_v4i;
typedef __attribute__((address_space(3))) v2d lds_v2d;

template <int MF, int VA>
__global__ __launch_bounds__(512) void k(int mode, int n_mfma, int n_valu, long long *cyc, int *sink, v4i *out) {
    extern __shared__ char sm[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned sm_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char *)sm;
    for (int i = threadIdx.x; i < 24 * 1024; i += 512) reinterpret_cast<int *>(sm)[i] = i * 2654435761u;
    __syncthreads();
    long long t0 = 0, t1 = 0;
    if (wave < 4) {
        if (!(mode & 1)) return;
        v4i W[24], b[3][2];
        for (int f = 0; f < 24; ++f) for (int e = 0; e < 4; ++e) W[f][e] = (lane * 2654435761u + e * 977u + f * 31337u) ^ 0x5a5a5a5a;
        for (int s = 0; s < 3; ++s) for (int t = 0; t < 2; ++t) for (int e = 0; e < 4; ++e) b[s][t][e] = W[s * 2 + t][e] * 40503u;
        v16i c[2][2] = {};
        const unsigned base = sm_lds + (lane & 31) * 64 + (lane >> 5) * 16;
        constexpr int PF = (MF & 4) ? 2 : 1;
        __syncthreads();
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < n_mfma; i += 12) {
            if (MF & 2) for (int s = 0; s < PF; ++s) for (int t = 0; t < 2; ++t) b[s][t] = *(lds_v4i *)(size_t)(base + s * 4096 + t * 2048);
#pragma unroll
            for (int ks = 0; ks < 12; ++ks) {
                __builtin_amdgcn_sched_barrier(0);
                if ((MF & 2) && ks + PF < 12)
                    for (int t = 0; t < 2; ++t) b[(ks + PF) % (PF + 1)][t] = *(lds_v4i *)(size_t)(base + ((ks + PF) % 6) * 4096 + t * 2048);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                    for (int t = 0; t < 2; ++t)
                        c[cc][t] = __builtin_amdgcn_mfma_i32_32x32x32_i8((MF & 1) ? W[cc * 12 + ks] : W[cc], b[(MF & 2) ? ks % (PF + 1) : 0][t], c[cc][t], 0, 0, 0);
            }
        }
        t1 = __builtin_readcyclecounter();
        int s = 0;
        for (int r = 0; r < 16; ++r) s += c[0][0][r] + c[0][1][r] + c[1][0][r] + c[1][1][r];
        if (s == 0x12345) sink[0] = s;
    } else {
        if (!(mode & 2)) { if (mode & 1) __syncthreads(); return; }
        int z[16];
        for (int e = 0; e < 16; ++e) z[e] = lane * 7919 + e * 104729;
        v2d cq[8];
        for (int j = 0; j < 8; ++j) { cq[j][0] = 1.0e-4 + lane * 1e-9 + j * 1e-7; cq[j][1] = cq[j][0] * 1.01; }
        const unsigned cqa = sm_lds + 8192 + (lane >> 5) * 128;
        v4i *op = out + (size_t)(blockIdx.x * 512 + threadIdx.x);
        if (mode & 1) __syncthreads();
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < n_valu; ++i) {
            // one (channel tile, token tile) of the epilogue: 16 outputs -> 4 dwords: 16 cvt + 16 fma + 8 + 8 + 4 + 4 = 56 VALU
            if (VA & 1) for (int j = 0; j < 8; ++j) cq[j] = *(lds_v2d *)(size_t)(cqa + j * 16);
            v4i o4;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                int o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const double t = __builtin_fma((double)z[4 * q4 + e], cq[2 * q4 + (e >> 1)][e & 1], 6755399441055744.0 + 128.0);
                    o[e] = __double2loint(t);
                }
                unsigned p01, p23, b01, b23;
                asm("v_cvt_pk_i16_i32 %0, %1, %2" : "=v"(p01) : "v"(o[0]), "v"(o[1]));
                asm("v_cvt_pk_i16_i32 %0, %1, %2" : "=v"(p23) : "v"(o[2]), "v"(o[3]));
                asm("v_sat_pk_u8_i16 %0, %1" : "=v"(b01) : "v"(p01));
                asm("v_sat_pk_u8_i16 %0, %1" : "=v"(b23) : "v"(p23));
                int hq = (int)(__builtin_amdgcn_perm(b23, b01, 0x05040100u) ^ 0x80808080u);
                asm volatile("" : "+v"(hq));
                o4[q4] = hq;
                z[4 * q4] += hq | 1;
            }
            if (VA & 2) op[(size_t)(i & 7) * 256 * 512] = o4;
            else asm volatile("" ::"v"(o4));
        }
        t1 = __builtin_readcyclecounter();
    }
    if (lane == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}

template <int MF, int VA>
static void run(long long *cyc, int *sink, v4i *out) {
    hipFuncSetAttribute((const void *)k<MF, VA>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    const int n_mfma = 12 * 1024, n_valu = 6144;
    for (int mode = 1; mode <= 3; ++mode) {
        hipMemset(cyc, 0, 64);
        k<MF, VA><<<256, 512, 100 * 1024>>>(mode, n_mfma, n_valu, cyc, sink, out);
        hipDeviceSynchronize();
        long long h[8]; hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
        printf("MF %d VA %d mode %d (%s): %.1f cycles per MFMA | %.1f cycles per 16-output group (56 requant VALU + ~8)\n", MF, VA, mode,
               mode == 1 ? "MFMA waves only" : mode == 2 ? "VALU waves only" : "both           ", (double)h[0] / (4.0 * n_mfma), (double)h[4] / n_valu);
    }
}

int main() {
    long long *cyc; int *sink; v4i *out;
    hipMalloc(&cyc, 64); hipMalloc(&sink, 64); hipMalloc(&out, (size_t)8 * 256 * 512 * 16);
    run<0, 0>(cyc, sink, out); run<1, 0>(cyc, sink, out); run<3, 0>(cyc, sink, out); run<7, 0>(cyc, sink, out);
    run<3, 1>(cyc, sink, out); run<3, 2>(cyc, sink, out); run<3, 3>(cyc, sink, out);
    return 0;
}
