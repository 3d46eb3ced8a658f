// The fused Mlp's weight stream in isolation: 12 waves per CU, every CU reads the same 576 KB buffer, wave w takes fragment
// s * 12 + w of step s (ONE contiguous 12 KB window per step and CU), DEPTH loads in flight per wave; optionally each
// fragment feeds two 32x32x32 MFMAs (MF = 1) the way the kernel's K loop does.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/l2_window.hip -o tools/ubench/l2_window
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
// k_kloop: the kernel's K loop proper — weight fragment from L2 (DEPTH ahead), two activation fragments from a K-blocked LDS
// image (BD ahead; PITCH bytes between tokens), two MFMAs
template <int DEPTH, int BD, int PITCH, int LD>
__global__ __launch_bounds__(768, 1) void k_kloop(const v4i *buf, int reps, int *out) {
    extern __shared__ __attribute__((aligned(256))) char sm[];
    constexpr int KBLK = 64 * PITCH + 64, STEPS = 48;
    for (int i = threadIdx.x; i < 24 * KBLK / 4; i += 768) reinterpret_cast<int *>(sm)[i] = i * 2654435761u;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const v4i *p = buf + (size_t)wave * 64 + lane;
    const unsigned fb = (lane & 31) * PITCH + (lane >> 5) * 16;
    v16i c0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, c1 = c0;
    for (int r = 0; r < reps; ++r) {
        v4i v[DEPTH + 1], b[BD + 1][2];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) v[d] = p[(size_t)d * 12 * 64];
#pragma unroll
        for (int d = 0; d < BD; ++d) {
            b[d][0] = *reinterpret_cast<const v4i *>(sm + (d >> 1) * KBLK + (d & 1) * 32 + fb);
            b[d][1] = *reinterpret_cast<const v4i *>(sm + (d >> 1) * KBLK + 32 * PITCH + (d & 1) * 32 + fb);
        }
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            __builtin_amdgcn_sched_barrier(0);
            if (s + DEPTH < STEPS) v[(s + DEPTH) % (DEPTH + 1)] = p[(size_t)(s + DEPTH) * 12 * 64];
            if (LD && s + BD < STEPS) {
                const int d = s + BD;
                b[d % (BD + 1)][0] = *reinterpret_cast<const v4i *>(sm + (d >> 1) * KBLK + (d & 1) * 32 + fb);
                b[d % (BD + 1)][1] = *reinterpret_cast<const v4i *>(sm + (d >> 1) * KBLK + 32 * PITCH + (d & 1) * 32 + fb);
            }
            __builtin_amdgcn_sched_barrier(0);
            c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(v[s % (DEPTH + 1)], b[LD ? s % (BD + 1) : 0][0], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(v[s % (DEPTH + 1)], b[LD ? s % (BD + 1) : 0][1], c1, 0, 0, 0);
        }
    }
    int acc = 0;
    for (int e = 0; e < 16; ++e) acc ^= c0[e] ^ c1[e];
    out[blockIdx.x * 768 + threadIdx.x] = acc;
}
template <int DEPTH, int BD, int PITCH, int LD>
void runk(const v4i *buf, int *out) {
    const int reps = 64, smem = 24 * (64 * PITCH + 64);
    hipFuncSetAttribute((const void *)k_kloop<DEPTH, BD, PITCH, LD>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k_kloop<DEPTH, BD, PITCH, LD><<<256, 768, smem>>>(buf, 2, out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k_kloop<DEPTH, BD, PITCH, LD><<<256, 768, smem>>>(buf, reps, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    printf("K loop: %2d weight loads ahead, LDS fragments %s (%d ahead, pitch %d): %6.1f us per 48-step pass (MFMA alone ~5.8 us)\n", DEPTH,
           LD ? "read" : "NOT read", BD, PITCH, ms * 1e3 / reps);
}

template <int DEPTH, int MF, int WIN>
__global__ __launch_bounds__(768, 1) void k_win(const v4i *buf, int steps, int reps, int *out, int cold) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // WIN = 1: window order (fragment s * 12 + wave); WIN = 0: each wave streams its own contiguous twelfth
    const v4i *p = WIN ? buf + (size_t)wave * 64 + lane : buf + (size_t)wave * steps * 64 + lane;
    const size_t stride = WIN ? 12 * 64 : 64;
    v16i c0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, c1 = c0;
    v4i x = {lane, wave, 3, 4};
    for (int r = 0; r < reps; ++r) {
        v4i v[DEPTH];
        if (cold) p += (size_t)steps * 12 * 64;            // the next 576 KB of a buffer larger than the L2s: every pass misses
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) v[d] = p[(size_t)d * stride];
        for (int s = 0; s < steps; s += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const v4i cur = v[d];
                if (s + DEPTH + d < steps) v[d] = p[(size_t)(s + DEPTH + d) * stride];
                if (MF) {
                    c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(cur, x, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(cur, x, c1, 0, 0, 0);
                } else {
                    x ^= cur;
                }
            }
        }
    }
    int acc = x[0] ^ x[1] ^ x[2] ^ x[3];
    for (int e = 0; e < 16; ++e) acc ^= c0[e] ^ c1[e];
    out[blockIdx.x * 768 + threadIdx.x] = acc;
}
template <int DEPTH, int MF, int WIN>
void run(const v4i *buf, int *out, int cold = 0) {
    const int steps = 48, reps = 64;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k_win<DEPTH, MF, WIN><<<256, 768>>>(buf, steps, 2, out, cold);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k_win<DEPTH, MF, WIN><<<256, 768>>>(buf, steps, reps, out, cold);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double bytes = 256.0 * reps * steps * 12 * 1024.0;
    printf("%s%s order, %2d loads in flight per wave, %s: %6.2f TB/s = %5.1f B/clk/CU at 2.1 GHz; %6.1f us per 576 KB pass%s\n", cold ? "COLD (every pass misses the L2s) " : "", WIN ? "window" : "stream",
           DEPTH, MF ? "2 MFMAs per fragment" : "xor only", bytes / (ms * 1e-3) / 1e12, bytes / (ms * 1e-3) / 256 / 2.1e9, ms * 1e3 / reps,
           MF ? " (MFMA alone: 288 per SIMD x ~20 ns = 5.8 us)" : "");
}
int main() {
    v4i *buf; int *out;
    hipMalloc(&buf, 48 << 20); hipMemset(buf, 1, 48 << 20); hipMalloc(&out, 256 * 768 * 4);
    run<6, 0, 1>(buf, out, 1); run<12, 0, 1>(buf, out, 1); run<24, 0, 1>(buf, out, 1); run<6, 1, 1>(buf, out, 1); run<12, 1, 1>(buf, out, 1); run<24, 1, 1>(buf, out, 1);
    runk<6, 3, 80, 0>(buf, out); runk<6, 3, 80, 1>(buf, out); runk<12, 3, 80, 1>(buf, out); runk<12, 6, 80, 1>(buf, out);
    runk<12, 3, 80, 0>(buf, out); runk<6, 1, 80, 1>(buf, out);
    run<6, 0, 1>(buf, out); run<6, 0, 0>(buf, out); run<12, 0, 1>(buf, out);
    run<6, 1, 1>(buf, out); run<6, 1, 0>(buf, out); run<12, 1, 1>(buf, out); run<3, 1, 1>(buf, out);
    return 0;
}
