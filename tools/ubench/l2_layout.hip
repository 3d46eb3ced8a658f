// Which placement of the fused Mlp's weight fragments lets 256 CUs x 12 waves stream them fastest?  Every CU reads the same
// 48 steps x 12 fragments of 1 KB; fragment (step s, wave w) lives at byte offset map(s, w).  Straight-line 48-step pass,
// DEPTH loads in flight per wave, xor consumer.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/l2_layout.hip -o tools/ubench/l2_layout
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef int v4i __attribute__((ext_vector_type(4)));
template <int MAP>
__device__ __forceinline__ unsigned frag_off(int s, int w) {
    if (MAP == 0) return (unsigned)(s * 12 + w) * 1024u;                                  // one contiguous 12 KB window per step
    if (MAP == 1) return (unsigned)((s >> 2) * 16 + w) * 4096u + (unsigned)(s & 3) * 1024u;   // waves 4 KB apart
    if (MAP == 2) return (unsigned)w * (48u * 1024u + 256u) + (unsigned)s * 1024u;           // per-wave streams, skewed by 256 B
    if (MAP == 3) return (unsigned)((s >> 1) * 16 + w) * 2048u + (unsigned)(s & 1) * 1024u;   // waves 2 KB apart
    return (unsigned)(s * 16 + w) * 1024u + (unsigned)(w >> 2) * 0;                           // 16 KB window pitch
}
template <int DEPTH, int MAP>
__global__ __launch_bounds__(768, 1) void k(const char *buf, int reps, int *out, unsigned region) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(buf), 0, 0x7fffffff, 0x00020000);
    const unsigned loff = lane * 16;
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    v4u x = {0, 0, 0, 0};
    unsigned base = 0;
    for (int r = 0; r < reps; ++r) {
        v4u v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) v[d] = __builtin_amdgcn_raw_buffer_load_b128(rs, loff, base + frag_off<MAP>(d, wave), 0);
#pragma unroll
        for (int s = 0; s < 48; ++s) {
            __builtin_amdgcn_sched_barrier(0);
            const v4u cur = v[s % DEPTH];
            x ^= cur;
            if (s + DEPTH < 48) v[s % DEPTH] = __builtin_amdgcn_raw_buffer_load_b128(rs, loff, base + frag_off<MAP>(s + DEPTH, wave), 0);
        }
        base += region;                       // 0: the same region every pass (L2-warm); else a fresh one (L2-cold)
    }
    out[blockIdx.x * 768 + threadIdx.x] = x[0] ^ x[1] ^ x[2] ^ x[3];
}
// the Mlp kernel's K loop proper: weight fragment (buffer load, DEPTH ahead), two activation fragments from a K-blocked LDS
// image (BD ahead), two 32x32x32 MFMAs; optionally a workgroup barrier before every pass (the kernel's phases start from one)
typedef int v16i __attribute__((ext_vector_type(16)));
template <int DEPTH, int BD, int BAR>
__global__ __launch_bounds__(768, 1) void kk(const char *buf, int reps, int *out, unsigned region) {
    extern __shared__ __attribute__((aligned(256))) char sm[];
    constexpr int KBLK = 64 * 80 + 64;
    for (int i = threadIdx.x; i < 24 * KBLK / 4; i += 768) reinterpret_cast<int *>(sm)[i] = i * 2654435761u;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(buf), 0, 0x7fffffff, 0x00020000);
    const unsigned loff = lane * 16, fb = (lane & 31) * 80 + (lane >> 5) * 16;
    v16i c0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, c1 = c0;
    unsigned base = 0;
    for (int r = 0; r < reps; ++r) {
        if (BAR) __syncthreads();
        v4i v[DEPTH + 2], b[BD + 2][2];
        int so = base + wave * 1024;
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            v[d] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rs, loff, so, 0));
            asm volatile("s_add_u32 %0, %0, 0x3000" : "+s"(so));
        }
#pragma unroll
        for (int d = 0; d < BD; ++d) {
            b[d][0] = *reinterpret_cast<const v4i *>(sm + (d >> 1) * KBLK + (d & 1) * 32 + fb);
            b[d][1] = *reinterpret_cast<const v4i *>(sm + (d >> 1) * KBLK + 32 * 80 + (d & 1) * 32 + fb);
        }
#pragma unroll
        for (int s = 0; s < 48; ++s) {
            __builtin_amdgcn_sched_barrier(0);
            if (s + DEPTH < 48) {
                v[(s + DEPTH) % (DEPTH + 2)] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rs, loff, so, 0));
                asm volatile("s_add_u32 %0, %0, 0x3000" : "+s"(so));
            }
            if (s + BD < 48) {
                const int d = s + BD;
                b[d % (BD + 2)][0] = *reinterpret_cast<const v4i *>(sm + (d >> 1) * KBLK + (d & 1) * 32 + fb);
                b[d % (BD + 2)][1] = *reinterpret_cast<const v4i *>(sm + (d >> 1) * KBLK + 32 * 80 + (d & 1) * 32 + fb);
            }
            __builtin_amdgcn_sched_barrier(0);
            c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(v[s % (DEPTH + 2)], b[s % (BD + 2)][0], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(v[s % (DEPTH + 2)], b[s % (BD + 2)][1], c1, 0, 0, 0);
        }
        base += region;
    }
    int acc = 0;
    for (int e = 0; e < 16; ++e) acc ^= c0[e] ^ c1[e];
    out[blockIdx.x * 768 + threadIdx.x] = acc;
}
template <int DEPTH, int BD, int BAR>
void runk(const char *buf, int *out, unsigned region) {
    const int reps = 48, smem = 24 * (64 * 80 + 64);
    hipFuncSetAttribute((const void *)kk<DEPTH, BD, BAR>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    kk<DEPTH, BD, BAR><<<256, 768, smem>>>(buf, 2, out, region);
    hipDeviceSynchronize();
    hipEventRecord(a);
    kk<DEPTH, BD, BAR><<<256, 768, smem>>>(buf, reps, out, region);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    printf("K loop (weights + 2 LDS fragments + 2 MFMAs per step) %s, %s, %2d weight / %d LDS steps ahead: %6.2f us per pass (MFMA alone ~5.8)\n",
           region ? "cold" : "warm", BAR ? "barrier per pass" : "free-running", DEPTH, BD, ms * 1e3 / reps);
}

template <int DEPTH, int MAP>
void run(const char *buf, int *out, unsigned region, const char *name) {
    const int reps = 48;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k<DEPTH, MAP><<<256, 768>>>(buf, 2, out, region);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<DEPTH, MAP><<<256, 768>>>(buf, reps, out, region);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    printf("%-34s %s, %2d in flight per wave: %6.2f us per 576 KB pass = %5.1f B/clk/CU at 2.1 GHz\n", name, region ? "cold" : "warm", DEPTH,
           ms * 1e3 / reps, 576.0 * 1024 / (ms * 1e-3 / reps) / 2.1e9);
}
int main() {
    char *buf; int *out;
    hipMalloc(&buf, 64 << 20); hipMemset(buf, 1, 64 << 20); hipMalloc(&out, 256 * 768 * 4);
    for (unsigned region : {0u, 1u << 20}) {
        runk<6, 2, 0>(buf, out, region); runk<6, 2, 1>(buf, out, region); runk<12, 2, 1>(buf, out, region); runk<16, 3, 1>(buf, out, region);
    }
    if (getenv("KLOOP_ONLY")) return 0;
    for (unsigned region : {0u, 1u << 20}) {
        run<6, 0>(buf, out, region, "contiguous 12 KB window per step"); run<12, 0>(buf, out, region, "contiguous 12 KB window per step");
        run<6, 1>(buf, out, region, "waves 4 KB apart"); run<12, 1>(buf, out, region, "waves 4 KB apart");
        run<6, 3>(buf, out, region, "waves 2 KB apart"); run<12, 3>(buf, out, region, "waves 2 KB apart");
        run<6, 2>(buf, out, region, "per-wave streams, 256 B skew"); run<12, 2>(buf, out, region, "per-wave streams, 256 B skew");
        run<6, 4>(buf, out, region, "16 KB window pitch"); run<12, 4>(buf, out, region, "16 KB window pitch");
    }
    return 0;
}
