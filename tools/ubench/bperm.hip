// ds_bpermute_b32 vs ds_read_u8 gather throughput (table lookups of the ShiftGELU pass), 12 waves per CU.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/bperm.hip -o tools/ubench/bperm
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(768) void k_bperm(const unsigned *in, unsigned *out, int iters) {
    unsigned x = in[threadIdx.x], line = in[768 + (threadIdx.x & 63)], acc = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned q = (x >> (8 * k)) & 0xffu;
            const unsigned v = (unsigned)__builtin_amdgcn_ds_bpermute((int)q, (int)line);
            acc += __builtin_amdgcn_ubfe(v, q << 3, 8);
        }
        x = x * 1664525u + 1013904223u + acc;
    }
    out[blockIdx.x * 768 + threadIdx.x] = acc;
}
__global__ __launch_bounds__(768) void k_gather(const unsigned *in, unsigned *out, int iters) {
    __shared__ unsigned char tab[12 * 256];
    for (int i = threadIdx.x; i < 12 * 256; i += 768) tab[i] = (unsigned char)in[i & 1023];
    __syncthreads();
    unsigned x = in[threadIdx.x], acc = 0;
    const unsigned char *t = tab + (threadIdx.x >> 6) * 256;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k) acc += t[(x >> (8 * k)) & 0xffu];
        x = x * 1664525u + 1013904223u + acc;
    }
    out[blockIdx.x * 768 + threadIdx.x] = acc;
}
__global__ void k_sem(const unsigned *in, unsigned *out) {   // semantics: lane = (addr >> 2) & 63, low bits ignored
    const unsigned line = threadIdx.x * 0x01010101u;
    out[threadIdx.x] = (unsigned)__builtin_amdgcn_ds_bpermute((int)in[threadIdx.x], (int)line);
}
int main() {
    unsigned *in, *out; hipMalloc(&in, 4096 * 4); hipMalloc(&out, 256 * 768 * 4);
    unsigned h[4096]; srand(1); for (auto &v : h) v = rand() * 65536u + rand();
    hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    for (int which = 0; which < 2; ++which) for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (which == 0) k_bperm<<<256, 768>>>(in, out, iters); else k_gather<<<256, 768>>>(in, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // per CU: 12 waves x iters x 4 lookups wave-instructions
        printf("%s: %.3f ms, %.2f ns per wave-lookup per CU (= %.1f cycles at 2.1 GHz)\n", which ? "ds_read_u8 gather" : "ds_bpermute", ms,
               ms * 1e6 / (12.0 * iters * 4), ms * 1e6 / (12.0 * iters * 4) * 2.1);
    }
    unsigned a[64]; for (int i = 0; i < 64; ++i) a[i] = (unsigned)(i * 37 + 3) & 0xff;
    hipMemcpy(in, a, sizeof a, hipMemcpyHostToDevice);
    k_sem<<<1, 64>>>(in, out);
    unsigned o[64]; hipMemcpy(o, out, sizeof o, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 64; ++i) bad += (o[i] & 0xff) != ((a[i] >> 2) & 63);
    printf("bpermute semantics (lane = addr[7:2]): %s\n", bad ? "MISMATCH" : "ok");
    return 0;
}
