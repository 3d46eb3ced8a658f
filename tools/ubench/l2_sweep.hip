// L2 -> register streaming rate per CU as a function of the FOOTPRINT every CU re-reads (the fused Mlp's weight sweep):
// all 256 CUs in phase, W waves per CU, wave w takes fragment s * W + w of step s (one contiguous W KB window per step
// and CU), DEPTH 16-byte loads in flight per wave, straight-line unconditional loads.  Prices a D = 768 fused Mlp
// (4.7 MB per pass through 4 MB of L2 per XCD) and a role-split D = 384 one (two 590 KB streams side by side).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/l2_sweep.hip -o tools/ubench/l2_sweep
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int v4i __attribute__((ext_vector_type(4)));

template <int DEPTH, int W>
__global__ __launch_bounds__(W * 64, 1) void k_sweep(const v4i *buf, int steps, int reps, int *out, unsigned long long *clk) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const v4i *p0 = buf + (size_t)wave * 64 + lane;
    v4i x = {lane, wave, 3, 4};
    const unsigned long long t0 = __builtin_readcyclecounter();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int r = 0; r < reps; ++r) {
        const v4i *p = p0;
        v4i v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) v[d] = p[(size_t)d * W * 64];
        for (int s = 0; s < steps; s += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const v4i cur = v[d];
                v[d] = p[(size_t)(s + DEPTH + d) * W * 64];        // the buffer is padded by DEPTH steps: no condition
                x ^= cur;
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * W * 64 + threadIdx.x] = x[0] ^ x[1] ^ x[2] ^ x[3];
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int DEPTH, int W>
void run(const v4i *buf, int *out, unsigned long long *clk, size_t footprint) {
    const int steps = (int)(footprint / (W * 1024)) / DEPTH * DEPTH;
    const int reps = (int)((256ull << 20) / footprint) + 2;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k_sweep<DEPTH, W><<<256, W * 64>>>(buf, steps, 2, out, clk);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k_sweep<DEPTH, W><<<256, W * 64>>>(buf, steps, reps, out, clk);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    static unsigned long long h[512];
    hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    double cyc = 0, rt = 0;
    for (int i = 0; i < 256; ++i) { cyc += (double)h[2 * i]; rt += (double)h[2 * i + 1]; }
    const double bytes = (double)reps * steps * W * 1024.0;                  // per CU
    printf("footprint %7.0f KB, %2d waves x %2d loads in flight: %6.2f TB/s aggregate, %5.1f B/clk/CU (shader clock %4.0f MHz), %7.1f us per pass\n",
           steps * W * 1.0, W, DEPTH, bytes * 256 / (ms * 1e-3) / 1e12, bytes / (cyc / 256), cyc / rt * 100.0, ms * 1e3 / reps);
}

int main() {
    v4i *buf; int *out; unsigned long long *clk;
    hipMalloc(&buf, 64 << 20); hipMemset(buf, 1, 64 << 20); hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&clk, 512 * 8);
    const size_t fp[] = {576u << 10, 1180u << 10, 2360u << 10, 3540u << 10, 4720u << 10, 9440u << 10, 18880u << 10};
    for (size_t f : fp) {
        run<6, 8>(buf, out, clk, f);
        run<12, 8>(buf, out, clk, f);
        run<12, 12>(buf, out, clk, f);
        run<12, 4>(buf, out, clk, f);
        run<24, 4>(buf, out, clk, f);
    }
    return 0;
}
