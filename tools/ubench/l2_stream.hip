// L2 -> register streaming rate when EVERY CU reads the same small buffer (the fused Mlp's weight stream): 8 waves per CU,
// each wave reads its own 1/8 of a `bytes`-sized buffer in 1 KB (16 B per lane) pieces, DEPTH loads in flight per wave.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/l2_stream.hip -o tools/ubench/l2_stream
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int v4i __attribute__((ext_vector_type(4)));
template <int DEPTH>
__global__ __launch_bounds__(512, 2) void k_stream(const v4i *buf, int frags_per_wave, int reps, int *out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const v4i *p = buf + (size_t)wave * frags_per_wave * 64 + lane;
    v4i acc = {0, 0, 0, 0};
    for (int r = 0; r < reps; ++r) {
        for (int f = 0; f < frags_per_wave; f += DEPTH) {
            v4i v[DEPTH];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) v[d] = p[(size_t)(f + d) * 64];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) acc ^= v[d];
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
}
template <int DEPTH>
void run(const v4i *buf, size_t bytes, int *out, int wgs) {
    const int fpw = (int)(bytes / 1024 / 8), reps = 64;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k_stream<DEPTH><<<wgs, 512>>>(buf, fpw, 2, out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k_stream<DEPTH><<<wgs, 512>>>(buf, fpw, reps, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double tb = (double)wgs * reps * fpw * 8 * 1024.0 / (ms * 1e-3) / 1e12;
    printf("buffer %7.0f KB, %3d WGs x 8 waves, %2d loads in flight per wave: %6.2f TB/s  (%5.1f B/clk/CU at 2.1 GHz)\n",
           bytes / 1024.0, wgs, DEPTH, tb, tb * 1e12 / wgs / 2.1e9);
}
int main() {
    v4i *buf; int *out;
    hipMalloc(&buf, 64 << 20); hipMemset(buf, 1, 64 << 20); hipMalloc(&out, 512 * 512 * 4);
    for (size_t kb : {576, 1152, 4608, 32768}) {
        run<3>(buf, kb * 1024, out, 256); run<6>(buf, kb * 1024, out, 256); run<12>(buf, kb * 1024, out, 256); run<24>(buf, kb * 1024, out, 256);
    }
    run<12>(buf, 1152 * 1024, out, 512);
    return 0;
}
