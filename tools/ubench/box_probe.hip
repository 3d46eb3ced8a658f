// box_probe.hip — "which box is this" probe for bench.py (NOT part of the product library): register-only int8 MFMA loops on
// random operands (both shapes the kernels use) and one device-to-device copy, ~50 ms each, timed with HIP events.  MI355X boxes
// of this pool differ by 5-8 % in sustained clock under int8 MFMA load; the bench line prints these three numbers so that a
// reader can tell a faster box from a faster kernel.
// Build: hipcc --offload-arch=gfx950 -O3 -fPIC -shared tools/ubench/box_probe.hip -o tools/ubench/libbox_probe.so
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int SHAPE>
__global__ __launch_bounds__(256) void box_mfma_kernel(int *out, int n, unsigned seed) {
    v4i a[4], b[4];
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 4; ++e) {
            unsigned x = (threadIdx.x * 2654435761u + i * 40503u + e * 977u) * seed;
            a[i][e] = (int)x;
            b[i][e] = (int)(x * 2246822519u);
        }
    int s = 0;
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
        for (int i = 0; i < 4; ++i)
            for (int r = 0; r < 16; ++r) s += c[i][r];
    } else {
        v4i c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
        // eight independent accumulators, the MFMAs as asm statements: through the builtin hipcc rotates the accumulators through
        // AGPR copies every iteration (7 VALU moves per MFMA) and the loop measures those
#define BOX_MFMA16(c, x, y) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(c) : "v"(x), "v"(y))
        for (int it = 0; it < n; ++it) {
            BOX_MFMA16(c0, a[0], b[0]); BOX_MFMA16(c1, a[1], b[1]); BOX_MFMA16(c2, a[2], b[2]); BOX_MFMA16(c3, a[3], b[3]);
            BOX_MFMA16(c4, b[0], a[0]); BOX_MFMA16(c5, b[1], a[1]); BOX_MFMA16(c6, b[2], a[2]); BOX_MFMA16(c7, b[3], a[3]);
        }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // the last MFMAs' results before the compiler's readers
        for (int r = 0; r < 4; ++r) s += c0[r] + c1[r] + c2[r] + c3[r] + c4[r] + c5[r] + c6[r] + c7[r];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void box_copy_kernel(const int4 *__restrict__ src, int4 *__restrict__ dst, long long n16) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long long)gridDim.x * 256) dst[i] = src[i];
}

template <int SHAPE>
static double mfma_tops(int *buf, int num_cu, int waves_per_simd, double target_ms, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
    const dim3 grid(num_cu * waves_per_simd), block(256);
    const double op_per_mfma = SHAPE == 32 ? 65536.0 : 32768.0;
    int n = 256;
    float ms = 0.f;
    box_mfma_kernel<SHAPE><<<grid, block, 0, st>>>(buf, 16, 12345u);
    for (int pass = 0; pass < 2; ++pass) {       // pass 0 sizes the loop for ~target_ms, pass 1 is the measurement
        if (hipEventRecord(e0, st) != hipSuccess) return -1.0;
        box_mfma_kernel<SHAPE><<<grid, block, 0, st>>>(buf, n, 12345u);
        if (hipEventRecord(e1, st) != hipSuccess || hipEventSynchronize(e1) != hipSuccess) return -1.0;
        if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess || ms <= 0.f) return -1.0;
        if (pass == 0) n = (int)(n * target_ms / ms) + 1;
    }
    return (double)grid.x * 4 * n * 8 * op_per_mfma / (ms * 1e-3) / 1e12;
}

// out[0] = TOP/s of v_mfma_i32_32x32x32_i8 (4 waves per SIMD, random operands), out[1] = the same for 16x16x64 (4 waves per SIMD,
// 8 accumulators), out[2] = GB/s (read + write) of a 256 MB device copy, out[3] = CUs.  Returns 0, or a HIP error code.
extern "C" int box_probe(int device, double target_ms, double *out) {
    if (hipSetDevice(device) != hipSuccess) return 1;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return 2;
    const int num_cu = prop.multiProcessorCount;
    hipStream_t st;
    hipEvent_t e0, e1;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return 3;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return 4;
    const size_t bytes = (size_t)256 << 20;
    char *src = nullptr, *dst = nullptr;
    int *buf = nullptr;
    int rc = 0;
    if (hipMalloc((void **)&src, bytes) != hipSuccess || hipMalloc((void **)&dst, bytes) != hipSuccess ||
        hipMalloc((void **)&buf, (size_t)num_cu * 4 * 256 * 4) != hipSuccess) rc = 5;
    if (!rc) {
        out[0] = mfma_tops<32>(buf, num_cu, 4, target_ms, st, e0, e1);
        out[1] = mfma_tops<16>(buf, num_cu, 4, target_ms, st, e0, e1);
        (void)hipMemsetAsync(src, 1, bytes, st);
        const long long n16 = (long long)(bytes / 16);
        box_copy_kernel<<<num_cu * 8, 256, 0, st>>>((const int4 *)src, (int4 *)dst, n16);
        int reps = 4;
        float ms = 0.f;
        for (int pass = 0; pass < 2 && !rc; ++pass) {
            (void)hipEventRecord(e0, st);
            for (int r = 0; r < reps; ++r) box_copy_kernel<<<num_cu * 8, 256, 0, st>>>((const int4 *)src, (int4 *)dst, n16);
            (void)hipEventRecord(e1, st);
            if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess || ms <= 0.f) rc = 6;
            else if (pass == 0) reps = (int)(reps * target_ms / ms) + 1;
        }
        if (!rc) out[2] = 2.0 * (double)bytes * reps / (ms * 1e-3) / 1e9;
        out[3] = (double)num_cu;
        if (out[0] < 0 || out[1] < 0) rc = 7;
    }
    (void)hipFree(src); (void)hipFree(dst); (void)hipFree(buf);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipStreamDestroy(st);
    return rc;
}
