// Minimal form of the round-3 / round-4 one-LSB LayerNorm differences.  tools/ubench/ln_s1_standalone.hip established (round 5) that
// the aggressor is NOT gemm_glds_kernel but gemm_nt_kernel (ivit_gemm.h), whose epilogue requantises with v_mul_f64 x 2, v_rndne_f64,
// double compares / selects and v_cvt_i32_f64, and that the victim needs v_pk_*_f32 in its ISA.  This probe strips both down:
//   victim:    waves that add / multiply / fma pairs of floats with v_pk_{add,mul,fma}_f32 and the same values with the scalar
//              v_{add,mul,fma}_f32, and count iterations where the two differ in any bit;
//   aggressor: waves on the SAME CUs that issue ONE kind of instruction in a tight loop (KIND below), launched first and long-lived.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/pk_f64_hazard.hip -o tools/ubench/pk_f64_hazard
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float v2f __attribute__((ext_vector_type(2)));

#include "f64_aggressors.h"

// OPS: 0 add, 1 mul, 2 fma
template <int OPS>
__global__ __launch_bounds__(128) void victim(const float *in, int *bad, int n) {
    v2f x = {in[threadIdx.x], in[threadIdx.x + 128]}, y = {in[(threadIdx.x * 7) & 255], in[(threadIdx.x * 13) & 255]};
    v2f acc = {0.f, 0.f};
    float s0 = 0.f, s1 = 0.f;
    int nbad = 0;
    for (int it = 0; it < n; ++it) {
        if (OPS == 0) {
            asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(acc) : "v"(x));
            asm volatile("v_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %3" : "+v"(s0), "+v"(s1) : "v"(x[0]), "v"(x[1]));
        } else if (OPS == 1) {
            asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(acc) : "v"(x), "v"(y));
            asm volatile("v_mul_f32 %0, %2, %4\n\tv_mul_f32 %1, %3, %5" : "=&v"(s0), "=&v"(s1) : "v"(x[0]), "v"(x[1]), "v"(y[0]), "v"(y[1]));
        } else {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y));
            asm volatile("v_fma_f32 %0, %2, %4, %0\n\tv_fma_f32 %1, %3, %5, %1" : "+v"(s0), "+v"(s1) : "v"(x[0]), "v"(x[1]), "v"(y[0]), "v"(y[1]));
        }
        if (__float_as_int(acc[0]) != __float_as_int(s0) || __float_as_int(acc[1]) != __float_as_int(s1)) { ++nbad; acc[0] = s0; acc[1] = s1; }
        x[0] = x[0] * 1.0001f + 1e-3f; x[1] = x[1] * 0.9999f + 2e-3f;
        y[0] = y[0] * 0.99995f + 1e-4f; y[1] = y[1] * 1.00005f - 1e-4f;
        if (OPS == 2 && (it & 63) == 63) { acc[0] = s0 = acc[0] * 1e-6f; acc[1] = s1 = acc[1] * 1e-6f; }
    }
    if (nbad) atomicAdd(bad, nbad);
}

int main(int argc, char **argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 10;
    float *in; int *bad; double *sink;
    hipMalloc(&in, 1024); hipMalloc(&bad, 8); hipMalloc(&sink, 1024 * 256 * 8);
    float h[256];
    for (int i = 0; i < 256; ++i) h[i] = 1.0f + (rand() % 100000) * 1e-5f;
    hipMemcpy(in, h, 1024, hipMemcpyHostToDevice);
    hipStream_t sv, sa;
    hipStreamCreateWithFlags(&sv, hipStreamNonBlocking); hipStreamCreateWithFlags(&sa, hipStreamNonBlocking);
    const char *const *names = f64_aggr_names;
    for (int kind = 0; kind < 14; ++kind) {
        int tot[3] = {0, 0, 0};
        for (int ops = 0; ops < 3; ++ops) {
            hipMemset(bad, 0, 8);
            hipDeviceSynchronize();
            for (int r = 0; r < reps; ++r) {
                launch_f64_aggressor(kind, sink, 1024, 20000, sa);
                for (int k = 0; k < 6; ++k) {
                    if (ops == 0) victim<0><<<2048, 128, 0, sv>>>(in, bad, 4000);
                    else if (ops == 1) victim<1><<<2048, 128, 0, sv>>>(in, bad, 4000);
                    else victim<2><<<2048, 128, 0, sv>>>(in, bad, 4000);
                }
                hipDeviceSynchronize();
            }
            hipMemcpy(&tot[ops], bad, 4, hipMemcpyDeviceToHost);
        }
        printf("aggressor %-32s: v_pk_add_f32 %8d  v_pk_mul_f32 %8d  v_pk_fma_f32 %8d  mismatching iterations (of %.2g per op)\n", names[kind], tot[0], tot[1], tot[2],
               (double)reps * 6 * 2048 * 128 * 4000);
    }
    return 0;
}
