// Per-instruction VALU throughput probe for gfx950 (cycles per wave-instruction per SIMD).
// Build here: hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rates.hip -o tools/ubench/valu_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#define N_IT 2048
#define DEFK(name, T, init, body)                                                    \
    __global__ void name(T *out, int n) {                                            \
        T a0 = init + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;             \
        T a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;                          \
        for (int i = 0; i < n; ++i) {                                                \
            body(a0) body(a1) body(a2) body(a3) body(a4) body(a5) body(a6) body(a7)   \
        }                                                                            \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7; \
    }
#define B_F32MUL(x) asm volatile("v_mul_f32 %0, %0, %0" : "+v"(x));
#define B_F32RND(x) asm volatile("v_rndne_f32 %0, %0" : "+v"(x));
#define B_F32CVTI(x) asm volatile("v_cvt_i32_f32 %0, %0\n v_cvt_f32_i32 %0, %0" : "+v"(x));
#define B_F64MUL(x) asm volatile("v_mul_f64 %0, %0, %0" : "+v"(x));
#define B_F64FMA(x) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(x));
#define B_F64RND(x) asm volatile("v_rndne_f64 %0, %0" : "+v"(x));
#define B_F64ADD(x) asm volatile("v_add_f64 %0, %0, %0" : "+v"(x));
#define B_F32DIV(x) x = 1.0f / x;
#define B_F32RCP(x) asm volatile("v_rcp_f32 %0, %0" : "+v"(x));
#define B_F32FMA(x) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x));
#define B_F32FLOOR(x) asm volatile("v_floor_f32 %0, %0" : "+v"(x));
#define B_F32LDEXP(x) asm volatile("v_ldexp_f32 %0, %0, 1" : "+v"(x));
DEFK(k_f32mul, float, 1.0f, B_F32MUL)
DEFK(k_f32rnd, float, 1.5f, B_F32RND)
DEFK(k_f32cvt2, float, 1.5f, B_F32CVTI)
DEFK(k_f64mul, double, 1.0, B_F64MUL)
DEFK(k_f64fma, double, 1.0, B_F64FMA)
DEFK(k_f64rnd, double, 1.5, B_F64RND)
DEFK(k_f64add, double, 1.5, B_F64ADD)
DEFK(k_f32div, float, 1.5f, B_F32DIV)
DEFK(k_f32rcp, float, 1.5f, B_F32RCP)
DEFK(k_f32fma, float, 1.0f, B_F32FMA)
DEFK(k_f32floor, float, 1.5f, B_F32FLOOR)
DEFK(k_f32ldexp, float, 1.5f, B_F32LDEXP)
typedef float v2f __attribute__((ext_vector_type(2)));
__global__ void k_pkfma(float *out, int n) {   // v_pk_fma_f32: two fp32 FMAs per lane per instruction
    v2f a[8];
    for (int k = 0; k < 8; ++k) a[k] = v2f{1.0f + threadIdx.x + k, 0.5f + k};
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(a[k]));
    }
    float s = 0;
    for (int k = 0; k < 8; ++k) s += a[k][0] + a[k][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_pkadd(float *out, int n) {
    v2f a[8];
    for (int k = 0; k < 8; ++k) a[k] = v2f{1.0f + threadIdx.x + k, 0.5f + k};
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(a[k]));
    }
    float s = 0;
    for (int k = 0; k < 8; ++k) s += a[k][0] + a[k][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_pkmul(float *out, int n) {
    v2f a[8];
    for (int k = 0; k < 8; ++k) a[k] = v2f{1.0f + threadIdx.x * 1e-3f + k, 0.5f + k};
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(a[k]));
    }
    float s = 0;
    for (int k = 0; k < 8; ++k) s += a[k][0] + a[k][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_f32add(float *out, int n) {
    float a[8];
    for (int k = 0; k < 8; ++k) a[k] = 1.0f + threadIdx.x + k;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) asm volatile("v_add_f32 %0, %0, %0" : "+v"(a[k]));
    }
    float s = 0;
    for (int k = 0; k < 8; ++k) s += a[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_ldsgather(float *out, int n) {   // dependent-free random ds_read_b32 over a 1 KB table
    __shared__ float tbl[256];
    tbl[threadIdx.x & 255] = threadIdx.x;
    __syncthreads();
    unsigned idx[8];
    for (int k = 0; k < 8; ++k) idx[k] = (threadIdx.x * 2654435761u + k * 40503u) >> 8;
    float s = 0;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float v;
            asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"((idx[k] & 255u) * 4u));
            asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
            idx[k] = idx[k] * 1664525u + 1013904223u + (unsigned)__float_as_int(v);
        }
    }
    for (int k = 0; k < 8; ++k) s += idx[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_cvt64(double *out, int n) {   // cvt_f64_i32 + cvt_i32_f64 pair
    int a[8];
    for (int k = 0; k < 8; ++k) a[k] = threadIdx.x + k;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            double d;
            asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(d) : "v"(a[k]));
            asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(a[k]) : "v"(d));
        }
    }
    int s = 0;
    for (int k = 0; k < 8; ++k) s += a[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename K, typename T>
void run(const char *name, K kern, T *buf, int per_iter) {
    int dev_cus = 256;
    dim3 grid(dev_cus * 4), block(256);   // 4 blocks x 4 waves per CU = 4 waves/SIMD
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(kern, grid, block, 0, 0, buf, 16);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(kern, grid, block, 0, 0, buf, N_IT);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    double wave_insts_per_simd = (double)N_IT * per_iter * 4.0;   // 4 waves per SIMD
    printf("%-12s %8.3f ms  -> %6.2f ns per wave-instr per SIMD (x2.4 GHz = %5.1f cyc)\n", name, ms,
           ms * 1e6 / wave_insts_per_simd, ms * 1e6 / wave_insts_per_simd * 2.4);
}
int main() {
    void *buf;
    hipMalloc(&buf, 256 * 4 * 256 * 8);
    run("f32 mul", k_f32mul, (float *)buf, 8);
    run("f32 fma", k_f32fma, (float *)buf, 8);
    run("f32 rndne", k_f32rnd, (float *)buf, 8);
    run("f32 floor", k_f32floor, (float *)buf, 8);
    run("f32 ldexp", k_f32ldexp, (float *)buf, 8);
    run("f32 cvt x2", k_f32cvt2, (float *)buf, 16);
    run("f32 rcp", k_f32rcp, (float *)buf, 8);
    run("f32 div(1/x)", k_f32div, (float *)buf, 8);
    run("f64 mul", k_f64mul, (double *)buf, 8);
    run("f64 fma", k_f64fma, (double *)buf, 8);
    run("f64 add", k_f64add, (double *)buf, 8);
    run("f64 rndne", k_f64rnd, (double *)buf, 8);
    run("cvt64 pair", k_cvt64, (double *)buf, 16);
    run("pk_fma_f32", k_pkfma, (float *)buf, 8);
    run("pk_add_f32", k_pkadd, (float *)buf, 8);
    run("pk_mul_f32", k_pkmul, (float *)buf, 8);
    run("f32 add", k_f32add, (float *)buf, 8);
    run("lds gather(+2 valu)", k_ldsgather, (float *)buf, 8);
    return 0;
}
