// Which exact-requant instruction mix hides under int8 MFMAs on gfx950?
//
// A wave issues v_mfma_i32_32x32x32_i8 (random operands) and, behind each MFMA, NE requantised outputs in one of
// several instruction mixes.  Reported in SHADER cycles (s_memtime) per MFMA slot per SIMD, together with the
// shader clock the run sustained (s_memtime ticks / s_memrealtime ticks, the latter at 100 MHz).
//   MIX 0: nothing
//   MIX 1: fp64   v_cvt_f64_i32 + v_fma_f64 (magic) + v_med3_i32                (the round-2 epilogue)
//   MIX 2: int    v_mul_hi_i32 + v_add_u32 + v_ashrrev_i32 + v_med3_i32         (exact integer dyadic)
//   MIX 3: fp32   v_cvt_f32_i32 + v_fma_f32 + v_cvt_pk_u8_f32                   (inexact: cost reference only)
//   MIX 4: plain  4 x v_add_u32                                                 (issue-slot reference)
//   MIX 5: i24    v_mul_hi_i32_i24 + v_mul_i32_i24 + v_alignbit + v_med3        (24-bit multiplier pieces)
//   MIX 6: fp64   v_cvt_f64_i32 + v_mul_f64 + v_add_f64 + v_med3_i32
// Also: VALU-only rates of the single instructions (cycles per wave-instruction per SIMD, 4 waves/SIMD).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/requant_mix.hip -o tools/ubench/requant_mix
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int MIX>
__device__ __forceinline__ int requant_one(int z, double cd, int ci, float cf, int sh) {
    int v;
    if (MIX == 1) {
        double t;
        asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(t) : "v"(z));
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(t) : "v"(cd), "v"(6755399441055744.0));
        v = __double2loint(t);
        asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(v) : "v"(-128), "v"(127));
    } else if (MIX == 2) {
        asm volatile("v_mul_hi_i32 %0, %1, %2" : "=v"(v) : "v"(z), "v"(ci));
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(v) : "v"(1 << 10));
        asm volatile("v_ashrrev_i32 %0, %1, %0" : "+v"(v) : "v"(sh));
        asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(v) : "v"(-128), "v"(127));
    } else if (MIX == 3) {
        float t;
        asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(t) : "v"(z));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(t) : "v"(cf), "v"(128.0f));
        v = 0;
        asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, %0" : "+v"(v) : "v"(t));
    } else if (MIX == 4) {
        v = z;
        asm volatile("v_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1" : "+v"(v) : "v"(ci));
    } else if (MIX == 5) {
        int hi, lo;
        asm volatile("v_mul_hi_i32_i24 %0, %1, %2" : "=v"(hi) : "v"(z), "v"(ci));
        asm volatile("v_mul_i32_i24 %0, %1, %2" : "=v"(lo) : "v"(z), "v"(ci));
        asm volatile("v_alignbit_b32 %0, %1, %2, %3" : "=v"(v) : "v"(hi), "v"(lo), "v"(sh));
        asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(v) : "v"(-128), "v"(127));
    } else if (MIX == 6) {
        double t;
        asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(t) : "v"(z));
        asm volatile("v_mul_f64 %0, %0, %1" : "+v"(t) : "v"(cd));
        asm volatile("v_add_f64 %0, %0, %1" : "+v"(t) : "v"(6755399441055744.0));
        v = __double2loint(t);
        asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(v) : "v"(-128), "v"(127));
    } else {
        v = z;
    }
    return v;
}

template <int MIX, int NE, int MF, int WPS>
__global__ __launch_bounds__(256, WPS) void k_mix(int *out, unsigned long long *clk, int n, int seed, double cd) {
    v4i a[2], b[2];
    for (int i = 0; i < 2; ++i)
        for (int e = 0; e < 4; ++e) {
            unsigned x = (threadIdx.x * 2654435761u + i * 40503u + e * 977u) * (unsigned)seed;
            a[i][e] = (int)x;
            b[i][e] = (int)(x * 2246822519u);
        }
    v16i c[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) c[i][r] = 0;
    int z[16];
    for (int i = 0; i < 16; ++i) z[i] = threadIdx.x * 31 + i * 1001;
    const int ci = 0x5a3c1e77 ^ seed;
    const float cf = (float)cd;
    const unsigned long long t0 = __builtin_readcyclecounter();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (MF) c[m & 3] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[m & 1], b[(m >> 1) & 1], c[m & 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const int g = (m * NE + e) & 15;
                z[g] = requant_one<MIX>(z[g] + it, cd, ci, cf, 11) + 77 * g;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    int s = 0;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += c[i][r];
    for (int i = 0; i < 16; ++i) s ^= z[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int MIX, int NE, int MF, int WPS>
void run(int *buf, unsigned long long *clk, const char *name) {
    const int n = 2048;
    dim3 grid(256 * WPS), block(256);
    hipLaunchKernelGGL((k_mix<MIX, NE, MF, WPS>), grid, block, 0, 0, buf, clk, 64, 12345, 1.0e-3);
    hipDeviceSynchronize();
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    hipLaunchKernelGGL((k_mix<MIX, NE, MF, WPS>), grid, block, 0, 0, buf, clk, n, 12345, 1.0e-3);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    static unsigned long long h[2 * 256 * 8];
    hipMemcpy(h, clk, sizeof(unsigned long long) * 2 * grid.x, hipMemcpyDeviceToHost);
    double cyc = 0, rt = 0;
    for (unsigned i = 0; i < grid.x; ++i) { cyc += (double)h[2 * i]; rt += (double)h[2 * i + 1]; }
    const double mhz = cyc / rt * 100.0;                      // s_memrealtime ticks at 100 MHz
    const double slots = (double)n * 8 * WPS;                 // MFMA slots per SIMD
    const double wave_cyc = (cyc / grid.x) / ((double)n * 8);   // a wave's shader cycles per slot; WPS waves share the SIMD
    printf("%-6s NE %d %s w/SIMD %d : %7.3f ms  clock %5.0f MHz  %6.1f cyc/slot/SIMD (wall %6.2f ns/slot)", name, NE,
           MF ? "mfma+valu" : "valu only", WPS, ms, mhz, wave_cyc / WPS, ms * 1e6 / slots);
    if (MF) printf("  %5.0f TOP/s", (double)grid.x * 4 * n * 8 * 65536.0 / ms / 1e9);
    printf("\n");
}


// ---- MFMA issue rate, both int8 shapes, NACC independent accumulators, shader cycles + sustained clock
template <int SHAPE, int NACC, int WPS>
__global__ __launch_bounds__(256, WPS) void k_peak(int *out, unsigned long long *clk, int n, int seed) {
    v4i a[4], b[4];
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 4; ++e) {
            unsigned x = (threadIdx.x * 2654435761u + i * 40503u + e * 977u) * (unsigned)seed;
            a[i][e] = (int)x;
            b[i][e] = (int)(x * 2246822519u);
        }
    int s = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    if (SHAPE == 32) {
        v16i c[NACC];
        for (int i = 0; i < NACC; ++i)
            for (int r = 0; r < 16; ++r) c[i][r] = 0;
        for (int it = 0; it < n; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) c[i % NACC] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i & 3], b[(i >> 2) & 3], c[i % NACC], 0, 0, 0);
        }
        for (int i = 0; i < NACC; ++i)
            for (int r = 0; r < 16; ++r) s += c[i][r];
    } else {
        v4i c[NACC];
        for (int i = 0; i < NACC; ++i)
            for (int r = 0; r < 4; ++r) c[i][r] = 0;
        for (int it = 0; it < n; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) c[i % NACC] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[i & 3], b[(i >> 2) & 3], c[i % NACC], 0, 0, 0);
        }
        for (int i = 0; i < NACC; ++i)
            for (int r = 0; r < 4; ++r) s += c[i][r];
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}
template <int SHAPE, int NACC, int WPS>
void peak(int *buf, unsigned long long *clk, int seed) {
    const int n = 4096;
    dim3 grid(256 * WPS), block(256);
    hipLaunchKernelGGL((k_peak<SHAPE, NACC, WPS>), grid, block, 0, 0, buf, clk, 64, seed);
    hipDeviceSynchronize();
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    hipLaunchKernelGGL((k_peak<SHAPE, NACC, WPS>), grid, block, 0, 0, buf, clk, n, seed);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    static unsigned long long h[2 * 256 * 8];
    hipMemcpy(h, clk, sizeof(unsigned long long) * 2 * grid.x, hipMemcpyDeviceToHost);
    double cyc = 0, rt = 0;
    for (unsigned i = 0; i < grid.x; ++i) { cyc += (double)h[2 * i]; rt += (double)h[2 * i + 1]; }
    const double op = SHAPE == 32 ? 65536.0 : 32768.0;
    printf("peak %s acc %2d w/SIMD %d %-6s: %7.3f ms  clock %5.0f MHz  %5.1f shader cyc/MFMA/SIMD  %6.0f TOP/s\n",
           SHAPE == 32 ? "32x32x32" : "16x16x64", NACC, WPS, seed ? "random" : "zero", ms, cyc / rt * 100.0,
           (cyc / grid.x) / ((double)n * 16) / WPS, (double)grid.x * 4 * n * 16 * op / ms / 1e9);
}

// ---- single-instruction VALU rates, 4 waves per SIMD, shader cycles
#define DEFR(name, decl, body)                                                              \
    __global__ __launch_bounds__(256) void name(int *out, unsigned long long *clk, int n) { \
        decl;                                                                               \
        const unsigned long long t0 = __builtin_readcyclecounter();                         \
        for (int i = 0; i < n; ++i) {                                                       \
            _Pragma("unroll") for (int k = 0; k < 8; ++k) { body; }                         \
        }                                                                                   \
        const unsigned long long t1 = __builtin_readcyclecounter();                         \
        int s = 0;                                                                          \
        for (int k = 0; k < 8; ++k) s += (int)x[k];                                         \
        out[blockIdx.x * 256 + threadIdx.x] = s;                                            \
        if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;                                    \
    }
#define DI int x[8]; for (int k = 0; k < 8; ++k) x[k] = threadIdx.x * 77 + k; int y = threadIdx.x | 0x40000001
#define DF float x[8]; for (int k = 0; k < 8; ++k) x[k] = threadIdx.x * 1.5f + k; float y = 1.0000001f
#define DD double x[8]; for (int k = 0; k < 8; ++k) x[k] = threadIdx.x * 1.5 + k; double y = 1.0000001
DEFR(r_add, DI, asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[k]) : "v"(y)))
DEFR(r_mulhi, DI, asm volatile("v_mul_hi_i32 %0, %0, %1" : "+v"(x[k]) : "v"(y)))
DEFR(r_mullo, DI, asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[k]) : "v"(y)))
DEFR(r_mulhi24, DI, asm volatile("v_mul_hi_i32_i24 %0, %0, %1" : "+v"(x[k]) : "v"(y)))
DEFR(r_mul24, DI, asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(x[k]) : "v"(y)))
DEFR(r_mad24, DI, asm volatile("v_mad_i32_i24 %0, %0, %1, %0" : "+v"(x[k]) : "v"(y)))
DEFR(r_med3, DI, asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(y), "v"(127)))
DEFR(r_perm, DI, asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(y), "v"(0x05040100)))
DEFR(r_ashr, DI, asm volatile("v_ashrrev_i32 %0, 3, %0" : "+v"(x[k])))
DEFR(r_lshlor, DI, asm volatile("v_lshl_or_b32 %0, %0, 8, %1" : "+v"(x[k]) : "v"(y)))
DEFR(r_add3, DI, asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(x[k]) : "v"(y)))
DEFR(r_cvtpki16, DI, asm volatile("v_cvt_pk_i16_i32 %0, %0, %1" : "+v"(x[k]) : "v"(y)))
DEFR(r_dot4, DI, asm volatile("v_dot4_i32_i8 %0, %0, %1, %0" : "+v"(x[k]) : "v"(y)))
DEFR(r_fmaf, DF, asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[k]) : "v"(y)))
DEFR(r_mulf, DF, asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[k]) : "v"(y)))
DEFR(r_floorf, DF, asm volatile("v_floor_f32 %0, %0" : "+v"(x[k])))
DEFR(r_rndnef, DF, asm volatile("v_rndne_f32 %0, %0" : "+v"(x[k])))
DEFR(r_cvtf32i32, DF, asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(x[k])))
DEFR(r_cvti32f32, DF, asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(x[k])))
DEFR(r_cvtpku8, DF, asm volatile("v_cvt_pk_u8_f32 %0, %0, 1, %0" : "+v"(x[k])))
DEFR(r_rcpf, DF, asm volatile("v_rcp_f32 %0, %0" : "+v"(x[k])))
DEFR(r_sqrtf, DF, asm volatile("v_sqrt_f32 %0, %0" : "+v"(x[k])))
DEFR(r_ldexpf, DF, asm volatile("v_ldexp_f32 %0, %0, 1" : "+v"(x[k])))
DEFR(r_fmad, DD, asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x[k]) : "v"(y)))
DEFR(r_muld, DD, asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[k]) : "v"(y)))
DEFR(r_addd, DD, asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[k]) : "v"(y)))
DEFR(r_cvtf64i32, DD, { int t = (int)threadIdx.x + k; asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(x[k]) : "v"(t)); })
DEFR(r_cvtf64f32, DD, { float t = (float)threadIdx.x + k; asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(x[k]) : "v"(t)); })
DEFR(r_rndned, DD, asm volatile("v_rndne_f64 %0, %0" : "+v"(x[k])))

template <typename K>
void rate(const char *name, K kern, int *buf, unsigned long long *clk) {
    const int n = 2048;
    dim3 grid(256 * 4), block(256);
    hipLaunchKernelGGL(kern, grid, block, 0, 0, buf, clk, 16);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(kern, grid, block, 0, 0, buf, clk, n);
    hipDeviceSynchronize();
    static unsigned long long h[1024];
    hipMemcpy(h, clk, sizeof(unsigned long long) * 1024, hipMemcpyDeviceToHost);
    double cyc = 0;
    for (int i = 0; i < 1024; ++i) cyc += (double)h[i];
    // a wave's cycles for n*8 instructions, 4 waves share the SIMD
    printf("%-14s %6.2f cyc per wave-instruction per SIMD\n", name, cyc / 1024 / ((double)n * 8) / 4.0);
}

int main() {
    int *buf;
    unsigned long long *clk;
    hipMalloc(&buf, 256 * 8 * 256 * 4);
    hipMalloc(&clk, sizeof(unsigned long long) * 2 * 256 * 8);
#define R3(MIX, name)                                                                          \
    run<MIX, 1, 1, 2>(buf, clk, name); run<MIX, 2, 1, 2>(buf, clk, name); run<MIX, 1, 0, 2>(buf, clk, name); \
    run<MIX, 2, 1, 1>(buf, clk, name);
    if (getenv("ACC_SWEEP")) {   // dependent chains: how many independent accumulators a wave needs, 1..3 waves per SIMD
        peak<32, 1, 1>(buf, clk, 1); peak<32, 2, 1>(buf, clk, 1); peak<32, 1, 3>(buf, clk, 1); peak<32, 2, 3>(buf, clk, 1);
        peak<32, 2, 2>(buf, clk, 1); peak<32, 4, 3>(buf, clk, 1); peak<16, 2, 3>(buf, clk, 1); peak<16, 4, 3>(buf, clk, 1);
        return 0;
    }
    for (int seed = 0; seed < 2; ++seed) {
        peak<32, 4, 1>(buf, clk, seed); peak<32, 4, 2>(buf, clk, seed); peak<32, 4, 4>(buf, clk, seed);
        peak<16, 8, 1>(buf, clk, seed); peak<16, 16, 1>(buf, clk, seed); peak<16, 8, 2>(buf, clk, seed);
        peak<16, 16, 2>(buf, clk, seed); peak<16, 8, 4>(buf, clk, seed); peak<16, 16, 4>(buf, clk, seed);
    }
    run<0, 0, 1, 1>(buf, clk, "none");
    run<0, 0, 1, 2>(buf, clk, "none");
    run<0, 0, 1, 4>(buf, clk, "none");
    R3(1, "f64fma") R3(6, "f64mad") R3(2, "int") R3(3, "f32") R3(4, "add4") R3(5, "i24")
    rate("v_add_u32", r_add, buf, clk);
    rate("v_mul_hi_i32", r_mulhi, buf, clk);
    rate("v_mul_lo_u32", r_mullo, buf, clk);
    rate("v_mul_hi_i24", r_mulhi24, buf, clk);
    rate("v_mul_i32_i24", r_mul24, buf, clk);
    rate("v_mad_i32_i24", r_mad24, buf, clk);
    rate("v_med3_i32", r_med3, buf, clk);
    rate("v_perm_b32", r_perm, buf, clk);
    rate("v_ashrrev", r_ashr, buf, clk);
    rate("v_lshl_or", r_lshlor, buf, clk);
    rate("v_add3", r_add3, buf, clk);
    rate("v_cvt_pk_i16", r_cvtpki16, buf, clk);
    rate("v_dot4_i32_i8", r_dot4, buf, clk);
    rate("v_fma_f32", r_fmaf, buf, clk);
    rate("v_mul_f32", r_mulf, buf, clk);
    rate("v_floor_f32", r_floorf, buf, clk);
    rate("v_rndne_f32", r_rndnef, buf, clk);
    rate("v_cvt_f32_i32", r_cvtf32i32, buf, clk);
    rate("v_cvt_i32_f32", r_cvti32f32, buf, clk);
    rate("v_cvt_pk_u8_f32", r_cvtpku8, buf, clk);
    rate("v_rcp_f32", r_rcpf, buf, clk);
    rate("v_sqrt_f32", r_sqrtf, buf, clk);
    rate("v_ldexp_f32", r_ldexpf, buf, clk);
    rate("v_fma_f64", r_fmad, buf, clk);
    rate("v_mul_f64", r_muld, buf, clk);
    rate("v_add_f64", r_addd, buf, clk);
    rate("v_cvt_f64_i32", r_cvtf64i32, buf, clk);
    rate("v_cvt_f64_f32", r_cvtf64f32, buf, clk);
    rate("v_rndne_f64", r_rndned, buf, clk);
    return 0;
}
