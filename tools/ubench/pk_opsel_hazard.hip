// The one-LSB LayerNorm differences of rounds 3-5, reduced to one instruction.  tools/ubench/ln_s1_asm (assembly-level bisection of the
// real victim beside the real aggressor) leaves exactly one necessary-and-sufficient class:
//     v_pk_add_f32 D, A, B op_sel:[0,1] op_sel_hi:[1,0]         D.lo = A.lo + B.hi,  D.hi = A.hi + B.lo   (the cross-half add)
// failing only while another wave of the SIMD issues MFMAs.  Here: victims that execute ONE packed form and its two scalar adds on the
// same inputs and count the iterations whose bits differ, beside workgroups that do nothing but v_mfma_i32_32x32x32_i8.
//   form 0: D = B (in place, cross)      v_pk_add_f32 v[0:1], v[2:3], v[0:1] op_sel:[0,1] op_sel_hi:[1,0]   — as the compiler emitted it
//   form 1: D = A = B (in place, cross)  v_pk_add_f32 v[0:1], v[0:1], v[0:1] op_sel:[0,1] op_sel_hi:[1,0]
//   form 2: D distinct from A, B (cross)
//   form 3: D = B, no op_sel (control)   v_pk_add_f32 v[0:1], v[2:3], v[0:1]
//   form 4: D = A (in place, cross)      v_pk_add_f32 v[0:1], v[0:1], v[2:3] op_sel:[0,1] op_sel_hi:[1,0]
//   form 5: v_pk_mul_f32, D = B, cross
//   form 6: v_pk_add_f32 op_sel:[1,0] op_sel_hi:[0,1]  (the other diagonal: D.lo = A.hi + B.lo, D.hi = A.lo + B.hi)
//   form 7: v_pk_add_f32 op_sel:[0,1] only             (D.lo = A.lo + B.hi, D.hi = A.hi + B.hi: only the low lane selects a high half)
//   form 8: v_pk_add_f32 op_sel_hi:[1,0] only          (D.lo = A.lo + B.lo, D.hi = A.hi + B.lo: the broadcast form the compiler uses everywhere)
//   form 9: form 2 with s_nop 4 in front
//   form 10: v_pk_fma_f32 D = C, op_sel:[0,1,0] op_sel_hi:[1,0,1]
//   form 11: v_pk_add_u16 op_sel:[0,1] on 32-bit registers (the one VOP3P op_sel form the shipped library contains, 708 times: the 16-bit
//            halves of ONE dword, not the second dword of a pair)
// Build: hipcc --offload-arch=gfx950 -O3 -Wno-unused-value tools/ubench/pk_opsel_hazard.hip -o tools/ubench/pk_opsel_hazard
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// MODE 0: short-lived workgroups (8 MFMAs, one store: the K = 48 patch-embedding GEMM's profile); MODE 1: long-lived (n x 8 MFMAs)
__global__ __launch_bounds__(256) void mfma_aggressor(int *out, int n) {
    v16i acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0;
    v4i a = {(int)threadIdx.x, 3, 5, 7}, b = {11, (int)blockIdx.x, 13, 17};
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(b, a, acc[i], 0, 0, 0);
        }
    }
    int s = 0;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s ^= acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// the same long-lived aggressor with the accumulators pinned: ACC_AGPR = true in a[...] (what hipcc chose above and in gemm_nt_kernel),
// false in v[...] (what gemm_glds_kernel's register budget makes it choose — the GEMM beside which the LayerNorm never failed)
template <bool ACC_AGPR>
__global__ __launch_bounds__(256) void mfma_aggressor_pinned(int *out, int n) {
    v16i acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0;
    v4i a = {(int)threadIdx.x, 3, 5, 7}, b = {11, (int)blockIdx.x, 13, 17};
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (ACC_AGPR) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0\n\tv_mfma_i32_32x32x32_i8 %0, %2, %1, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
            else asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0\n\tv_mfma_i32_32x32x32_i8 %0, %2, %1, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15");
    int s = 0;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s ^= acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__device__ __forceinline__ float sadd(float x, float y) { float r; asm volatile("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; }
__device__ __forceinline__ float smul(float x, float y) { float r; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; }

struct Bad { int count; unsigned got[2], want[2], a[2], b[2]; int it, lane, first_it_hist[4]; };

template <int FORM>
__global__ __launch_bounds__(128) void victim(const float *in, Bad *bad, int n) {
    v2f a = {in[threadIdx.x], in[threadIdx.x + 128]}, b = {in[(threadIdx.x * 7) & 255], in[(threadIdx.x * 13) & 255]};
    int nbad = 0;
    for (int it = 0; it < n; ++it) {
        v2f d, a0 = a, b0 = b;
        float w0, w1;
        if (FORM == 0) { d = b; asm volatile("v_pk_add_f32 %0, %1, %0 op_sel:[0,1] op_sel_hi:[1,0]" : "+v"(d) : "v"(a)); w0 = sadd(a0[0], b0[1]); w1 = sadd(a0[1], b0[0]); }
        if (FORM == 1) { d = a; asm volatile("v_pk_add_f32 %0, %0, %0 op_sel:[0,1] op_sel_hi:[1,0]" : "+v"(d)); w0 = sadd(a0[0], a0[1]); w1 = sadd(a0[1], a0[0]); }
        if (FORM == 2) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=&v"(d) : "v"(a), "v"(b)); w0 = sadd(a0[0], b0[1]); w1 = sadd(a0[1], b0[0]); }
        if (FORM == 3) { d = b; asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(d) : "v"(a)); w0 = sadd(a0[0], b0[0]); w1 = sadd(a0[1], b0[1]); }
        if (FORM == 4) { d = a; asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "+v"(d) : "v"(b)); w0 = sadd(a0[0], b0[1]); w1 = sadd(a0[1], b0[0]); }
        if (FORM == 5) { d = b; asm volatile("v_pk_mul_f32 %0, %1, %0 op_sel:[0,1] op_sel_hi:[1,0]" : "+v"(d) : "v"(a)); w0 = smul(a0[0], b0[1]); w1 = smul(a0[1], b0[0]); }
        if (FORM == 6) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=&v"(d) : "v"(a), "v"(b)); w0 = sadd(a0[1], b0[0]); w1 = sadd(a0[0], b0[1]); }
        if (FORM == 7) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=&v"(d) : "v"(a), "v"(b)); w0 = sadd(a0[0], b0[1]); w1 = sadd(a0[1], b0[1]); }
        if (FORM == 8) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=&v"(d) : "v"(a), "v"(b)); w0 = sadd(a0[0], b0[0]); w1 = sadd(a0[1], b0[0]); }
        if (FORM == 9) { asm volatile("s_nop 4\n\tv_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=&v"(d) : "v"(a), "v"(b)); w0 = sadd(a0[0], b0[1]); w1 = sadd(a0[1], b0[0]); }
        if (FORM == 10) { d = b; asm volatile("v_pk_fma_f32 %0, %1, %1, %0 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "+v"(d) : "v"(a)); w0 = fmaf(a0[0], a0[1], b0[0]); w1 = fmaf(a0[1], a0[0], b0[1]); }
        if (FORM == 11) {
            unsigned ua = __float_as_uint(a[0]), ub = __float_as_uint(b[1]), ud;
            asm volatile("v_pk_add_u16 %0, %1, %2 op_sel:[0,1]" : "=&v"(ud) : "v"(ua), "v"(ub));
            const unsigned want = (((ua & 0xffffu) + (ub >> 16)) & 0xffffu) | (((ua >> 16) + (ub >> 16)) << 16);
            d[0] = __uint_as_float(ud); d[1] = 0.f; w0 = __uint_as_float(want); w1 = 0.f;
        }
        asm volatile("" : "+v"(w0), "+v"(w1));
        if (__float_as_uint(d[0]) != __float_as_uint(w0) || __float_as_uint(d[1]) != __float_as_uint(w1)) {
            if (nbad == 0 && atomicAdd(&bad->count, 1) == 0) {
                bad->got[0] = __float_as_uint(d[0]); bad->got[1] = __float_as_uint(d[1]); bad->want[0] = __float_as_uint(w0); bad->want[1] = __float_as_uint(w1);
                bad->it = it; bad->lane = threadIdx.x;
                bad->a[0] = __float_as_uint(a0[0]); bad->a[1] = __float_as_uint(a0[1]); bad->b[0] = __float_as_uint(b0[0]); bad->b[1] = __float_as_uint(b0[1]);
            } else if (nbad) atomicAdd(&bad->count, 1);
            atomicAdd(&bad->first_it_hist[it == 0 ? 0 : (it < 16 ? 1 : (it < 256 ? 2 : 3))], 1);
            ++nbad;
        }
        a[0] = a[0] * 1.0001f + 1e-3f; a[1] = a[1] * 0.9999f + 2e-3f;
        b[0] = b[0] * 0.99995f + 1e-4f; b[1] = b[1] * 1.00005f - 1e-4f;
    }
}

// Second experiment: WHO wrote the high dword of src1, and how many wait states before the packed add reads it through op_sel.  Fixed
// physical registers so that nothing but the named instructions sits between producer and consumer: v[100:101] = B (v101 first holds a
// recognisable stale value, then the producer overwrites it), v[104:105] = A, v[102:103] = D.
//   PROD 0: v_mov_b32   1: v_pk_mul_f32   2: v_pk_add_f32   3: global_load_dword + s_waitcnt vmcnt(0)   4: ds_read_b32 + s_waitcnt lgkmcnt(0)
//   GAP: wait states (s_nop) between the producer (or its s_waitcnt) and the v_pk_add_f32
#define PG_ASM(PRODSTR, GAPSTR)                                                                                                              \
    asm volatile("v_mov_b32 v100, %[b0]\n\tv_mov_b32 v101, %[stale]\n\tv_mov_b32 v104, %[a0]\n\tv_mov_b32 v105, %[a1]\n\ts_nop 7\n\ts_nop 7\n\t"          \
                 PRODSTR GAPSTR "v_pk_add_f32 v[102:103], v[104:105], v[100:101] op_sel:[0,1] op_sel_hi:[1,0]\n\t"                             \
                 "s_nop 7\n\tv_mov_b32 %[d0], v102\n\tv_mov_b32 %[d1], v103\n\tv_mov_b32 %[r0], v100\n\tv_mov_b32 %[r1], v101"                         \
                 : [d0] "=&v"(d0), [d1] "=&v"(d1), [r0] "=&v"(r0), [r1] "=&v"(r1)                                                            \
                 : [b0] "v"(b0), [stale] "v"(stale), [a0] "v"(a[0]), [a1] "v"(a[1]), [n1] "v"(x[1]), [x] "v"(x), [y] "v"(y), [gp] "v"(gp), [la] "v"(la) \
                 : "v100", "v101", "v102", "v103", "v104", "v105", "memory")
struct BadPG { int any, stale_lo, zero_lo, events; unsigned long long mask; unsigned a0, b1, got, it; };
template <int PROD, int GAP>
__global__ __launch_bounds__(128) void victim_pg(const float *in, BadPG *bad, int n) {
    __shared__ float lds[256];
    lds[threadIdx.x] = in[threadIdx.x]; lds[threadIdx.x + 128] = in[threadIdx.x + 128];
    __syncthreads();
    v2f a = {in[threadIdx.x], in[threadIdx.x + 128]}, x = {in[(threadIdx.x * 7) & 255], in[(threadIdx.x * 13) & 255]}, y = {1.25f, 0.75f};
    const float *gp = in + ((threadIdx.x * 5) & 255);
    const unsigned la = (unsigned)(size_t)(lds + ((threadIdx.x * 3) & 255));
    const float stale = 1024.0f, b0 = 7.0f;
    int nany = 0, nstale = 0, nzero = 0;
    for (int it = 0; it < n; ++it) {
        float d0, d1, r0, r1;
        if constexpr (PROD == 0 && GAP == 0) PG_ASM("v_mov_b32 v101, %[n1]\n\t", "");
        if constexpr (PROD == 0 && GAP == 1) PG_ASM("v_mov_b32 v101, %[n1]\n\t", "s_nop 0\n\t");
        if constexpr (PROD == 0 && GAP == 2) PG_ASM("v_mov_b32 v101, %[n1]\n\t", "s_nop 1\n\t");
        if constexpr (PROD == 0 && GAP == 4) PG_ASM("v_mov_b32 v101, %[n1]\n\t", "s_nop 3\n\t");
        if constexpr (PROD == 0 && GAP == 8) PG_ASM("v_mov_b32 v101, %[n1]\n\t", "s_nop 7\n\t");
        if constexpr (PROD == 1 && GAP == 0) PG_ASM("v_pk_mul_f32 v[100:101], %[x], %[y]\n\t", "");
        if constexpr (PROD == 1 && GAP == 1) PG_ASM("v_pk_mul_f32 v[100:101], %[x], %[y]\n\t", "s_nop 0\n\t");
        if constexpr (PROD == 1 && GAP == 2) PG_ASM("v_pk_mul_f32 v[100:101], %[x], %[y]\n\t", "s_nop 1\n\t");
        if constexpr (PROD == 1 && GAP == 4) PG_ASM("v_pk_mul_f32 v[100:101], %[x], %[y]\n\t", "s_nop 3\n\t");
        if constexpr (PROD == 1 && GAP == 8) PG_ASM("v_pk_mul_f32 v[100:101], %[x], %[y]\n\t", "s_nop 7\n\t");
        if constexpr (PROD == 2 && GAP == 0) PG_ASM("v_pk_add_f32 v[100:101], %[x], %[y]\n\t", "");
        if constexpr (PROD == 2 && GAP == 1) PG_ASM("v_pk_add_f32 v[100:101], %[x], %[y]\n\t", "s_nop 0\n\t");
        if constexpr (PROD == 2 && GAP == 2) PG_ASM("v_pk_add_f32 v[100:101], %[x], %[y]\n\t", "s_nop 1\n\t");
        if constexpr (PROD == 2 && GAP == 4) PG_ASM("v_pk_add_f32 v[100:101], %[x], %[y]\n\t", "s_nop 3\n\t");
        if constexpr (PROD == 2 && GAP == 8) PG_ASM("v_pk_add_f32 v[100:101], %[x], %[y]\n\t", "s_nop 7\n\t");
        if constexpr (PROD == 3 && GAP == 0) PG_ASM("global_load_dword v101, %[gp], off\n\ts_waitcnt vmcnt(0)\n\t", "");
        if constexpr (PROD == 3 && GAP == 1) PG_ASM("global_load_dword v101, %[gp], off\n\ts_waitcnt vmcnt(0)\n\t", "s_nop 0\n\t");
        if constexpr (PROD == 3 && GAP == 2) PG_ASM("global_load_dword v101, %[gp], off\n\ts_waitcnt vmcnt(0)\n\t", "s_nop 1\n\t");
        if constexpr (PROD == 3 && GAP == 4) PG_ASM("global_load_dword v101, %[gp], off\n\ts_waitcnt vmcnt(0)\n\t", "s_nop 3\n\t");
        if constexpr (PROD == 3 && GAP == 8) PG_ASM("global_load_dword v101, %[gp], off\n\ts_waitcnt vmcnt(0)\n\t", "s_nop 7\n\t");
        if constexpr (PROD == 4 && GAP == 0) PG_ASM("ds_read_b32 v101, %[la]\n\ts_waitcnt lgkmcnt(0)\n\t", "");
        if constexpr (PROD == 4 && GAP == 1) PG_ASM("ds_read_b32 v101, %[la]\n\ts_waitcnt lgkmcnt(0)\n\t", "s_nop 0\n\t");
        if constexpr (PROD == 4 && GAP == 2) PG_ASM("ds_read_b32 v101, %[la]\n\ts_waitcnt lgkmcnt(0)\n\t", "s_nop 1\n\t");
        if constexpr (PROD == 4 && GAP == 4) PG_ASM("ds_read_b32 v101, %[la]\n\ts_waitcnt lgkmcnt(0)\n\t", "s_nop 3\n\t");
        if constexpr (PROD == 4 && GAP == 8) PG_ASM("ds_read_b32 v101, %[la]\n\ts_waitcnt lgkmcnt(0)\n\t", "s_nop 7\n\t");
        // r0 / r1: B as it stands after the block (what the producer wrote); the packed add must have seen exactly that
        const float w0 = sadd(a[0], r1), w1 = sadd(a[1], r0), ws = sadd(a[0], stale);
        const bool wrong = __float_as_uint(d0) != __float_as_uint(w0) || __float_as_uint(d1) != __float_as_uint(w1);
        const unsigned long long m = __ballot(wrong);
        if (m) {
            if ((threadIdx.x & 63) == __ffsll((long long)m) - 1 && atomicAdd(&bad->events, 1) == 0) {
                bad->mask = m; bad->a0 = __float_as_uint(a[0]); bad->b1 = __float_as_uint(r1); bad->got = __float_as_uint(d0); bad->it = it;
            }
            if (wrong) { ++nany; nstale += __float_as_uint(d0) == __float_as_uint(ws); nzero += __float_as_uint(d0) == __float_as_uint(a[0]) && __float_as_uint(d1) == __float_as_uint(w1); }
        }
        a[0] = a[0] * 1.0001f + 1e-3f; a[1] = a[1] * 0.9999f + 2e-3f;
        x[0] = x[0] * 0.99995f + 1e-4f; x[1] = x[1] * 1.00005f - 1e-4f;
        y[0] = y[0] * 1.00001f; y[1] = y[1] * 0.99999f;
    }
    if (nany) { atomicAdd(&bad->any, nany); atomicAdd(&bad->stale_lo, nstale); atomicAdd(&bad->zero_lo, nzero); }
}

template <int FORM>
static void launch_victim(const float *in, Bad *bad, hipStream_t s) { victim<FORM><<<2048, 128, 0, s>>>(in, bad, 2000); }

int main(int argc, char **argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 20;
    float *in; Bad *bad; int *sink;
    hipMalloc(&in, 1024); hipMalloc(&bad, sizeof(Bad)); hipMalloc(&sink, 8192 * 256 * 4);
    float h[256];
    srand(5);
    for (int i = 0; i < 256; ++i) h[i] = 1.0f + (rand() % 100000) * 1e-5f;
    hipMemcpy(in, h, 1024, hipMemcpyHostToDevice);
    hipStream_t sv, sa[4];
    hipStreamCreateWithFlags(&sv, hipStreamNonBlocking);
    for (auto &s : sa) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const char *forms[12] = {"v_pk_add_f32 D=B cross", "v_pk_add_f32 D=A=B cross", "v_pk_add_f32 D distinct cross", "v_pk_add_f32 D=B plain", "v_pk_add_f32 D=A cross", "v_pk_mul_f32 D=B cross",
                             "v_pk_add_f32 op_sel:[1,0] hi:[0,1]", "v_pk_add_f32 op_sel:[0,1] only", "v_pk_add_f32 op_sel_hi:[1,0] only", "s_nop 4 + cross", "v_pk_fma_f32 op_sel:[0,1,0]", "v_pk_add_u16 op_sel:[0,1]"};
    const char *aggr[5] = {"none", "short-lived MFMA workgroups", "long-lived MFMA workgroups", "long-lived MFMA, acc in AGPRs", "long-lived MFMA, acc in VGPRs"};
    for (int ag = 0; ag < 5; ag += (ag == 0 ? 2 : 1))
        for (int form = 0; form < 12; ++form) {
            hipMemset(bad, 0, sizeof(Bad));
            hipDeviceSynchronize();
            for (int r = 0; r < reps; ++r) {
                for (int k = 0; k < 4; ++k) {
                    if (ag == 1) for (int q = 0; q < 8; ++q) mfma_aggressor<<<784, 256, 0, sa[k]>>>(sink, 1);
                    if (ag == 2) mfma_aggressor<<<1024, 256, 0, sa[k]>>>(sink, 4000);
                    if (ag == 3) mfma_aggressor_pinned<true><<<1024, 256, 0, sa[k]>>>(sink, 4000);
                    if (ag == 4) mfma_aggressor_pinned<false><<<1024, 256, 0, sa[k]>>>(sink, 4000);
                }
                for (int k = 0; k < 4; ++k) {
                    switch (form) {
                        case 0: launch_victim<0>(in, bad, sv); break; case 1: launch_victim<1>(in, bad, sv); break; case 2: launch_victim<2>(in, bad, sv); break;
                        case 3: launch_victim<3>(in, bad, sv); break; case 4: launch_victim<4>(in, bad, sv); break; case 5: launch_victim<5>(in, bad, sv); break;
                        case 6: launch_victim<6>(in, bad, sv); break; case 7: launch_victim<7>(in, bad, sv); break; case 8: launch_victim<8>(in, bad, sv); break;
                        case 9: launch_victim<9>(in, bad, sv); break; case 10: launch_victim<10>(in, bad, sv); break; default: launch_victim<11>(in, bad, sv); break;
                    }
                }
                hipDeviceSynchronize();
            }
            Bad hb;
            hipMemcpy(&hb, bad, sizeof(Bad), hipMemcpyDeviceToHost);
            printf("aggressor %-30s victim %-30s: %9d mismatching iterations of %.3g", aggr[ag], forms[form], hb.count, (double)reps * 4 * 2048 * 128 * 2000);
            if (hb.count) printf("  (by iteration: 0 %d, 1-15 %d, 16-255 %d, later %d)  first: it %d lane %d a = {%08x, %08x} b = {%08x, %08x} got {%08x, %08x} want {%08x, %08x}", hb.first_it_hist[0], hb.first_it_hist[1], hb.first_it_hist[2], hb.first_it_hist[3], hb.it, hb.lane, hb.a[0], hb.a[1], hb.b[0], hb.b[1], hb.got[0], hb.got[1], hb.want[0], hb.want[1]);
            printf("\n");
        }
    // ---- second experiment: producer x gap, beside the long-lived MFMA workgroups
    BadPG *pg;
    hipMalloc(&pg, sizeof(BadPG));
    const char *prods[5] = {"v_mov_b32 (one-pass VALU)", "v_pk_mul_f32 (two-pass VALU, writes both halves)", "v_pk_add_f32 (two-pass VALU, writes both halves)", "global_load_dword + s_waitcnt vmcnt(0)", "ds_read_b32 + s_waitcnt lgkmcnt(0)"};
    const int gapv[5] = {0, 1, 2, 4, 8};
    for (int P = 0; P < 5; ++P)
        for (int gi = 0; gi < 5; ++gi) {
            const int G = gapv[gi];
            hipMemset(pg, 0, sizeof(BadPG));
            hipDeviceSynchronize();
            for (int r = 0; r < reps; ++r) {
                for (int k = 0; k < 4; ++k) mfma_aggressor<<<1024, 256, 0, sa[k]>>>(sink, 4000);
                for (int k = 0; k < 4; ++k) {
                if (P == 0 && G == 0) victim_pg<0, 0><<<2048, 128, 0, sv>>>(in, pg, 2000);
                if (P == 0 && G == 1) victim_pg<0, 1><<<2048, 128, 0, sv>>>(in, pg, 2000);
                if (P == 0 && G == 2) victim_pg<0, 2><<<2048, 128, 0, sv>>>(in, pg, 2000);
                if (P == 0 && G == 4) victim_pg<0, 4><<<2048, 128, 0, sv>>>(in, pg, 2000);
                if (P == 0 && G == 8) victim_pg<0, 8><<<2048, 128, 0, sv>>>(in, pg, 2000);
                if (P == 1 && G == 0) victim_pg<1, 0><<<2048, 128, 0, sv>>>(in, pg, 2000);
                if (P == 1 && G == 1) victim_pg<1, 1><<<2048, 128, 0, sv>>>(in, pg, 2000);
                if (P == 1 && G == 2) victim_pg<1, 2><<<2048, 128, 0, sv>>>(in, pg, 2000);
                if (P == 1 && G == 4) victim_pg<1, 4><<<2048, 128, 0, sv>>>(in, pg, 2000);
                if (P == 1 && G == 8) victim_pg<1, 8><<<2048, 128, 0, sv>>>(in, pg, 2000);
                if (P == 2 && G == 0) victim_pg<2, 0><<<2048, 128, 0, sv>>>(in, pg, 2000);
                if (P == 2 && G == 1) victim_pg<2, 1><<<2048, 128, 0, sv>>>(in, pg, 2000);
                if (P == 2 && G == 2) victim_pg<2, 2><<<2048, 128, 0, sv>>>(in, pg, 2000);
                if (P == 2 && G == 4) victim_pg<2, 4><<<2048, 128, 0, sv>>>(in, pg, 2000);
                if (P == 2 && G == 8) victim_pg<2, 8><<<2048, 128, 0, sv>>>(in, pg, 2000);
                if (P == 3 && G == 0) victim_pg<3, 0><<<2048, 128, 0, sv>>>(in, pg, 2000);
                if (P == 3 && G == 1) victim_pg<3, 1><<<2048, 128, 0, sv>>>(in, pg, 2000);
                if (P == 3 && G == 2) victim_pg<3, 2><<<2048, 128, 0, sv>>>(in, pg, 2000);
                if (P == 3 && G == 4) victim_pg<3, 4><<<2048, 128, 0, sv>>>(in, pg, 2000);
                if (P == 3 && G == 8) victim_pg<3, 8><<<2048, 128, 0, sv>>>(in, pg, 2000);
                if (P == 4 && G == 0) victim_pg<4, 0><<<2048, 128, 0, sv>>>(in, pg, 2000);
                if (P == 4 && G == 1) victim_pg<4, 1><<<2048, 128, 0, sv>>>(in, pg, 2000);
                if (P == 4 && G == 2) victim_pg<4, 2><<<2048, 128, 0, sv>>>(in, pg, 2000);
                if (P == 4 && G == 4) victim_pg<4, 4><<<2048, 128, 0, sv>>>(in, pg, 2000);
                if (P == 4 && G == 8) victim_pg<4, 8><<<2048, 128, 0, sv>>>(in, pg, 2000);
                }
                hipDeviceSynchronize();
            }
            BadPG hb;
            hipMemcpy(&hb, pg, sizeof(BadPG), hipMemcpyDeviceToHost);
            printf("src1.hi written by %-50s, %d wait states: %6d wrong of %.3g: %d = A.lo + the stale value, %d = exactly A.lo (src1.hi read as 0) with D.hi right; %d wave-instructions hit, first: lanes %016llx it %u A.lo %08x B.hi %08x got %08x\n",
                   prods[P], G, hb.any, (double)reps * 4 * 2048 * 128 * 2000, hb.stale_lo, hb.zero_lo, hb.events, hb.mask, hb.it, hb.a0, hb.b1, hb.got);
        }
    return 0;
}
