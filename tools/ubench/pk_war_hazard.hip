// Does a packed-fp32 VALU instruction protect its SOURCE registers against the next instruction's write (WAR) when the
// SIMD is shared with MFMA-issuing waves?  (Round 4: layernorm_reg_kernel<192, 1> mis-rounded beside GEMM workgroups; its
// reduction is  v_mov_b32_dpp v55 <- ...; v_pk_add_f32 v[52:53], v[52:53], v[54:55]; v_mov_b32_dpp v55 <- ... .)
// Victim: per iteration  acc(2) += {b0, b1} with a v_pk_add_f32, immediately followed by DPP movs that overwrite b0 / b1
// with the next values; the same sum is kept by plain v_add_f32 on other registers.  Aggressor: MFMA loops on other
// streams, launched so that their workgroups share the victim's CUs.  Reports victim waves whose packed sum != plain sum.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/pk_war_hazard.hip -o tools/ubench/pk_war_hazard
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));

// KIND 0: MFMA only; 1: MFMA + LDS fragment reads; 2: MFMA + LDS reads + fp64 requant arithmetic; 3: fp64 arithmetic + LDS, no MFMA.
// 64 KB of dynamic LDS per workgroup: two aggressor workgroups per CU (16 waves), so that victim workgroups co-reside
template <int KIND>
__global__ __launch_bounds__(256) void aggressor(int *out, int n) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    v4i a, b;
    for (int e = 0; e < 4; ++e) { a[e] = threadIdx.x * 2654435761u + e; b[e] = threadIdx.x * 40503u + e * 977; }
    for (int i = threadIdx.x; i < 4096; i += 256) reinterpret_cast<v4i *>(lds)[i] = a;
    __syncthreads();
    v16i c[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) c[i][r] = 0;
    double acc64 = 0.0;
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (KIND == 1 || KIND == 2 || KIND == 3) a = reinterpret_cast<const v4i *>(lds)[(threadIdx.x + 64 * (m + it)) & 4095];
            if (KIND != 3) c[m & 3] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c[m & 3], 0, 0, 0);
            if (KIND >= 2) acc64 = __builtin_fma((double)(a[0] + it), 1.0000001, acc64);
        }
    }
    int s = (int)acc64;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += c[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NOPS>
__global__ __launch_bounds__(128) void victim(const float *in, int *bad, int n) {
    float x0 = in[threadIdx.x], x1 = in[threadIdx.x + 128];
    v2f accp = {0.f, 0.f};
    float a0 = 0.f, a1 = 0.f;
    int nbad = 0;
    for (int it = 0; it < n; ++it) {
        // new operands by DPP (quad broadcasts of changing lanes), consumed by a packed add, then overwritten at once
        float c0, c1, n0, n1;
#define PKWAR_BODY(NOP)                                                                                   \
        asm volatile(                                                                                     \
            "v_mov_b32_dpp v100, %[x0] quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
            "v_mov_b32_dpp v101, %[x1] quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
            "s_nop 1\n\t"                                                                                \
            "v_mov_b32 %[c0], v100\n\t"                                                                  \
            "v_mov_b32 %[c1], v101\n\t"                                                                  \
            "v_pk_add_f32 %[acc], %[acc], v[100:101]\n\t" NOP                                            \
            "v_mov_b32_dpp v100, %[x0] quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
            "v_mov_b32_dpp v101, %[x1] quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
            "s_nop 1\n\t"                                                                                \
            "v_mov_b32 %[n0], v100\n\t"                                                                  \
            "v_mov_b32 %[n1], v101\n\t"                                                                  \
            : [acc] "+v"(accp), [c0] "=&v"(c0), [c1] "=&v"(c1), [n0] "=&v"(n0), [n1] "=&v"(n1)            \
            : [x0] "v"(x0), [x1] "v"(x1) : "v100", "v101")
        if (NOPS == 0) PKWAR_BODY(""); else PKWAR_BODY("s_nop 7\n\t");
        float bp[2] = {n0, n1};
        a0 += c0; a1 += c1;
        x0 = x0 * 1.0001f + bp[0] * 1e-7f; x1 = x1 * 0.9999f + bp[1] * 1e-7f;
        if (__float_as_int(accp[0]) != __float_as_int(a0) || __float_as_int(accp[1]) != __float_as_int(a1)) { ++nbad; accp[0] = a0; accp[1] = a1; }
    }
    if (nbad) atomicAdd(bad, nbad);
}


// The reduction of layernorm_reg_kernel<192, 1> as hipcc emitted it (profiles/README.md round 4): two accumulator chains
// (v21, v22), quad broadcasts into a register PAIR, packed adds straight behind the DPP movs that write the pair, the DPP
// sources rewritten two instructions behind their last DPP read.  MODE 0: as emitted; 1: s_nop 7 after every instruction
// (reference); 2: scalar adds instead of the packed ones.
template <int MODE>
__global__ __launch_bounds__(128) void victim2(const float *in, int *bad, int n, float *dbg) {
    float x[12];
    for (int i = 0; i < 12; ++i) x[i] = in[(threadIdx.x + 17 * i) & 255];
    int nbad = 0;
    for (int it = 0; it < n; ++it) {
        float r0, r1, q0, q1;
#define SEQ(NP, ADD0, ADD1, ADD2)                                                                           \
        asm volatile(                                                                                        \
            "v_add_f32 v21, 0, %[a0]\n\t" NP "v_add_f32 v22, 0, %[a1]\n\t" NP                              \
            "v_add_f32 v21, %[a2], v21\n\t" NP "v_add_f32 v22, %[a3], v22\n\t" NP                          \
            "v_add_f32 v21, %[a4], v21\n\t" NP "v_add_f32 v22, %[a5], v22\n\t" NP                          \
            "s_nop 0\n\t"                                                                                    \
            "v_mov_b32_dpp v53, v21 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" NP     \
            "v_mov_b32_dpp v52, v22 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" NP     \
            "v_mov_b32_dpp v55, v21 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" NP     \
            "v_mov_b32_dpp v54, v22 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" NP     \
            ADD0 NP                                                                                          \
            "v_mov_b32_dpp v55, v21 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" NP     \
            "v_mov_b32_dpp v54, v22 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" NP     \
            ADD1 NP                                                                                          \
            "v_mov_b32_dpp v55, v21 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" NP     \
            "v_mov_b32_dpp v54, v22 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" NP     \
            "v_add_f32 v21, 0, %[b0]\n\t" NP "v_add_f32 v22, 0, %[b1]\n\t" NP                              \
            "v_add_f32 v21, %[b2], v21\n\t" NP "v_add_f32 v22, %[b3], v22\n\t" NP                          \
            "v_add_f32 v21, %[b4], v21\n\t" NP "v_add_f32 v22, %[b5], v22\n\t" NP                          \
            ADD2 NP                                                                                          \
            "s_nop 1\n\t"                                                                                    \
            "v_mov_b32 %[r0], v52\n\t v_mov_b32 %[r1], v53\n\t v_mov_b32 %[q0], v21\n\t v_mov_b32 %[q1], v22\n\t" \
            : [r0] "=&v"(r0), [r1] "=&v"(r1), [q0] "=&v"(q0), [q1] "=&v"(q1)                                \
            : [a0] "v"(x[0]), [a1] "v"(x[1]), [a2] "v"(x[2]), [a3] "v"(x[3]), [a4] "v"(x[4]), [a5] "v"(x[5]), \
              [b0] "v"(x[6]), [b1] "v"(x[7]), [b2] "v"(x[8]), [b3] "v"(x[9]), [b4] "v"(x[10]), [b5] "v"(x[11]) \
            : "v21", "v22", "v52", "v53", "v54", "v55")
#define PK "v_pk_add_f32 v[52:53], v[52:53], v[54:55]\n\t"
#define SC "v_add_f32 v52, v52, v54\n\t v_add_f32 v53, v53, v55\n\t"
        if (MODE == 0) SEQ("", PK, PK, PK);
        else if (MODE == 1) SEQ("s_nop 7\n\t", PK, PK, PK);
        else SEQ("", SC, SC, SC);
        // reference by shuffles
        const float s0 = x[0] + 0.f, s1 = x[1] + 0.f;
        const float t21 = x[4] + (x[2] + s0), t22 = x[5] + (x[3] + s1);
        const int l0 = (threadIdx.x & 63) & ~3;
        float e1 = __shfl(t21, l0), e0 = __shfl(t22, l0);
        e1 += __shfl(t21, l0 + 1); e0 += __shfl(t22, l0 + 1);
        e1 += __shfl(t21, l0 + 2); e0 += __shfl(t22, l0 + 2);
        e1 += __shfl(t21, l0 + 3); e0 += __shfl(t22, l0 + 3);
        const float u21 = x[10] + (x[8] + (x[6] + 0.f)), u22 = x[11] + (x[9] + (x[7] + 0.f));
        if (__float_as_int(r0) != __float_as_int(e0) || __float_as_int(r1) != __float_as_int(e1) ||
            __float_as_int(q0) != __float_as_int(u21) || __float_as_int(q1) != __float_as_int(u22)) ++nbad;
        for (int i = 0; i < 12; ++i) x[i] = x[i] * 1.0003f + (float)((it * 7 + i) & 15) * 0.125f + r0 * 1e-6f;
    }
    if (nbad) atomicAdd(bad, nbad);
    if (dbg) dbg[blockIdx.x * 128 + threadIdx.x] = x[0];
}

int main(int argc, char **argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 40;
    float *in; int *bad, *sink;
    hipMalloc(&in, 1024); hipMalloc(&bad, 8); hipMalloc(&sink, 256 * 4096 * 4);
    float h[256];
    for (int i = 0; i < 256; ++i) h[i] = 1.0f + (rand() % 1000) * 1e-3f;
    hipMemcpy(in, h, 1024, hipMemcpyHostToDevice);
    hipStream_t sv, sa[3];
    hipStreamCreate(&sv);
    for (auto &s : sa) hipStreamCreate(&s);
    hipFuncSetAttribute((const void *)aggressor<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void *)aggressor<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void *)aggressor<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void *)aggressor<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int kind = -1; kind < 4; ++kind)
        for (int mode = 0; mode < 3; ++mode) {
            hipMemset(bad, 0, 8);
            hipDeviceSynchronize();
            for (int r = 0; r < reps; ++r) {
                // aggressors first and long-lived (512 workgroups x 2 streams = 2 per CU), victims stream in beside them
                for (int st = 0; st < 2 && kind >= 0; ++st) {
                    if (kind == 0) aggressor<0><<<512, 256, 65536, sa[st]>>>(sink, 3000);
                    if (kind == 1) aggressor<1><<<512, 256, 65536, sa[st]>>>(sink, 3000);
                    if (kind == 2) aggressor<2><<<512, 256, 65536, sa[st]>>>(sink, 3000);
                    if (kind == 3) aggressor<3><<<512, 256, 65536, sa[st]>>>(sink, 3000);
                }
                for (int k = 0; k < 4; ++k) {
                    if (mode == 0) victim2<0><<<2048, 128, 0, sv>>>(in, bad, 1000, nullptr);
                    else if (mode == 1) victim2<1><<<2048, 128, 0, sv>>>(in, bad, 1000, nullptr);
                    else victim2<2><<<2048, 128, 0, sv>>>(in, bad, 1000, nullptr);
                }
                hipDeviceSynchronize();
            }
            int hb = 0;
            hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
            printf("aggressor kind %d (-1 none, 0 MFMA, 1 +LDS reads, 2 +fp64, 3 fp64+LDS only), victim mode %d (0 emitted, 1 s_nop 7, 2 scalar adds): %d mismatching iterations in %d x 4 launches\n", kind, mode, hb, reps);
        }
    return 0;
}
