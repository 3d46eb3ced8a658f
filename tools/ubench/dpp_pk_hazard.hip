// Does v_mov_b32_dpp interleaved with v_pk_add_f32 (the instruction pattern of the undispatched layernorm_reg_kernel<C, 1>)
// always give the result of the same sequence with wait states between the instructions?  Run beside an MFMA-heavy kernel
// on other streams (the condition under which the LayerNorm kernel produced one-LSB differences).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/dpp_pk_hazard.hip -o tools/ubench/dpp_pk_hazard
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// quad sums of two values per lane, the way finish() of the S = 1 kernel was compiled: t0 = bcast0(a)+bcast1(a)+bcast2(a)+bcast3(a)
template <bool SAFE>
__global__ __launch_bounds__(128, 4) void k_quad(const float *in, float *out, int iters) {
    const int tid = blockIdx.x * 128 + threadIdx.x;
    float x = in[tid * 2], y = in[tid * 2 + 1], accx = 0.f, accy = 0.f;
    for (int it = 0; it < iters; ++it) {
        float ax, ay, r0, r1;
        // ax = x + small, ay = y + small (fresh VALU results feeding the DPP reads)
        if (SAFE) {
            asm volatile(
                "v_add_f32 %2, %4, %6\n\tv_add_f32 %3, %5, %6\n\ts_nop 4\n\t"
                "v_mov_b32_dpp v47, %2 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 4\n\t"
                "v_mov_b32_dpp v49, %2 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 4\n\t"
                "v_mov_b32_dpp v46, %3 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 4\n\t"
                "v_mov_b32_dpp v48, %3 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 4\n\t"
                "v_pk_add_f32 v[46:47], v[46:47], v[48:49]\n\ts_nop 4\n\t"
                "v_mov_b32_dpp v49, %2 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 4\n\t"
                "v_mov_b32_dpp v48, %3 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 4\n\t"
                "v_pk_add_f32 v[46:47], v[46:47], v[48:49]\n\ts_nop 4\n\t"
                "v_mov_b32_dpp v49, %2 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 4\n\t"
                "v_mov_b32_dpp v48, %3 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 4\n\t"
                "v_pk_add_f32 v[46:47], v[46:47], v[48:49]\n\ts_nop 4\n\t"
                "v_mov_b32 %0, v47\n\tv_mov_b32 %1, v46\n\t"
                : "=v"(r0), "=v"(r1), "=&v"(ax), "=&v"(ay) : "v"(x), "v"(y), "v"(accx) : "v46", "v47", "v48", "v49");
        } else {
            asm volatile(
                "v_add_f32 %2, %4, %6\n\tv_add_f32 %3, %5, %6\n\t"
                "v_mov_b32_dpp v47, %2 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                "v_mov_b32_dpp v49, %2 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                "v_mov_b32_dpp v46, %3 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                "v_mov_b32_dpp v48, %3 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                "v_pk_add_f32 v[46:47], v[46:47], v[48:49]\n\t"
                "v_mov_b32_dpp v49, %2 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                "v_mov_b32_dpp v48, %3 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                "v_pk_add_f32 v[46:47], v[46:47], v[48:49]\n\t"
                "v_mov_b32_dpp v49, %2 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                "v_mov_b32_dpp v48, %3 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                "v_pk_add_f32 v[46:47], v[46:47], v[48:49]\n\t"
                "s_nop 4\n\tv_mov_b32 %0, v47\n\tv_mov_b32 %1, v46\n\t"
                : "=v"(r0), "=v"(r1), "=&v"(ax), "=&v"(ay) : "v"(x), "v"(y), "v"(accx) : "v46", "v47", "v48", "v49");
        }
        accx = accx * 0.5f + r0 * 1e-3f;
        accy = accy * 0.5f + r1 * 1e-3f;
        x = x * 1.0001f + 0.37f; y = y * 0.9999f - 0.11f;
    }
    out[tid * 2] = accx; out[tid * 2 + 1] = accy;
}
// neighbour: MFMA + LDS heavy workgroups (large static LDS, many registers) on every CU
__global__ __launch_bounds__(256, 2) void k_noise(int *out, int iters) {
    __shared__ int lds[12 * 1024];
    for (int i = threadIdx.x; i < 12 * 1024; i += 256) lds[i] = i * 2654435761u;
    __syncthreads();
    v16i c[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) c[i][r] = 0;
    v4i a = {(int)threadIdx.x * 977, 3, 5, 7}, b = {11, (int)threadIdx.x * 131, 17, 19};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c[i], 0, 0, 0);
        a[0] ^= lds[(threadIdx.x * 4 + it) & (12 * 1024 - 1)];
    }
    int s = 0;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += c[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    const int nblk = 2048, n = nblk * 128 * 2, iters = 2000;
    std::vector<float> h(n);
    srand(1); for (auto &v : h) v = (float)(rand() % 20001 - 10000) * 1.37f;
    float *in, *o_safe, *o_fast[8]; int *nz;
    hipMalloc(&in, n * 4); hipMalloc(&o_safe, n * 4); hipMalloc(&nz, 4096 * 256 * 4);
    hipMemcpy(in, h.data(), n * 4, hipMemcpyHostToDevice);
    hipStream_t st[8]; for (auto &s : st) hipStreamCreate(&s);
    for (int i = 0; i < 8; ++i) hipMalloc(&o_fast[i], n * 4);
    k_quad<true><<<nblk, 128>>>(in, o_safe, iters);
    hipDeviceSynchronize();
    std::vector<float> ref(n), got(n);
    hipMemcpy(ref.data(), o_safe, n * 4, hipMemcpyDeviceToHost);
    for (int mode = 0; mode < 2; ++mode) {                       // 0: alone, 1: beside MFMA / LDS heavy neighbours
        long long bad = 0, launches = 0;
        for (int rep = 0; rep < 20; ++rep) {
            for (int i = 0; i < 8; ++i) {
                if (mode == 1 && (i & 1)) k_noise<<<1024, 256, 0, st[i]>>>(nz, 20000);
                else k_quad<false><<<nblk, 128, 0, st[i]>>>(in, o_fast[i], iters);
            }
            hipDeviceSynchronize();
            for (int i = 0; i < 8; ++i) {
                if (mode == 1 && (i & 1)) continue;
                hipMemcpy(got.data(), o_fast[i], n * 4, hipMemcpyDeviceToHost);
                ++launches;
                for (int j = 0; j < n; ++j) if (got[j] != ref[j]) { ++bad; break; }
            }
        }
        printf("%s: %lld of %lld launches of the back-to-back sequence differ from the wait-state sequence\n",
               mode ? "beside MFMA/LDS-heavy workgroups" : "alone", bad, launches);
    }
    return 0;
}
