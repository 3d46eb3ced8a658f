// ivit_swin.h — kernels specific to the Swin path (reference models/swin_quant.py).
#pragma once
#include "ivit_device.h"

// ---------------------------------------------------------------------------
// IntSoftmax fed by `attn + mask` (swin_quant.py:151-156): the float mask (0 / -100.0) is added
// to fl(Q*s) BEFORE the division by s, so masked inputs are non-integers and the row max must be
// taken on the fp32 values.  Row r of the flattened [B_, H, n] rows uses
// mask[(r / (H*n)) % nW][r % n][:].  mask == nullptr: plain Shiftmax.  One wavefront per row.
__global__ __launch_bounds__(256) void shiftmax_masked_kernel(const int8_t *__restrict__ x, long long rows, int n,
                                                              int ld_in, float s, int out_bits,
                                                              const float *__restrict__ mask, int nW, int H,
                                                              uint16_t *__restrict__ out, int ld_out) {
    extern __shared__ __attribute__((aligned(16))) char dsmem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *er = reinterpret_cast<float *>(dsmem) + (size_t)wave * n;
    const long long row = (long long)blockIdx.x * 4 + wave;
    if (row >= rows) return;
    const int8_t *xp = x + row * ld_in;
    const float *mr = mask ? mask + (((row / ((long long)H * n)) % nW) * n + (row % n)) * n : nullptr;
    const RcpC sr = rcp_prepare(s);
    float mx = -INFINITY;
    for (int j = lane; j < n; j += 64) {
        float X = (float)xp[j] * s;
        if (mr) X = X + mr[j];
        float xt = lean_div(X, sr);
        er[j] = xt;
        mx = fmaxf(mx, xt);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    const float x0 = floorf(-1.0f / s);
    const RcpC x0r = rcp_prepare(x0);
    const float nx0 = 15.0f * x0;
    for (int j = lane; j < n; j += 64) er[j] = shift_exp_c(er[j] - mx, x0r, nx0, 15);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    float S = torch_order_sum32(n, lane & 31, [&](int idx) { return er[idx]; });
    const float F = recip_factor(S);
    const float div = ldexpf(1.0f, out_bits - 32);
    uint16_t *op = out + row * ld_out;
    for (int j = lane; j < n; j += 64) op[j] = (uint16_t)(int)floorf((er[j] * F) * div);
}

// ---------------------------------------------------------------------------
// QuantAct with an identity that repeats with period `id_period` elements (the relative
// position bias [H, N, N] broadcast over windows, swin_quant.py:149).
template <int BITS>
__global__ __launch_bounds__(256) void requant_bcast_kernel(const int32_t *__restrict__ z, ivit_dyadic dy,
                                                            const int32_t *__restrict__ z_id, long long id_period,
                                                            ivit_dyadic dy_id, void *__restrict__ out,
                                                            long long total) {
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long stride = (long long)gridDim.x * 256;
    const double c = dy.m * dy.r, ci = dy_id.m * dy_id.r;
    for (; i < total; i += stride) {
        double o = __builtin_rint((double)z_id[i % id_period] * ci) + __builtin_rint((double)z[i] * c);
        int v = clamp_b<BITS>(o);
        if (BITS == 8) reinterpret_cast<int8_t *>(out)[i] = (int8_t)v;
        else reinterpret_cast<int16_t *>(out)[i] = (int16_t)v;
    }
}

// ---------------------------------------------------------------------------
// Swin head: AdaptiveAvgPool1d(1) over L tokens then QuantAct (swin_quant.py:553-555).
// The reference averages fl(Q*s) in fp32 and the next QuantAct takes round(fl(mean/s)); with
// L odd the exact quotient sum(Q)/L is never within 1/(2L) of a tie, far outside fp32 noise,
// so z = rne(sum(Q)/L) in integers (pinned against the fp32 restatement in the oracle).
__global__ __launch_bounds__(256) void avgpool_requant_kernel(const int8_t *__restrict__ x, int B, int L, int C,
                                                              ivit_dyadic dy, int8_t *__restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * C) return;
    const int b = i / C, c = i - b * C;
    int sum = 0;
    for (int l = 0; l < L; ++l) sum += (int)x[((long long)b * L + l) * C + c];
    const int z = (int)__builtin_rint((double)sum / (double)L);
    out[i] = (int8_t)rq_c((double)z, dy.m * dy.r, -128, 127);
}

// ---------------------------------------------------------------------------
// I-LayerNorm whose two sums follow torch's order for a TOKEN-contiguous input (Swin stage 0:
// the activation keeps the layout of flatten(2).transpose(1,2), so torch reduces over a strided
// dim: ATen vectorized_outer_sum).  Per token: sequential accumulation over channels with the
// 16-step cascade; the last (L mod 32) tokens of an image use 4 interleaved accumulators.
// One thread per token, rows staged (fp32, padded) in LDS; 64 tokens per 64-thread block.
__device__ __forceinline__ float cascade_seq_sum_dev(const float *x, int n, int stride, bool sq, float mean) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int i = 0;
    for (; i + 16 <= n;) {
        for (int j = 0; j < 16; ++j, ++i) {
            float v = x[i * stride];
            if (sq) { v = v - mean; v = v * v; }
            a0 += v;
        }
        a1 += a0; a0 = 0.f;
        if ((i & 0xF0) != 0) continue;
        a2 += a1; a1 = 0.f;
        if ((i & 0xF00) != 0) continue;
        a3 += a2; a2 = 0.f;
    }
    for (; i < n; ++i) {
        float v = x[i * stride];
        if (sq) { v = v - mean; v = v * v; }
        a0 += v;
    }
    a0 += a1; a0 += a2; a0 += a3;
    return a0;
}
__device__ __forceinline__ float strided_order_sum(const float *x, int n, bool ilp4, bool sq, float mean) {
    if (!ilp4) return cascade_seq_sum_dev(x, n, 1, sq, mean);
    const int size_ilp = n >> 2;
    float ps0 = cascade_seq_sum_dev(x, size_ilp, 4, sq, mean), ps1 = cascade_seq_sum_dev(x + 1, size_ilp, 4, sq, mean);
    float ps2 = cascade_seq_sum_dev(x + 2, size_ilp, 4, sq, mean), ps3 = cascade_seq_sum_dev(x + 3, size_ilp, 4, sq, mean);
    for (int i = size_ilp * 4; i < n; ++i) {
        float v = x[i];
        if (sq) { v = v - mean; v = v * v; }
        ps0 += v;
    }
    ps0 += ps1; ps0 += ps2; ps0 += ps3;
    return ps0;
}

template <bool OUT8>
__global__ __launch_bounds__(64) void layernorm_tokenorder_kernel(const int16_t *__restrict__ x, long long rows, int C,
                                                                  float s, const float *__restrict__ bias_int,
                                                                  const float *__restrict__ sc,
                                                                  const ivit_dyadic *__restrict__ dy, int L,
                                                                  void *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char dsmem[];
    float *tile = reinterpret_cast<float *>(dsmem);   // [64][C + 1]
    const int LD = C + 1;
    const int tid = threadIdx.x;
    const long long row0 = (long long)blockIdx.x * 64;
    const RcpC sr = rcp_prepare(s);
    // coalesced load of up to 64 rows
    for (long long e = tid; e < (long long)64 * C; e += 64) {
        const int r = (int)(e / C), c = (int)(e - (long long)r * C);
        const long long gr = row0 + r;
        tile[r * LD + c] = gr < rows ? requotient_c((float)x[gr * C + c], sr) : 0.f;
    }
    __syncthreads();
    const long long row = row0 + tid;
    float *xr = tile + tid * LD;
    if (row < rows) {
        const bool ilp4 = (row % L) >= (L / 32) * 32;
        const float sum = strided_order_sum(xr, C, ilp4, false, 0.f);
        const float mean = rintf(sum / (float)C);
        const float var = strided_order_sum(xr, C, ilp4, true, mean);
        float k = 65536.0f;
        for (int n = 0; n < 10; ++n) k = floorf((k + floorf(var / k)) * 0.5f);
        const float F = floorf((1.0f / k) * 2147483648.0f);
        for (int c = 0; c < C; ++c) {
            const float y = xr[c] - mean;
            const float yi = floorf((y * F) * 0.5f);
            const float o = yi + bias_int[c];
            const float scv = sc[c];
            xr[c] = rintf((o * scv) / scv);
        }
    }
    __syncthreads();
    for (long long e = tid; e < (long long)64 * C; e += 64) {
        const int r = (int)(e / C), c = (int)(e - (long long)r * C);
        const long long gr = row0 + r;
        if (gr < rows) {
            const float zv = tile[r * LD + c];
            if (OUT8) reinterpret_cast<int8_t *>(out)[gr * C + c] = (int8_t)rq_c((double)zv, dy[c].m * dy[c].r, -128, 127);
            else reinterpret_cast<float *>(out)[gr * C + c] = zv;
        }
    }
}
