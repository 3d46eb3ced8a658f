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

// LNT_ROWS tokens per 256-thread block: the loads (with fl(fl(Q*s)/s)) and the normalise / requant / store pass
// are spread over all threads; only the two order-sensitive sums and the integer square root of a token run
// on one thread (the first LNT_ROWS threads, one token each).  Per-channel divisors use the hoisted reciprocal (lean_div).
#define LNT_ROWS 32
// OUTM: 0 = fp32 z, 1 = int8 (per-channel QuantAct), 2 = int16 after TWO QuantActs (per-channel 16 bit, then the
// per-tensor dy2: PatchEmbed's norm -> qact(16) -> qact1(16), layers_quant.py:193-195 + swin_quant.py:543).
// XT: element type of x (int16, or int8 straight from the 8-bit QuantAct before the norm).
// CC: compile-time channel count (index arithmetic by constants), 0 = run-time.
template <int OUTM, int CC, typename XT = int16_t>
__global__ __launch_bounds__(256) void layernorm_tokenorder_kernel(const XT *__restrict__ x, long long rows, int C_rt,
                                                                   float s, const float *__restrict__ bias_int,
                                                                   const float *__restrict__ sc,
                                                                   const ivit_dyadic *__restrict__ dy, int L,
                                                                   void *__restrict__ out, ivit_dyadic dy2 = ivit_dyadic{0.0, 0.0}) {
    constexpr bool OUT8 = (OUTM == 1);
    const int C = CC ? CC : C_rt;
    extern __shared__ __attribute__((aligned(16))) char dsmem[];
    const int LD = C + 1;
    float *tile = reinterpret_cast<float *>(dsmem);            // [LNT_ROWS][C + 1]
    float *cSc = tile + LNT_ROWS * LD, *cY = cSc + C, *cB = cY + C;  // per-channel sc, refined 1/sc, bias_int
    float *rMean = cB + C, *rF = rMean + LNT_ROWS;                   // per-token mean and factor
    double *cC = reinterpret_cast<double *>(rF + LNT_ROWS + ((LNT_ROWS * LD + 3 * C + 2 * LNT_ROWS) & 1));   // 8-byte aligned
    const int tid = threadIdx.x;
    const long long row0 = (long long)blockIdx.x * LNT_ROWS;
    const float ys = rcp_rn(s);            // requotients by product + exact residual + one correction (ivit_layernorm.h)
    for (int c = tid; c < C; c += 256) {
        const float scv = sc[c];
        cSc[c] = scv;
        cY[c] = rcp_rn(scv);
        cB[c] = bias_int[c];
        if (OUTM != 0) cC[c] = dy[c].m * dy[c].r;
    }
    const double c2 = dy2.m * dy2.r;
    const int total = LNT_ROWS * C;
    for (int e = tid; e < total; e += 256) {
        const int r = e / C, c = e - r * C;
        const long long gr = row0 + r;
        tile[r * LD + c] = gr < rows ? requotient_m((float)x[gr * C + c], s, ys) : 0.f;
    }
    __syncthreads();
    if (tid < LNT_ROWS) {
        const long long row = row0 + tid;
        const float *xr = tile + tid * LD;
        float mean = 0.f, F = 0.f;
        if (row < rows) {
            const bool ilp4 = (row % L) >= (L / 32) * 32;
            const float sum = strided_order_sum(xr, C, ilp4, false, 0.f);
            mean = rintf(sum / (float)C);
            const float var = strided_order_sum(xr, C, ilp4, true, mean);
            float k = 65536.0f;
            for (int n = 0; n < 10; ++n) {          // the iteration is idempotent once it has converged
                const float kn = floorf((k + floorf(var / k)) * 0.5f);
                if (kn == k) break;
                k = kn;
            }
            F = floorf((1.0f / k) * 2147483648.0f);
        }
        rMean[tid] = mean;
        rF[tid] = F;
    }
    __syncthreads();
    for (int e = tid; e < total; e += 256) {
        const int r = e / C, c = e - r * C;
        const long long gr = row0 + r;
        if (gr >= rows) continue;
        const float y = tile[r * LD + c] - rMean[r];
        const float yi = floorf((y * rF[r]) * 0.5f);
        const float o = yi + cB[c];
        const float zv = rintf(requotient_m(o, cSc[c], cY[c]));
        if (OUT8) {
            reinterpret_cast<int8_t *>(out)[gr * C + c] = (int8_t)rq_c((double)zv, cC[c], -128, 127);
        } else if (OUTM == 2) {
            const int v16 = rq_c((double)zv, cC[c], -32768, 32767);
            reinterpret_cast<int16_t *>(out)[gr * C + c] = (int16_t)rq_c((double)v16, c2, -32768, 32767);
        } else {
            reinterpret_cast<float *>(out)[gr * C + c] = zv;
        }
    }
}

// The same operator for the stage-0 channel counts (C = 96, 128), 128 tokens per block, 8 channels per thread: 16-byte
// loads and 8 / 16-byte stores (the kernel above moves one element per thread-iteration: 2-byte loads, 1-byte stores,
// and runs its serial phase on 32 of 256 threads).  A thread keeps ONE channel group for the whole block — block size
// (C / 8) x 16 — so its eight per-channel constants live in registers; the order-sensitive sums and the integer square
// root run on 128 threads, one token each, on rows staged as fp32 with an odd pitch (conflict-free per-token walks).
#ifndef LNT8_ROWS
#define LNT8_ROWS 96       // 37 KB of LDS per block = 4 blocks per CU (128 rows: 3 blocks, the requant pass 10 % slower; 64: no better)
#endif
template <int OUTM, int CC, typename XT>
__global__ __launch_bounds__(CC / 8 * 16) void layernorm_tokenorder8_kernel(const XT *__restrict__ x, long long rows, float s,
                                                                            const float *__restrict__ bias_int,
                                                                            const float *__restrict__ sc,
                                                                            const ivit_dyadic *__restrict__ dy, int L,
                                                                            void *__restrict__ out, ivit_dyadic dy2) {
    static_assert(OUTM == 1 || OUTM == 2, "int8 or twice-requantised int16 output");
    constexpr int NC8 = CC / 8, NT = NC8 * 16, LD = CC + 1, NIT = LNT8_ROWS / 16;
    extern __shared__ __attribute__((aligned(16))) char dsmem[];
    float *tile = reinterpret_cast<float *>(dsmem);            // [LNT8_ROWS][CC + 1]
    float *cY = tile + LNT8_ROWS * LD, *rMean = cY + CC, *rF = rMean + LNT8_ROWS;
    const int tid = threadIdx.x, c8 = tid % NC8, rr = tid / NC8;
    const long long row0 = (long long)blockIdx.x * LNT8_ROWS;
    const float ys = rcp_rn(s);
    for (int c = tid; c < CC; c += NT) cY[c] = rcp_rn(sc[c]);
    float scv[8], bv[8];
    double cv[8];
    {
        const v4f s0 = *reinterpret_cast<const v4f *>(sc + c8 * 8), s1 = *reinterpret_cast<const v4f *>(sc + c8 * 8 + 4);
        const v4f b0 = *reinterpret_cast<const v4f *>(bias_int + c8 * 8), b1 = *reinterpret_cast<const v4f *>(bias_int + c8 * 8 + 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) { scv[k] = s0[k]; scv[4 + k] = s1[k]; bv[k] = b0[k]; bv[4 + k] = b1[k]; }
#pragma unroll
        for (int k = 0; k < 8; ++k) cv[k] = dy[c8 * 8 + k].m * dy[c8 * 8 + k].r;
    }
    const double c2 = dy2.m * dy2.r;
    // ---- rows -> fp32 tile, x = fl(fl(Q*s)/s)
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const int r = rr + 16 * i;
        const long long gr = row0 + r;
        float v[8];
        if (gr < rows) {
            if constexpr (sizeof(XT) == 2) {
                const v8s t = *reinterpret_cast<const v8s *>(x + gr * CC + c8 * 8);
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = requotient_m((float)t[k], s, ys);
            } else {
                const v2i t = *reinterpret_cast<const v2i *>(x + gr * CC + c8 * 8);
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = requotient_m((float)(int)(signed char)(t[k >> 2] >> (8 * (k & 3))), s, ys);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = 0.f;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) tile[r * LD + c8 * 8 + k] = v[k];
    }
    __syncthreads();
    float yv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) yv[k] = cY[c8 * 8 + k];
    // ---- per token: the two sums in torch's order for a token-contiguous input, integer square root
    if (tid < LNT8_ROWS) {
        const long long row = row0 + tid;
        const float *xr = tile + tid * LD;
        float mean = 0.f, F = 0.f;
        if (row < rows) {
            const bool ilp4 = (row % L) >= (L / 32) * 32;
            const float sum = strided_order_sum(xr, CC, ilp4, false, 0.f);
            mean = rintf(sum / (float)CC);
            const float var = strided_order_sum(xr, CC, ilp4, true, mean);
            float k = 65536.0f;
            for (int n = 0; n < 10; ++n) {
                const float kn = floorf((k + floorf(var / k)) * 0.5f);
                if (kn == k) break;
                k = kn;
            }
            F = floorf((1.0f / k) * 2147483648.0f);
        }
        rMean[tid] = mean;
        rF[tid] = F;
    }
    __syncthreads();
    // ---- normalise, requotient by the channel scale, requant(s), store
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const int r = rr + 16 * i;
        const long long gr = row0 + r;
        if (gr >= rows) continue;
        const float mean = rMean[r], F = rF[r];
        int o[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float y = tile[r * LD + c8 * 8 + k] - mean;
            const float yi = floorf((y * F) * 0.5f);
            const float zv = rintf(requotient_m(yi + bv[k], scv[k], yv[k]));
            if (OUTM == 1) o[k] = rq_c((double)zv, cv[k], -128, 127);
            else o[k] = rq_c((double)rq_c((double)zv, cv[k], -32768, 32767), c2, -32768, 32767);
        }
        if (OUTM == 1) {
            v2i w;
#pragma unroll
            for (int d = 0; d < 2; ++d)
                w[d] = (o[4 * d] & 0xff) | ((o[4 * d + 1] & 0xff) << 8) | ((o[4 * d + 2] & 0xff) << 16) | (o[4 * d + 3] << 24);
            *reinterpret_cast<v2i *>(reinterpret_cast<int8_t *>(out) + gr * CC + c8 * 8) = w;
        } else {
            v4i w;
#pragma unroll
            for (int d = 0; d < 4; ++d) w[d] = (o[2 * d] & 0xffff) | (o[2 * d + 1] << 16);
            *reinterpret_cast<v4i *>(reinterpret_cast<int16_t *>(out) + gr * CC + c8 * 8) = w;
        }
    }
}

// ---------------------------------------------------------------------------
// a12 fused windowed attention (WindowAttention.forward, swin_quant.py:121-169, between the qkv
// QuantAct and proj), window 7x7 (N = 49 tokens), head dim 32.  One wavefront per (image, window,
// head); cyclic shift, window partition and their inverses are index arithmetic on the natural
// [B, R, R, 3C] qkv tensor and [B, R*R, C] context tensor — no permuted copies.
//
//   S^T = K Q^T      v_mfma_i32_32x32x32_i8, K = dh = 32: lane holds query (lane & 31) and 16 keys
//                    per 32-key tile -> a query's row of 49 sits in 2 lanes x (16 + 9) registers
//   a   = clamp8(rq(clamp8(rq(S, dy_qk)), dy_a) + relb[h][q][k])      (qact_attn1, qact2 + bias)
//   P   = Shiftmax_8bit(a (+ shift mask))   torch's sum order for n = 49 is lane-local:
//         p[l] = ((((x[l]+x[32+l])+x[40+l])+x[8+l])+x[16+l])+x[24+l],  S = x[48]+p[0]+...+p[7]
//   O^T = V^T P^T    the P registers ARE the B fragments (same key permutation on both operands);
//         P <= 128 does not fit int8: two MFMAs, B = P - 64 and B = 64
//   out = clamp8(rq(O, dy_pv)) -> 16-byte stores at the token's natural position
typedef int v16i_sw __attribute__((ext_vector_type(16)));
struct WinAttnArgs {
    const int8_t *qkv;      // [B, R, R, 3, heads, 32]
    int8_t *ctx;            // [B, R*R, heads*32]
    const int16_t *relb;    // [heads, 49, 49]: rq(quantised relative position bias, dy_table -> qact2)
    int B, R, shift, heads;
    ivit_dyadic dy_qk, dy_a, dy_pv;
    float s;                // Shiftmax input scale (qact2)
    long long units;        // B * (R/7)^2 * heads
    // optional Shiftmax tables (ivit_amd.freeze.shiftmax_tables, as in AttnArgs) for the windows without a shift mask
    const uint16_t *aq;     // [nc][256]
    const float *et;        // [t_count]
    const uint8_t *cls;     // [256]
    int nc, t_count, dmin;
    int wpw;                // windows per wavefront (LUT form: the tables are staged once per block)
};

// LUT = false: a block = ONE head x 4 windows (4 wavefronts), the head's relative-position slab staged once for the four.
// LUT = true:  8 wavefronts x p.wpw windows each, so that the Shiftmax tables of the layer (up to 24 KB, the same for every
//              head and window) and the slab are staged once per 8 * wpw windows; windows under the shift mask (the float
//              -100 lives between the requotient's multiply and divide) keep the arithmetic shift-exp.
#ifndef WA_PROBE          // timing probes only (results invalid): 1 no shift-exp arithmetic, 2 no score gathers (requant only),
#define WA_PROBE 0         // 4 no global stores, 8 return after the operand loads
#endif
#define WA_FIXED(NW) ((NW) * (2048 + 64) + 4816 + 512 + 1024)
#ifndef WA_MINW
#define WA_MINW 6          // waves per SIMD the arithmetic kernel is compiled for (80 registers, three spilled dwords outside the score loops; 4: +3 %, 8: +18 %)
#endif
template <bool LUT>
__global__ __launch_bounds__(LUT ? 512 : 256, LUT ? 4 : WA_MINW) void window_attention_kernel(WinAttnArgs p) {
    constexpr int NW = LUT ? 8 : 4;
    extern __shared__ __attribute__((aligned(16))) char sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    char *sV = sm + wave * 2048;                                            // [64 keys][32 d]
    unsigned char *sReg = reinterpret_cast<unsigned char *>(sm + NW * 2048 + wave * 64);
    int16_t *sRel = reinterpret_cast<int16_t *>(sm + NW * (2048 + 64));          // [49][49], shared by the block
    int16_t *sTa = reinterpret_cast<int16_t *>(sm + NW * (2048 + 64) + 4816);    // rq(v, dy_a), v = -128..127
    float *sTx = reinterpret_cast<float *>(sm + NW * (2048 + 64) + 4816 + 512);  // fl(fl(a*s)/s)
    float *sT = reinterpret_cast<float *>(sm + WA_FIXED(NW));                    // LUT: exp table, offsets, classes
    unsigned short *sAQ = reinterpret_cast<unsigned short *>(sT + (LUT ? (p.t_count + 3) & ~3 : 0));
    unsigned char *sCls = reinterpret_cast<unsigned char *>(sAQ + (LUT ? p.nc * 256 : 0));
    const float s = p.s;
    const RcpC sr = rcp_prepare(s);
    if (tid < 256) {
        const double ca = p.dy_a.m * p.dy_a.r;
        sTa[tid] = (int16_t)(int)__builtin_rint((double)(tid - 128) * ca);
        sTx[tid] = requotient_c((float)(tid - 128), sr);
    }
    if (LUT) {
        const int n4 = p.t_count >> 2, a4 = p.nc * 32;          // whole 16-byte chunks (alignment checked by the launcher)
        for (int i = tid; i < n4; i += NW * 64) reinterpret_cast<v4i *>(sT)[i] = reinterpret_cast<const v4i *>(p.et)[i];
        if (tid < (p.t_count & 3)) sT[(n4 << 2) + tid] = p.et[(n4 << 2) + tid];
        // offsets staged as byte offsets into sT (x4; t_count <= 16384 keeps them in 16 bits)
        for (int i = tid; i < a4; i += NW * 64) {
            v4i t = reinterpret_cast<const v4i *>(p.aq)[i];
#pragma unroll
            for (int u = 0; u < 4; ++u) t[u] = (int)(((unsigned)t[u] & 0x3fff3fffu) << 2);
            *reinterpret_cast<v4i *>(reinterpret_cast<char *>(sAQ) + i * 16) = t;
        }
        if (tid < 64) reinterpret_cast<unsigned *>(sCls)[tid] = reinterpret_cast<const unsigned *>(p.cls)[tid];
    }
    // Block -> (window group, head): the heads of one window group sit 8 block ids apart, i.e. on ONE XCD and close in time.
    // A token's q | k | v row is 3 * heads * 32 contiguous bytes of which a block reads 32-byte pieces; with head-fastest block
    // ids the 3 ... 24 heads of a window ran on different XCDs and every L2 fetched the same lines (round 3 counters: 3.5x the
    // algorithmic bytes from HBM; the operand loads alone were 91 of the stage-0 launch's 215 us).
    const int xcd = (int)(blockIdx.x & 7), bq = (int)(blockIdx.x >> 3);
    const int head = bq % p.heads;
    const long long wgrp = (long long)(bq / p.heads) * 8 + xcd;
    if (wgrp * (LUT ? p.wpw : 1) * NW >= (long long)p.B * (p.R / 7) * (p.R / 7)) return;      // padding blocks of the last group of 8
    {
        const unsigned *rb = reinterpret_cast<const unsigned *>(p.relb + (long long)head * 2401);   // 2401 int16: 1200 dwords + 1
        const bool al = ((head * 2401) & 1) == 0;              // odd heads start on a 2-byte boundary
        // every load of the slab is issued before the first one is stored (as a load-store loop it was 5 / 10 serial memory
        // latencies per block)
        if (al) {
            constexpr int NI = (1200 + NW * 64 - 1) / (NW * 64);
            unsigned t[NI];
#pragma unroll
            for (int u = 0; u < NI; ++u) if (tid + u * NW * 64 < 1200) t[u] = rb[tid + u * NW * 64];
#pragma unroll
            for (int u = 0; u < NI; ++u) if (tid + u * NW * 64 < 1200) reinterpret_cast<unsigned *>(sRel)[tid + u * NW * 64] = t[u];
            if (tid == 0) sRel[2400] = p.relb[(long long)head * 2401 + 2400];
        } else {
            constexpr int NI = (2401 + NW * 64 - 1) / (NW * 64);
            int16_t t[NI];
#pragma unroll
            for (int u = 0; u < NI; ++u) if (tid + u * NW * 64 < 2401) t[u] = p.relb[(long long)head * 2401 + tid + u * NW * 64];
#pragma unroll
            for (int u = 0; u < NI; ++u) if (tid + u * NW * 64 < 2401) sRel[tid + u * NW * 64] = t[u];
        }
    }
    __syncthreads();
    const int R = p.R, nw = R / 7, C = p.heads * 32;
    const int wpw = LUT ? p.wpw : 1;
    for (int it = 0; it < wpw; ++it) {       // no workgroup barrier below: the wavefronts run their windows independently
    const long long wlin = (wgrp * wpw + it) * NW + wave;
    if (wlin >= (long long)p.B * nw * nw) return;
    if (LUT && it) {                         // this wavefront's V rows / regions of the previous window are dead now
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    const int win = (int)(wlin % (nw * nw)), b = (int)(wlin / (nw * nw));
    const int wi = win / nw, wj = win - wi * nw;
    auto tok_off = [&](int n) -> long long {                 // natural token index of window token n
        const int wy = n / 7, wx = n - wy * 7;
        int y = wi * 7 + wy + p.shift, x = wj * 7 + wx + p.shift;
        y = y >= R ? y - R : y;
        x = x >= R ? x - R : x;
        return ((long long)b * R + y) * R + x;
    };
    const bool masked = p.shift > 0 && (wi == nw - 1 || wj == nw - 1);

    // ---- operands: Q / K fragments straight from global, V rows and the bias slab through LDS
    v4i qf[2], kf[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int n = t * 32 + l31;
        qf[t] = v4i{0, 0, 0, 0};
        kf[t] = v4i{0, 0, 0, 0};
        if (n < 49) {
            const int8_t *row = p.qkv + tok_off(n) * (3 * C) + head * 32 + half * 16;
            qf[t] = *reinterpret_cast<const v4i *>(row);
            kf[t] = *reinterpret_cast<const v4i *>(row + C);
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int piece = lane + i * 64, n = piece >> 1, hh = piece & 1;    // 128 pieces of 16 B
        v4i v = {0, 0, 0, 0};
        if (n < 49) v = *reinterpret_cast<const v4i *>(p.qkv + tok_off(n) * (3 * C) + 2 * C + head * 32 + hh * 16);
        *reinterpret_cast<v4i *>(sV + n * 32 + hh * 16) = v;
    }
    if (lane < 49) {
        const int wy = lane / 7, wx = lane - wy * 7, ys = wi * 7 + wy, xs = wj * 7 + wx;
        const int ry = ys < R - 7 ? 0 : (ys < R - p.shift ? 1 : 2), rx = xs < R - 7 ? 0 : (xs < R - p.shift ? 1 : 2);
        sReg[lane] = (unsigned char)(ry * 3 + rx);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // V^T fragments (A operand of O^T = V^T P^T): row d = lane & 31, byte i of key tile kt = key
    // 32kt + (i&3) + 8(i>>2) + 4*half — the order in which this lane's S^T registers hold the keys
    v4i vf[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            unsigned word = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int key = 32 * kt + e + 8 * w + 4 * half;
                word |= (unsigned)(unsigned char)sV[key * 32 + l31] << (8 * e);
            }
            vf[kt][w] = (int)word;
        }

    if ((WA_PROBE & 8) && (vf[0][0] ^ vf[1][3] ^ qf[0][0] ^ kf[1][1]) != 0x12345678) return;
    const double c_qk = p.dy_qk.m * p.dy_qk.r, c_pv = p.dy_pv.m * p.dy_pv.r;
    const float x0 = floorf(-1.0f / s), nx0 = 15.0f * x0;
    const RcpC x0r = rcp_prepare(x0);
    const v4i c64 = {0x40404040, 0x40404040, 0x40404040, 0x40404040};

#pragma unroll 1
    for (int qt = 0; qt < 2; ++qt) {
        const int q = qt * 32 + l31;                       // this lane's query (rows >= 49 are padding)
        const bool qlive = q < 49;
        const int qq = qlive ? q : 48;
        v16i_sw acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0; acc1[r] = 0; }
        acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(kf[0], qf[qt], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(kf[1], qf[qt], acc1, 0, 0, 0);
        // valid keys: tile 0 all 16 registers; tile 1 registers 0..7 (keys 32..47) and, for half 0, r = 8 (key 48)
        float f[25];
        const int regq = sReg[qq];
        auto score = [&](int i, int &kk) -> int {         // a = clamp8(rq(clamp8(rq(S, dy_qk)), dy_a) + relb)
            const int kt = i < 16 ? 0 : 1, r = i < 16 ? i : i - 16;
            const int key = 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * half;      // < 49 except i == 24 on half 1
            kk = (i == 24 && half) ? 48 : key;
            const int z = kt ? acc1[r] : acc0[r];
            const int v = min(max(__double2loint((double)z * c_qk + 6755399441055744.0), -128), 127);
            if (WA_PROBE & 2) return v;
            return min(max((int)sTa[v + 128] + (int)sRel[qq * 49 + kk], -128), 127);
        };
        if (LUT && !masked) {
            // exp_int by table: e = et[aq[class(amax)][a] + max(a - amax, dmin) - dmin] (ivit_attention.h has the derivation);
            // fl(fl(a*s)/s) is monotone in a, so the row maximum is taken on the integers
            int av[25], amax = -128;
#pragma unroll
            for (int i = 0; i < 25; ++i) {
                int kk;
                av[i] = score(i, kk);
                if (!(i == 24 && half)) amax = max(amax, av[i]);
                if (i == 8 || i == 16) __builtin_amdgcn_sched_barrier(0);     // three batches of gathers: bounded live registers
            }
            amax = max(amax, __shfl_xor(amax, 32));
            typedef __attribute__((address_space(3))) const char wa_lds_c;
            const unsigned aqrow = (unsigned)(size_t)((wa_lds_c *)sAQ) + ((unsigned)sCls[amax + 128] * 256u + 128u) * 2u;
            const unsigned tb = (unsigned)(size_t)((wa_lds_c *)sT);
            const int qd = amax + p.dmin;
#pragma unroll
            for (int c0 = 0; c0 < 25; c0 += 13) {
                int e1[13];
#pragma unroll
                for (int i = c0; i < c0 + 13 && i < 25; ++i)
                    e1[i - c0] = (int)*reinterpret_cast<__attribute__((address_space(3))) const unsigned short *>((size_t)(aqrow + (unsigned)(av[i] << 1)));
#pragma unroll
                for (int i = c0; i < c0 + 13 && i < 25; ++i)
                    f[i] = *reinterpret_cast<__attribute__((address_space(3))) const float *>(
                        (size_t)(tb + (unsigned)e1[i - c0] + ((unsigned)max(av[i] - qd, 0) << 2)));
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            float mx = -INFINITY;
#pragma unroll
            for (int i = 0; i < 25; ++i) {
                int kk;
                const int a = score(i, kk);
                float xt;
                if (masked) {
                    float X = (float)a * s;
                    X = X + ((sReg[kk] != regq) ? -100.0f : 0.0f);
                    xt = lean_div(X, sr);
                } else {
                    xt = sTx[a + 128];
                }
                f[i] = xt;
                if (!(i == 24 && half)) mx = fmaxf(mx, xt);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32));
#pragma unroll
            for (int i = 0; i < 25; ++i) f[i] = (WA_PROBE & 1) ? f[i] - mx : shift_exp_nonpos(f[i] - mx, x0r, nx0, 15);
        }
        // torch-order row sum (n = 49), lane-local partials for l = 4*half + e
        float pl[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) pl[e] = ((((f[e] + f[16 + e]) + f[20 + e]) + f[4 + e]) + f[8 + e]) + f[12 + e];
        float fin = f[24];                                    // x[48] (half 0)
        fin = (((fin + pl[0]) + pl[1]) + pl[2]) + pl[3];
        const float lo = __shfl(fin, l31);                    // half 0's partial
        const float hi = (((lo + pl[0]) + pl[1]) + pl[2]) + pl[3];   // meaningful on half 1
        const float S = __shfl(hi, l31 + 32);
        const float F16 = recip_factor(S) * 5.9604644775390625e-08f;   // * 2^-24 (exact scaling)
        // probabilities (0..128) -> B fragments P - 64 in the lane's own key order
        v4i pf[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                unsigned word = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = w * 4 + e, i = kt ? 16 + r : r;
                    int P = 64;                                // padding keys: P - 64 = 0 ... V rows there are 0 anyway
                    if (kt == 0 || r < 8 || (r == 8 && !half)) P = (int)(f[i < 25 ? i : 24] * F16);   // >= 0: truncation is the floor
                    word |= (unsigned)((P - 64) & 0xff) << (8 * e);
                }
                pf[kt][w] = (int)word;
            }
        v16i_sw o;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = 0;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            o = __builtin_amdgcn_mfma_i32_32x32x32_i8(vf[kt], pf[kt], o, 0, 0, 0);
            o = __builtin_amdgcn_mfma_i32_32x32x32_i8(vf[kt], c64, o, 0, 0, 0);
        }
        // O^T[d][query]: lane = query, register quad g -> d = 8g + 4*half + (0..3)
        unsigned W[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            int ob[4];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                ob[e] = min(max(__double2loint((double)o[g * 4 + e] * c_pv + 6755399441055744.0), -128), 127);
            unsigned w01 = __builtin_amdgcn_perm((unsigned)ob[1], (unsigned)ob[0], 0x0c0c0400u);
            unsigned w23 = __builtin_amdgcn_perm((unsigned)ob[3], (unsigned)ob[2], 0x0c0c0400u);
            W[g] = __builtin_amdgcn_perm(w23, w01, 0x05040100u);
        }
        auto s02 = __builtin_amdgcn_permlane32_swap(W[0], W[2], false, false);
        auto s13 = __builtin_amdgcn_permlane32_swap(W[1], W[3], false, false);
        const v4i outv = {(int)s02[0], (int)s02[1], (int)s13[0], (int)s13[1]};   // d = 16*half .. +16
        if (qlive && (!(WA_PROBE & 4) || outv[0] == 0x12345678)) *reinterpret_cast<v4i *>(p.ctx + tok_off(q) * C + head * 32 + half * 16) = outv;
    }
    }
}

// ---------------------------------------------------------------------------
// PatchMerging gather (swin_quant.py:336-342): x [B, R, R, C] -> [B, R/2, R/2, 4C] with the channel
// blocks ordered (0::2,0::2), (1::2,0::2), (0::2,1::2), (1::2,1::2).  IN = int8 or int16 elements
// (after the first merge the stream is 8-bit); output int16 for the LayerNorm that follows.
template <typename IN>
__global__ __launch_bounds__(256) void patch_merge_gather_kernel(const IN *__restrict__ x, int B, int R, int C,
                                                                 int16_t *__restrict__ out) {
    // one thread per 8 consecutive output channels (C % 8 == 0: they come from one contiguous input run)
    const int R2 = R / 2, C8 = C / 8;
    const long long total = (long long)B * R2 * R2 * 4 * C8;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c8 = (int)(i % (4 * C8));
        const long long t = i / (4 * C8);
        const int xo = (int)(t % R2), yo = (int)((t / R2) % R2), b = (int)(t / ((long long)R2 * R2));
        const int blk = c8 / C8, c = (c8 - blk * C8) * 8;
        const int y = 2 * yo + (blk & 1), xx = 2 * xo + (blk >> 1);
        const IN *src = x + (((long long)b * R + y) * R + xx) * C + c;
        v8s o;
        if (sizeof(IN) == 2) {
            o = *reinterpret_cast<const v8s *>(src);
        } else {
            const v2i w = *reinterpret_cast<const v2i *>(src);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (short)(int8_t)(w[e >> 2] >> (8 * (e & 3)));
        }
        *reinterpret_cast<v8s *>(out + i * 8) = o;
    }
}

__global__ __launch_bounds__(256) void widen_i8_i16_kernel(const int8_t *__restrict__ x, int16_t *__restrict__ out,
                                                           long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        out[i] = (int16_t)x[i];
}

// ---------------------------------------------------------------------------
// a9 for narrow stages: Mlp.forward (layers_quant.py:144-153) + the block's closing QuantAct with identity
// (swin_quant.py:293-296) in ONE kernel for C = 96, hidden = 384 (Swin-T/S stage 0):
//     fc1 -> qact_gelu(8) -> ShiftGELU -> qact1(8) -> fc2 -> qact2(16) -> qact4(16, + identity)
// Stage 0 is HBM-bound as separate kernels (the hidden tensor is 4x the activations: 308 MB per launch at
// 256 images, written once and read twice); here it never leaves the LDS.  One persistent workgroup per CU
// keeps BOTH weight matrices resident in LDS (2 x 36 KB) and walks 64-token tiles:
//   S1  fc1 with v_mfma_i32_32x32x32_i8 (24 sub-tiles over 8 waves), requant to int8 straight into the
//       fc2 B-fragment layout, per-token max of the hidden row via LDS ds_max;
//   S2  one 256-byte ShiftGELU table row per token (selected by its row max) copied into LDS;
//   S3  table gathers in place;   S4  fc2 (6 sub-tiles, K = 384), per-channel requant to 16 bit, then the
//       identity requant-add and 16-byte stores.
// Operand tiles are stored chunk-major ([k/32][row][32 B]) so every fragment read is one contiguous 1 KB.
struct MlpFusedArgs {
    const int8_t *x;          // [M, 96]  LN2 output
    const int8_t *w1; const int32_t *b1; const ivit_dyadic *dy1;   // fc1 [384, 96] -> 8 bit
    const int8_t *tab;        // ShiftGELU(+requant) table [256][256]
    const int8_t *w2; const int32_t *b2; const ivit_dyadic *dy2;   // fc2 [96, 384] -> 16 bit
    ivit_dyadic dy_main, dy_res;
    const int16_t *residual;  // [M, 96]
    int16_t *out;             // [M, 96]
    long long M;
};

#define MF_C 96
#define MF_HD 384
#define MF_BM 64
#define MF_W1 0
#define MF_W2 (MF_W1 + MF_HD * MF_C)                 // 36864
#define MF_X (MF_W2 + MF_C * MF_HD)                  // 73728: 2 x 6144
#define MF_H (MF_X + 2 * MF_BM * MF_C)               // 86016: 24576
#define MF_ROWS (MF_H + MF_BM * MF_HD)               // 110592: 16384
#define MF_C1 (MF_ROWS + MF_BM * 256)                // 126976: 384 doubles
#define MF_B1 (MF_C1 + MF_HD * 8)                    // 130048: 384 ints
#define MF_C2 (MF_B1 + MF_HD * 4)                    // 131584: 96 doubles
#define MF_B2 (MF_C2 + MF_C * 8)                     // 132352: 96 ints
#define MF_MAX (MF_B2 + MF_C * 4)                    // 132736: 64 ints
#define MF_SMEM (MF_MAX + MF_BM * 4)                 // 132992

#define MF_THREADS 1024
__global__ __launch_bounds__(MF_THREADS) void swin_mlp_fused_kernel(MlpFusedArgs p) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    double *sC1 = reinterpret_cast<double *>(sm + MF_C1), *sC2 = reinterpret_cast<double *>(sm + MF_C2);
    int *sB1 = reinterpret_cast<int *>(sm + MF_B1), *sB2 = reinterpret_cast<int *>(sm + MF_B2);
    int *sMax = reinterpret_cast<int *>(sm + MF_MAX);

    // ---- one-off: weights -> LDS in fragment (chunk-major) layout, constants
    // W1's rows are placed so that MFMA row 8g + 4h + e of a 32-channel tile is hidden channel 16h + 4g + e: a lane's 16
    // fc1 outputs (registers 4g + e) are then 16 CONSECUTIVE channels and leave as one ds_write_b128 (round 4; four
    // ds_write_b32 at a 32-byte token pitch were 8-way bank conflicts, 29 % of the kernel's LDS time by its PMC run)
    for (int i = tid; i < MF_HD * MF_C / 16; i += MF_THREADS) {          // W1 [384][96]: 6 chunks of 16 B per row
        const int n = i / 6, c16 = i - n * 6, kc = c16 >> 1, hh = c16 & 1;
        const int c = n & 31, slot = (n & ~31) + 8 * ((c >> 2) & 3) + 4 * (c >> 4) + (c & 3);
        *reinterpret_cast<v4i *>(sm + MF_W1 + kc * (MF_HD * 32) + slot * 32 + hh * 16) =
            *reinterpret_cast<const v4i *>(p.w1 + n * MF_C + c16 * 16);
    }
    for (int i = tid; i < MF_C * MF_HD / 16; i += MF_THREADS) {          // W2 [96][384]: 24 chunks per row
        const int n = i / 24, c16 = i - n * 24, kc = c16 >> 1, hh = c16 & 1;
        *reinterpret_cast<v4i *>(sm + MF_W2 + kc * (MF_C * 32) + n * 32 + hh * 16) =
            *reinterpret_cast<const v4i *>(p.w2 + n * MF_HD + c16 * 16);
    }
    if (tid < MF_HD) { sC1[tid] = p.dy1[tid].m * p.dy1[tid].r; sB1[tid] = p.b1 ? p.b1[tid] : 0; }
    if (tid < MF_C) { sC2[tid] = p.dy2[tid].m * p.dy2[tid].r; sB2[tid] = p.b2 ? p.b2[tid] : 0; }
    const double cm = p.dy_main.m * p.dy_main.r, cr = p.dy_res.m * p.dy_res.r;

    const long long ntiles = (p.M + MF_BM - 1) / MF_BM;
    auto issue_x = [&](long long tile, int buf) {                 // 6 pieces of 1 KB: (kc, mt); waves 0..5
        if (wave < 6) {
            const int kc = wave >> 1, mt = wave & 1;
            const long long t = min(tile * MF_BM + mt * 32 + l31, p.M - 1);
            const int8_t *src = p.x + t * MF_C + kc * 32 + half * 16;
            char *dst = sm + MF_X + buf * (MF_BM * MF_C) + (kc * 2 + mt) * 1024;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        }
    };
    long long tile = blockIdx.x;
    if (tile < ntiles) issue_x(tile, 0);
    int buf = 0;
    for (; tile < ntiles; tile += gridDim.x, buf ^= 1) {
        if (tid < MF_BM) sMax[tid] = -128;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();                                          // X(tile) landed; H / rows free again
        if (tile + gridDim.x < ntiles) issue_x(tile + gridDim.x, buf ^ 1);
        // identity rows of this tile for S4 (waves 0..5): requested now, consumed four phases later
        v4i rsv[2] = {v4i{0, 0, 0, 0}, v4i{0, 0, 0, 0}};
        if (wave < 6) {
            const long long tok = tile * MF_BM + (wave & 1) * 32 + l31;
            if (tok < p.M) {
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    rsv[q] = *reinterpret_cast<const v4i *>(p.residual + tok * MF_C + (wave >> 1) * 32 + half * 16 + q * 8);
            }
        }
        const char *sX = sm + MF_X + buf * (MF_BM * MF_C);
        // ---- S1: fc1, 24 (n-tile, m-tile) units over the wavefronts
        for (int unit = wave; unit < 24; unit += MF_THREADS / 64) {
            const int nt = unit >> 1, mt = unit & 1;
            v16i_sw acc;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const v4i b4 = *reinterpret_cast<const v4i *>(sB1 + nt * 32 + half * 16 + g * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[g * 4 + e] = b4[e];
            }
#pragma unroll
            for (int kc = 0; kc < 3; ++kc) {
                const v4i wf = *reinterpret_cast<const v4i *>(sm + MF_W1 + kc * (MF_HD * 32) + (nt * 32 + l31) * 32 + half * 16);
                const v4i xf = *reinterpret_cast<const v4i *>(sX + (kc * 2 + mt) * 1024 + lane * 16);
                acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf, xf, acc, 0, 0, 0);
            }
            int mx = -128;
            v4i hw;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n0 = nt * 32 + half * 16 + g * 4;              // registers 4g .. 4g + 3: hidden channels n0 .. n0 + 3
                int o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] = min(max(rint_sat_i32((double)acc[g * 4 + e] * sC1[n0 + e]), -128), 127);
                    mx = max(mx, o[e]);
                }
                unsigned w01 = __builtin_amdgcn_perm((unsigned)o[1], (unsigned)o[0], 0x0c0c0400u);
                unsigned w23 = __builtin_amdgcn_perm((unsigned)o[3], (unsigned)o[2], 0x0c0c0400u);
                hw[g] = (int)__builtin_amdgcn_perm(w23, w01, 0x05040100u);
            }
            // hidden channels 16 half .. + 15 of token (mt*32 + l31): fc2 B-fragment layout [k/32][token][32 B]
            *reinterpret_cast<v4i *>(sm + MF_H + nt * (MF_BM * 32) + (mt * 32 + l31) * 32 + half * 16) = hw;
            atomicMax(&sMax[mt * 32 + l31], mx);
        }
        __syncthreads();
        // ---- S2: the table row of each token's max -> LDS (64 rows x 256 B = 1024 chunks of 16 B)
#pragma unroll
        for (int i = 0; i < 1024 / MF_THREADS; ++i) {
            const int ch = tid + i * MF_THREADS, t = ch >> 4, c16 = ch & 15;
            *reinterpret_cast<v4i *>(sm + MF_ROWS + t * 256 + c16 * 16) =
                *reinterpret_cast<const v4i *>(p.tab + (sMax[t] + 128) * 256 + c16 * 16);
        }
        __syncthreads();
        // ---- S3: ShiftGELU(+requant) by table, in place: 24576 bytes = 6144 dwords
#pragma unroll
        for (int i = 0; i < 6144 / MF_THREADS; ++i) {
            const int dw = tid + i * MF_THREADS;                          // dword index in H: [kc][token][8 dwords]
            const int t = (dw >> 3) & (MF_BM - 1);
            const unsigned char *L = reinterpret_cast<const unsigned char *>(sm + MF_ROWS + t * 256);
            unsigned *hp = reinterpret_cast<unsigned *>(sm + MF_H) + dw;
            const unsigned w = *hp ^ 0x80808080u;
            *hp = (unsigned)L[w & 0xff] | ((unsigned)L[(w >> 8) & 0xff] << 8) | ((unsigned)L[(w >> 16) & 0xff] << 16) |
                  ((unsigned)L[w >> 24] << 24);
        }
        __syncthreads();
        // ---- S4: fc2 on waves 0..5: sub-tile (nt2, mt), K = 384
        if (wave < 6) {
            const int nt = wave >> 1, mt = wave & 1;
            v16i_sw acc;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const v4i b4 = *reinterpret_cast<const v4i *>(sB2 + nt * 32 + g * 8 + half * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[g * 4 + e] = b4[e];
            }
#pragma unroll
            for (int kc = 0; kc < 12; ++kc) {
                const v4i wf = *reinterpret_cast<const v4i *>(sm + MF_W2 + kc * (MF_C * 32) + (nt * 32 + l31) * 32 + half * 16);
                const v4i gf = *reinterpret_cast<const v4i *>(sm + MF_H + kc * (MF_BM * 32) + (mt * 32 + l31) * 32 + half * 16);
                acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf, gf, acc, 0, 0, 0);
            }
            // C^T[n][token]: lane = token, quad g -> channels nt*32 + 8g + 4*half + (0..3); 16-bit requant, pack
            unsigned W[4][2];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n0 = nt * 32 + g * 8 + half * 4;
                int o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    o[e] = min(max(rint_sat_i32((double)acc[g * 4 + e] * sC2[n0 + e]), -32768), 32767);
                W[g][0] = __builtin_amdgcn_perm((unsigned)o[1], (unsigned)o[0], 0x05040100u);
                W[g][1] = __builtin_amdgcn_perm((unsigned)o[3], (unsigned)o[2], 0x05040100u);
            }
            const long long tok = tile * MF_BM + mt * 32 + l31;
#pragma unroll
            for (int q = 0; q < 2; ++q) {                          // after the exchange: 8 channels per (q, half)
                auto x0 = __builtin_amdgcn_permlane32_swap(W[q][0], W[q + 2][0], false, false);
                auto x1 = __builtin_amdgcn_permlane32_swap(W[q][1], W[q + 2][1], false, false);
                v4i v = {(int)x0[0], (int)x1[0], (int)x0[1], (int)x1[1]};
                const int ch0 = nt * 32 + half * 16 + q * 8;
                if (tok < p.M) {
                    const long long off = tok * MF_C + ch0;
                    const v4i rs = rsv[q];
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const int t0 = (int)(short)(v[w] & 0xffff), t1 = v[w] >> 16;
                        const int r0 = (int)(short)(rs[w] & 0xffff), r1 = rs[w] >> 16;
                        int o0 = rint_sat_i32((double)r0 * cr) + rint_sat_i32((double)t0 * cm);
                        int o1 = rint_sat_i32((double)r1 * cr) + rint_sat_i32((double)t1 * cm);
                        o0 = min(max(o0, -32768), 32767);
                        o1 = min(max(o1, -32768), 32767);
                        v[w] = (o0 & 0xffff) | (o1 << 16);
                    }
                    *reinterpret_cast<v4i *>(p.out + off) = v;
                }
            }
        }
    }
}
