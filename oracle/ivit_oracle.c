/*
 * ivit_oracle.c — CPU restatement of the reference I-ViT integer-only inference
 * operators.  TEST INFRASTRUCTURE ONLY: used by tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg as the checker; never by the product path.
 *
 * Parity is PINNED: every function here is checked against outputs of the
 * reference itself (imported in the build container by tools/make_golden.py,
 * fixtures under tests/golden/) — see tests/test_oracle_golden.py.
 *
 * Each function cites the reference file:line it restates (paths relative to
 * the reference repo root).  The reference is "fake-quant": fp32 tensors that
 * hold integer*scale.  The contractions and the dyadic requantisation are exact
 * integer arithmetic; Shiftmax, ShiftGELU and I-LayerNorm are fp32 sequences
 * whose individual IEEE-754 roundings matter, so they are restated op by op in
 * binary32 (build with -ffp-contract=off, no fast-math).
 *
 * Build: see oracle/Makefile  (gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { double m; double r; } ivit_dyadic; /* out = rne((z*m)*r), r = 2^-e */

/* ------------------------------------------------------------------------- */
/* a4: input quantisation.  models/quantization_utils/quant_utils.py:12-48,77-96
 * q = clamp(round(fl(fl(1/s) * x)), -2^(b-1), 2^(b-1)-1)                      */
void ivit_ref_quantize_f32(const float *x, float scale, int bits, int32_t *q, int64_t n) {
    const float inv = 1.0f / scale;
    const float lo = -(float)(1 << (bits - 1)), hi = (float)((1 << (bits - 1)) - 1);
    for (int64_t i = 0; i < n; ++i) {
        float v = rintf(inv * x[i]);
        v = v < lo ? lo : (v > hi ? hi : v);
        q[i] = (int32_t)v;
    }
}

/* ------------------------------------------------------------------------- */
/* a1: QuantLinear.forward  models/quantization_utils/quant_modules.py:67-97
 * acc[i,j] = sum_k x[i,k]*w[j,k] + b[j]   (exact int32)                        */
void ivit_ref_linear_i8(const int8_t *x, const int8_t *w, const int32_t *bias, int32_t *acc,
                        int64_t M, int64_t N, int64_t K) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < M; ++i) {
        const int8_t *xi = x + i * K;
        for (int64_t j = 0; j < N; ++j) {
            const int8_t *wj = w + j * K;
            int32_t s = 0;
            for (int64_t k = 0; k < K; ++k) s += (int32_t)xi[k] * (int32_t)wj[k];
            acc[i * N + j] = s + (bias ? bias[j] : 0);
        }
    }
}

/* a2: QuantMatMul.forward  quant_modules.py:223-228.  Batched A[b]·B[b]ᵀ with
 * A int8 [nb,M,K] (row stride K), B int8 [nb,N,K]  -> int32 [nb,M,N]
 * (q·kᵀ: A=q, B=k.)                                                           */
void ivit_ref_bmm_nt_i8(const int8_t *A, const int8_t *B, int32_t *C, int64_t nb, int64_t M,
                        int64_t N, int64_t K) {
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < nb; ++b)
        for (int64_t i = 0; i < M; ++i)
            for (int64_t j = 0; j < N; ++j) {
                const int8_t *a = A + (b * M + i) * K, *bb = B + (b * N + j) * K;
                int32_t s = 0;
                for (int64_t k = 0; k < K; ++k) s += (int32_t)a[k] * (int32_t)bb[k];
                C[(b * M + i) * N + j] = s;
            }
}

/* a2 (attn·v): P uint16-valued [nb,M,T] (ldp), V int8 [nb,T,D] -> int32 [nb,M,D] */
void ivit_ref_bmm_av(const uint16_t *P, int64_t ldp, const int8_t *V, int32_t *C, int64_t nb,
                     int64_t M, int64_t T, int64_t D) {
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < nb; ++b)
        for (int64_t i = 0; i < M; ++i) {
            int32_t *c = C + (b * M + i) * D;
            for (int64_t d = 0; d < D; ++d) c[d] = 0;
            for (int64_t t = 0; t < T; ++t) {
                int32_t p = P[(b * M + i) * ldp + t];
                const int8_t *v = V + (b * T + t) * D;
                for (int64_t d = 0; d < D; ++d) c[d] += p * (int32_t)v[d];
            }
        }
}

/* ------------------------------------------------------------------------- */
/* a3: dyadic requantisation.  quant_utils.py:213-253 (fixedpoint_mul.forward):
 *   output = round( (double(z) * double(m)) / 2^e )  [+ same for the identity]
 *   clamp to [-2^(b-1), 2^(b-1)-1]
 * z is given as float (integer-valued; may exceed 2^31 after I-LayerNorm) or int32.
 * dy has `nch` entries (1 = per-tensor, C = per channel on the last dim).       */
static inline double rq(double z, ivit_dyadic d) { return rint((z * d.m) * d.r); }

void ivit_ref_requant_f32(const float *z, const ivit_dyadic *dy, int64_t nch, const float *z_id,
                          const ivit_dyadic *dy_id, int bits, int32_t *out, int64_t rows,
                          int64_t C) {
    const double lo = -(double)(1ll << (bits - 1)), hi = (double)((1ll << (bits - 1)) - 1);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < rows; ++i)
        for (int64_t c = 0; c < C; ++c) {
            double o = rq((double)z[i * C + c], dy[nch == 1 ? 0 : c]);
            if (z_id) o = rq((double)z_id[i * C + c], dy_id[0]) + o;
            float f = (float)o; /* output.type(torch.float) then clamp */
            f = f < (float)lo ? (float)lo : (f > (float)hi ? (float)hi : f);
            out[i * C + c] = (int32_t)f;
        }
}

void ivit_ref_requant_i32(const int32_t *z, const ivit_dyadic *dy, int64_t nch,
                          const int32_t *z_id, const ivit_dyadic *dy_id, int bits, int32_t *out,
                          int64_t rows, int64_t C) {
    const float lo = -(float)(1ll << (bits - 1)), hi = (float)((1ll << (bits - 1)) - 1);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < rows; ++i)
        for (int64_t c = 0; c < C; ++c) {
            double o = rq((double)z[i * C + c], dy[nch == 1 ? 0 : c]);
            if (z_id) o = rq((double)z_id[i * C + c], dy_id[0]) + o;
            float f = (float)o;
            f = f < lo ? lo : (f > hi ? hi : f);
            out[i * C + c] = (int32_t)f;
        }
}

/* ------------------------------------------------------------------------- */
/* torch-CPU `sum` over a contiguous last dim, fp32 (ATen SumKernel: 8-lane vector,
 * ILP 4, 4-level cascade).  Order matters for I-LayerNorm; pinned
 * against torch itself in tests/test_oracle_golden.py and tools/make_golden.py.  */
static int ceil_log2_i64(int64_t x) {
    if (x <= 2) return 1;
    int l = 0;
    int64_t v = x - 1;
    while (v > 0) { v >>= 1; ++l; }
    return l;
}

float ivit_ref_torch_sum_f32(const float *x, int64_t n) {
    enum { V = 8, ILP = 4, LEVELS = 4 };
    float acc[LEVELS][ILP][V];
    memset(acc, 0, sizeof(acc));
    const int64_t nvec = n / V;
    const int64_t size = nvec / ILP;
    int level_power = ceil_log2_i64(size) / LEVELS;
    if (level_power < 4) level_power = 4;
    const int64_t level_step = (int64_t)1 << level_power;
    const int64_t level_mask = level_step - 1;
    int64_t i = 0;
    for (; i + level_step <= size;) {
        for (int64_t j = 0; j < level_step; ++j, ++i)
            for (int k = 0; k < ILP; ++k)
                for (int l = 0; l < V; ++l) acc[0][k][l] += x[(i * ILP + k) * V + l];
        for (int j = 1; j < LEVELS; ++j) {
            for (int k = 0; k < ILP; ++k)
                for (int l = 0; l < V; ++l) {
                    acc[j][k][l] += acc[j - 1][k][l];
                    acc[j - 1][k][l] = 0.0f;
                }
            const int64_t mask = level_mask << (j * level_power);
            if ((i & mask) != 0) break;
        }
    }
    for (; i < size; ++i)
        for (int k = 0; k < ILP; ++k)
            for (int l = 0; l < V; ++l) acc[0][k][l] += x[(i * ILP + k) * V + l];
    for (int j = 1; j < LEVELS; ++j)
        for (int k = 0; k < ILP; ++k)
            for (int l = 0; l < V; ++l) acc[0][k][l] += acc[j][k][l];
    for (int64_t v = size * ILP; v < nvec; ++v)
        for (int l = 0; l < V; ++l) acc[0][0][l] += x[v * V + l];
    for (int k = 1; k < ILP; ++k)
        for (int l = 0; l < V; ++l) acc[0][0][l] += acc[0][k][l];
    float fin = 0.0f;
    for (int64_t k = nvec * V; k < n; ++k) fin += x[k];
    if (nvec > 0)
        for (int l = 0; l < V; ++l) fin += acc[0][0][l];
    return fin;
}

/* ------------------------------------------------------------------------- */
/* shift-exp shared by Shiftmax / ShiftGELU.  quant_modules.py:410-423,469-481
 * (int_exp_shift): x <= 0 "integer" (fp32), x0 = floor(-1/s), n = 15 or 23.    */
static inline float iexp_shift(float x, float x0, float nx0, int n) {
    float t = x + floorf(x / 2.0f);
    t = t - floorf(x / 16.0f);
    t = t > nx0 ? t : nx0;               /* torch.max(x_int, n*x0_int) */
    float q = floorf(t / x0);
    float r = t - x0 * q;
    float e = r / 2.0f - x0;
    e = floorf(e * ldexpf(1.0f, n - (int)q)); /* 2 ** (n - q): exact power of two */
    return e > 0.0f ? e : 0.0f;          /* clamp(min=0) */
}

/* a5: IntSoftmax.forward (Shiftmax)  quant_modules.py:483-497.
 * x int8 [rows, n] (row stride ld_in), per-tensor scale s; out_bits 16 (ViT/DeiT)
 * or 8 (Swin). Output integer probabilities (scale 2^-(b-1)); 16-bit results can
 * reach 32768, hence uint16 storage.  `addmask` (optional, [rows?]) is unused for
 * ViT; see Swin.                                                               */
void ivit_ref_shiftmax(const int8_t *x, int64_t rows, int64_t n, int64_t ld_in, float s,
                       int out_bits, uint16_t *out, int64_t ld_out) {
    const float x0 = floorf(-1.0f / s);
    const float nx0 = 15.0f * x0;
    const float div = ldexpf(1.0f, 31 - out_bits + 1);
#pragma omp parallel
    {
        float *xt = (float *)malloc(sizeof(float) * (size_t)n);
        float *e = (float *)malloc(sizeof(float) * (size_t)n);
#pragma omp for schedule(static)
        for (int64_t i = 0; i < rows; ++i) {
            const int8_t *xi = x + i * ld_in;
            float mx = -INFINITY;
            for (int64_t j = 0; j < n; ++j) {
                float X = (float)xi[j] * s;   /* previous QuantAct: Q*s   (quant_modules.py:206) */
                xt[j] = X / s;                /* x / scaling_factor        (:484) */
                mx = xt[j] > mx ? xt[j] : mx;
            }
            for (int64_t j = 0; j < n; ++j) e[j] = iexp_shift(xt[j] - mx, x0, nx0, 15);
            float S = ivit_ref_torch_sum_f32(e, n);
            S = S < 2147483648.0f ? S : 2147483648.0f;  /* clamp_max_(2**31-1) in fp32 */
            float F = floorf((1.0f / S) * 2147483648.0f); /* (2**31-1)/sum = recip*scalar */
            uint16_t *o = out + i * ld_out;
            for (int64_t j = 0; j < n; ++j) o[j] = (uint16_t)floorf((e[j] * F) / div);
        }
        free(xt);
        free(e);
    }
}

/* a6: IntGELU.forward (ShiftGELU)  quant_modules.py:425-445.
 * x int8 [rows, C], per-tensor scale s -> out[i] = Q * sigmoid_int (|.| <= 128*127,
 * int16), output scale s*2^-7.                                                  */
void ivit_ref_shiftgelu(const int8_t *x, int64_t rows, int64_t C, float s, int16_t *out) {
    const float ssig = s * 1.702f;
    const float x0 = floorf(-1.0f / ssig);
    const float nx0 = 23.0f * x0;
#pragma omp parallel
    {
        float *p = (float *)malloc(sizeof(float) * (size_t)C);
#pragma omp for schedule(static)
        for (int64_t i = 0; i < rows; ++i) {
            const int8_t *xi = x + i * C;
            float mx = -INFINITY;
            for (int64_t j = 0; j < C; ++j) {
                float X = (float)xi[j] * s;
                p[j] = X / s;
                mx = p[j] > mx ? p[j] : mx;
            }
            const float emax = iexp_shift(-mx, x0, nx0, 23);
            for (int64_t j = 0; j < C; ++j) {
                float e = iexp_shift(p[j] - mx, x0, nx0, 23);
                float S = e + emax;
                S = S < 2147483648.0f ? S : 2147483648.0f;
                float F = floorf((1.0f / S) * 2147483648.0f);
                float sig = floorf((e * F) / 16777216.0f); /* 2^(31-8+1) */
                float o = p[j] * sig;
                /* the next QuantAct takes round(fl(fl(o*sf)/sf)); see requant */
                out[i * C + j] = (int16_t)rintf(o);
            }
        }
        free(p);
    }
}

/* a7: IntLayerNorm.forward  quant_modules.py:353-386.
 * x int16 [rows, C] with per-tensor scale s; bias_int[c] = floor(fl(fl(b/w)/sf)),
 * sc[c] = fl(sf*w[c]) prepared on the host (sf = fl(sqrt(C))/2^30).
 * Output z[i,c] = round(fl(fl(out*sc)/sc)) — the integer the following QuantAct
 * derives (quant_utils.py:220) — as float (can exceed 2^31).                     */
void ivit_ref_layernorm(const int16_t *x, int64_t rows, int64_t C, float s, const float *bias_int,
                        const float *sc, float *z) {
#pragma omp parallel
    {
        float *xt = (float *)malloc(sizeof(float) * (size_t)C);
        float *y2 = (float *)malloc(sizeof(float) * (size_t)C);
#pragma omp for schedule(static)
        for (int64_t i = 0; i < rows; ++i) {
            const int16_t *xi = x + i * C;
            for (int64_t j = 0; j < C; ++j) {
                float X = (float)xi[j] * s;
                xt[j] = X / s;
            }
            float mean = rintf(ivit_ref_torch_sum_f32(xt, C) / (float)C);
            for (int64_t j = 0; j < C; ++j) {
                xt[j] = xt[j] - mean;
                y2[j] = xt[j] * xt[j];
            }
            float var = ivit_ref_torch_sum_f32(y2, C);
            float k = 65536.0f;
            for (int it = 0; it < 10; ++it) k = floorf((k + floorf(var / k)) / 2.0f);
            float F = floorf((1.0f / k) * 2147483648.0f);
            for (int64_t j = 0; j < C; ++j) {
                float yi = floorf((xt[j] * F) / 2.0f);
                float o = yi + bias_int[j];
                float Xo = o * sc[j];
                z[i * C + j] = rintf(Xo / sc[j]);
            }
        }
        free(xt);
        free(y2);
    }
}

/* torch-CPU `sum` over a NON-contiguous last dim whose neighbouring dim (tokens) is the
 * contiguous one — what the reference's IntLayerNorm sees in Swin stage 0, where the
 * activation keeps the layout of `x.flatten(2).transpose(1, 2)` (layers_quant.py:188,
 * swin_quant.py:251-258).  ATen SumKernel vectorized_outer_sum: tokens are processed in groups of
 * 32 (4 x 8 lanes) with multi_row_sum over the channels = plain sequential accumulation with
 * the 16-step cascade; the last (L mod 32) tokens of an image use row_sum = 4 interleaved
 * accumulators.  parallel chunks are rounded to 32 tokens, so this is thread-count invariant
 * whenever L % 32 == 0 (Swin-T: L = 3136).                                              */
static float cascade_seq_sum(const float *x, int64_t n, int64_t stride) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int level_power = ceil_log2_i64(n) / 4;
    if (level_power < 4) level_power = 4;
    const int64_t step = (int64_t)1 << level_power, lmask = step - 1;
    int64_t i = 0;
    for (; i + step <= n;) {
        for (int64_t j = 0; j < step; ++j, ++i) a0 += x[i * stride];
        a1 += a0; a0 = 0.f;
        if ((i & (lmask << level_power)) != 0) continue;
        a2 += a1; a1 = 0.f;
        if ((i & (lmask << (2 * level_power))) != 0) continue;
        a3 += a2; a2 = 0.f;
    }
    for (; i < n; ++i) a0 += x[i * stride];
    a0 += a1; a0 += a2; a0 += a3;
    return a0;
}
float ivit_ref_torch_sum_strided_f32(const float *x, int64_t n, int ilp4) {
    if (!ilp4) return cascade_seq_sum(x, n, 1);
    const int64_t size_ilp = n / 4;
    float ps[4];
    for (int k = 0; k < 4; ++k) ps[k] = cascade_seq_sum(x + k, size_ilp, 4);
    for (int64_t i = size_ilp * 4; i < n; ++i) ps[0] += x[i];
    ps[0] += ps[1]; ps[0] += ps[2]; ps[0] += ps[3];
    return ps[0];
}

/* I-LayerNorm with the summation order selected by the reference-side memory layout:
 * order 0 = channel-contiguous input (ivit_ref_layernorm); order 1 = token-contiguous input,
 * L tokens per image (see above).                                                       */
void ivit_ref_layernorm_ord(const int16_t *x, int64_t rows, int64_t C, float s, const float *bias_int,
                            const float *sc, float *z, int order, int64_t L) {
#pragma omp parallel
    {
        float *xt = (float *)malloc(sizeof(float) * (size_t)C);
        float *y2 = (float *)malloc(sizeof(float) * (size_t)C);
#pragma omp for schedule(static)
        for (int64_t i = 0; i < rows; ++i) {
            const int16_t *xi = x + i * C;
            const int ilp4 = order == 1 && (i % L) >= (L / 32) * 32;
            for (int64_t j = 0; j < C; ++j) {
                float X = (float)xi[j] * s;
                xt[j] = X / s;
            }
            float sum = order == 0 ? ivit_ref_torch_sum_f32(xt, C) : ivit_ref_torch_sum_strided_f32(xt, C, ilp4);
            float mean = rintf(sum / (float)C);
            for (int64_t j = 0; j < C; ++j) {
                xt[j] = xt[j] - mean;
                y2[j] = xt[j] * xt[j];
            }
            float var = order == 0 ? ivit_ref_torch_sum_f32(y2, C) : ivit_ref_torch_sum_strided_f32(y2, C, ilp4);
            float k = 65536.0f;
            for (int it = 0; it < 10; ++it) k = floorf((k + floorf(var / k)) / 2.0f);
            float F = floorf((1.0f / k) * 2147483648.0f);
            for (int64_t j = 0; j < C; ++j) {
                float yi = floorf((xt[j] * F) / 2.0f);
                float o = yi + bias_int[j];
                float Xo = o * sc[j];
                z[i * C + j] = rintf(Xo / sc[j]);
            }
        }
        free(xt);
        free(y2);
    }
}

/* a8: patch gather for QuantConv2d with kernel=stride=P (layers_quant.py:184-196,
 * quant_modules.py:297-330).  NCHW int8 image -> rows [B*gh*gw, Cin*P*P] in the
 * conv-weight element order (c, ph, pw), so conv == ivit_ref_linear_i8.          */
void ivit_ref_im2col_patch(const int8_t *img, int64_t B, int64_t Cin, int64_t H, int64_t W,
                           int64_t P, int8_t *out) {
    const int64_t gh = H / P, gw = W / P, K = Cin * P * P;
    for (int64_t b = 0; b < B; ++b)
        for (int64_t gy = 0; gy < gh; ++gy)
            for (int64_t gx = 0; gx < gw; ++gx) {
                int8_t *o = out + ((b * gh + gy) * gw + gx) * K;
                for (int64_t c = 0; c < Cin; ++c)
                    for (int64_t py = 0; py < P; ++py)
                        memcpy(o + (c * P + py) * P,
                               img + ((b * Cin + c) * H + gy * P + py) * W + gx * P, (size_t)P);
            }
}

/* ------------------------------------------------------------------------- */
/* Swin: IntSoftmax fed by `attn + mask` (models/swin_quant.py:151-156): the float mask
 * (0 / -100.0) is added to the fp32 logits fl(Q*s) BEFORE IntSoftmax divides by s, so masked
 * inputs are non-integers.  Row r of the flattened [B_, H, n] rows uses mask[(r/(H*n)) % nW][r % n][:].
 * mask == NULL reproduces ivit_ref_shiftmax.                                         */
void ivit_ref_shiftmax_masked(const int8_t *x, int64_t rows, int64_t n, int64_t ld_in, float s,
                              int out_bits, const float *mask, int64_t nW, int64_t H, uint16_t *out,
                              int64_t ld_out) {
    const float x0 = floorf(-1.0f / s);
    const float nx0 = 15.0f * x0;
    const float div = ldexpf(1.0f, 31 - out_bits + 1);
#pragma omp parallel
    {
        float *xt = (float *)malloc(sizeof(float) * (size_t)n);
        float *e = (float *)malloc(sizeof(float) * (size_t)n);
#pragma omp for schedule(static)
        for (int64_t i = 0; i < rows; ++i) {
            const int8_t *xi = x + i * ld_in;
            const float *mr = mask ? mask + (((i / (H * n)) % nW) * n + (i % n)) * n : NULL;
            float mx = -INFINITY;
            for (int64_t j = 0; j < n; ++j) {
                float X = (float)xi[j] * s;
                if (mr) X = X + mr[j];
                xt[j] = X / s;
                mx = xt[j] > mx ? xt[j] : mx;
            }
            for (int64_t j = 0; j < n; ++j) e[j] = iexp_shift(xt[j] - mx, x0, nx0, 15);
            float S = ivit_ref_torch_sum_f32(e, n);
            S = S < 2147483648.0f ? S : 2147483648.0f;
            float F = floorf((1.0f / S) * 2147483648.0f);
            uint16_t *o = out + i * ld_out;
            for (int64_t j = 0; j < n; ++j) o[j] = (uint16_t)floorf((e[j] * F) / div);
        }
        free(xt);
        free(e);
    }
}

/* Swin head: AdaptiveAvgPool1d(1) over the L tokens of fl(Q*s) (swin_quant.py:554), fp32
 * sequential sum then /L, followed by the next QuantAct's z = round(fl(mean/s))
 * (quant_utils.py:220).  x int8 [B, L, C] -> z int32 [B, C].                          */
void ivit_ref_avgpool_z(const int8_t *x, int64_t B, int64_t L, int64_t C, float s, int32_t *z) {
    for (int64_t b = 0; b < B; ++b)
        for (int64_t c = 0; c < C; ++c) {
            float sum = 0.0f;
            for (int64_t l = 0; l < L; ++l) sum += (float)x[(b * L + l) * C + c] * s;
            float mean = sum / (float)L;
            z[b * C + c] = (int32_t)rintf(mean / s);
        }
}
