/*
 * ivit_twin.c — CPU twin of the C-ABI (SURVEY.md §8b "a CPU twin of every entry point (same signatures, host
 * pointers) used as the portable oracle").  TEST INFRASTRUCTURE ONLY, like the rest of oracle/: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product path never does.
 *
 * Every function ivit_cpu_X has the parameter list of ivit_X in include/ivit.h (tests/test_twin.py compares the two
 * headers textually) and the same result on host buffers.  The handle argument is ignored (pass NULL).  The bodies are
 * compositions of the restated operators of ivit_oracle.c (each of which cites the reference lines it follows), in the
 * order the reference's modules chain them; the reference citation of an entry point is the one in include/ivit.h.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "ivit_twin.h"

/* ---- restated operators (ivit_oracle.c) */
void ivit_ref_quantize_f32(const float *x, float scale, int bits, int32_t *q, int64_t n);
void ivit_ref_linear_i8(const int8_t *x, const int8_t *w, const int32_t *bias, int32_t *acc, int64_t M, int64_t N,
                        int64_t K);
void ivit_ref_bmm_av(const uint16_t *P, int64_t ldp, const int8_t *V, int32_t *C, int64_t nb, int64_t M, int64_t T,
                     int64_t D);
void ivit_ref_requant_f32(const float *z, const ivit_dyadic *dy, int64_t nch, const float *z_id,
                          const ivit_dyadic *dy_id, int bits, int32_t *out, int64_t rows, int64_t C);
void ivit_ref_requant_i32(const int32_t *z, const ivit_dyadic *dy, int64_t nch, const int32_t *z_id,
                          const ivit_dyadic *dy_id, int bits, int32_t *out, int64_t rows, int64_t C);
void ivit_ref_shiftmax(const int8_t *x, int64_t rows, int64_t n, int64_t ld_in, float s, int out_bits, uint16_t *out,
                       int64_t ld_out);
void ivit_ref_shiftgelu(const int8_t *x, int64_t rows, int64_t C, float s, int16_t *out);
void ivit_ref_layernorm(const int16_t *x, int64_t rows, int64_t C, float s, const float *bias_int, const float *sc,
                        float *z);
void ivit_ref_im2col_patch(const int8_t *img, int64_t B, int64_t Cin, int64_t H, int64_t W, int64_t P, int8_t *out);

#define TW_OK 0
#define TW_INVALID 1
#define TW_REQ(c) do { if (!(c)) return TW_INVALID; } while (0)

static void *xmalloc(size_t n) { return malloc(n ? n : 1); }

/* int32 -> narrower integer store of a requantised tensor (the values are already clamped to `bits`) */
static void store_bits(const int32_t *v, int bits, void *out, int64_t n) {
    if (bits == 8) for (int64_t i = 0; i < n; ++i) ((int8_t *)out)[i] = (int8_t)v[i];
    else if (bits == 16) for (int64_t i = 0; i < n; ++i) ((int16_t *)out)[i] = (int16_t)v[i];
    else memcpy(out, v, (size_t)n * 4);
}

int ivit_cpu_quantize_input_f32(ivit_handle h, const float *x, float scale, int8_t *q, int64_t n) {
    (void)h;
    TW_REQ(x && q && n > 0 && scale > 0.f);
    int32_t *t = (int32_t *)xmalloc((size_t)n * 4);
    ivit_ref_quantize_f32(x, scale, 8, t, n);
    store_bits(t, 8, q, n);
    free(t);
    return TW_OK;
}

int ivit_cpu_linear_i8(ivit_handle h, const int8_t *x, const int8_t *w, const int32_t *bias, int32_t *acc, int M,
                       int N, int K) {
    (void)h;
    TW_REQ(x && w && acc && M > 0 && N > 0 && K > 0);
    ivit_ref_linear_i8(x, w, bias, acc, M, N, K);
    return TW_OK;
}

int ivit_cpu_linear_i8_requant(ivit_handle h, const int8_t *x, const int8_t *w, const int32_t *bias,
                               const ivit_dyadic *dy_ch, int bits, void *out, int M, int N, int K) {
    (void)h;
    TW_REQ(x && w && dy_ch && out && (bits == 8 || bits == 16) && M > 0 && N > 0 && K > 0);
    int32_t *acc = (int32_t *)xmalloc((size_t)M * N * 4);
    ivit_ref_linear_i8(x, w, bias, acc, M, N, K);
    ivit_ref_requant_i32(acc, dy_ch, N, NULL, NULL, bits, acc, M, N);
    store_bits(acc, bits, out, (int64_t)M * N);
    free(acc);
    return TW_OK;
}

/* t = clamp16(rq(acc, dy_ch[j])); out = clamp16(rq(t, dy_main) + rq(residual, dy_res)) */
static void residual_tail(int32_t *t, ivit_dyadic dy_main, ivit_dyadic dy_res, const int16_t *residual, int16_t *out,
                          int64_t M, int64_t N) {
    int32_t *r32 = (int32_t *)xmalloc((size_t)M * N * 4);
    for (int64_t i = 0; i < M * N; ++i) r32[i] = residual[i];
    ivit_ref_requant_i32(t, &dy_main, 1, r32, &dy_res, 16, t, M, N);
    store_bits(t, 16, out, M * N);
    free(r32);
}

int ivit_cpu_linear_i8_requant_residual(ivit_handle h, const int8_t *x, const int8_t *w, const int32_t *bias,
                                        const ivit_dyadic *dy_ch, ivit_dyadic dy_main, ivit_dyadic dy_res,
                                        const int16_t *residual, int16_t *out, int M, int N, int K) {
    (void)h;
    TW_REQ(x && w && dy_ch && residual && out && M > 0 && N > 0 && K > 0);
    int32_t *acc = (int32_t *)xmalloc((size_t)M * N * 4);
    ivit_ref_linear_i8(x, w, bias, acc, M, N, K);
    ivit_ref_requant_i32(acc, dy_ch, N, NULL, NULL, 16, acc, M, N);
    residual_tail(acc, dy_main, dy_res, residual, out, M, N);
    free(acc);
    return TW_OK;
}

int ivit_cpu_linear_i8_qkv(ivit_handle h, const int8_t *x, const int8_t *w, const int32_t *bias,
                           const ivit_dyadic *dy_ch, int8_t *q, int8_t *k, int8_t *vt, int B, int T, int H, int dh,
                           int ldv) {
    (void)h;
    TW_REQ(x && w && dy_ch && q && k && vt && B > 0 && T > 0 && H > 0 && dh > 0 && ldv >= T);
    const int64_t D = (int64_t)H * dh, M = (int64_t)B * T, N = 3 * D;
    int32_t *acc = (int32_t *)xmalloc((size_t)M * N * 4);
    ivit_ref_linear_i8(x, w, bias, acc, M, N, D);
    ivit_ref_requant_i32(acc, dy_ch, N, NULL, NULL, 8, acc, M, N);
    /* vit_quant.py:63-69: [B, T, 3, H, dh] -> q, k [B, H, T, dh]; v transposed [B, H, dh, ldv] (pad columns untouched) */
    for (int64_t b = 0; b < B; ++b)
        for (int64_t t = 0; t < T; ++t)
            for (int64_t hh = 0; hh < H; ++hh)
                for (int64_t d = 0; d < dh; ++d) {
                    const int32_t *row = acc + (b * T + t) * N;
                    const int64_t bh = b * H + hh;
                    q[(bh * T + t) * dh + d] = (int8_t)row[hh * dh + d];
                    k[(bh * T + t) * dh + d] = (int8_t)row[D + hh * dh + d];
                    vt[(bh * dh + d) * ldv + t] = (int8_t)row[2 * D + hh * dh + d];
                }
    free(acc);
    return TW_OK;
}

/* ---- plans: the twin keeps the pointers; the planned entry points are the unplanned ones */
struct ivit_cpu_plan_s { const int8_t *w; const int32_t *bias; const ivit_dyadic *dy; int N, K; };

int ivit_cpu_linear_plan_create(ivit_handle h, const int8_t *w, const int32_t *bias, const ivit_dyadic *dy_ch, int N,
                                int K, ivit_linear_plan *out) {
    (void)h;
    TW_REQ(w && dy_ch && out && N > 0 && K > 0);
    struct ivit_cpu_plan_s *p = (struct ivit_cpu_plan_s *)xmalloc(sizeof(*p));
    p->w = w; p->bias = bias; p->dy = dy_ch; p->N = N; p->K = K;
    *out = (ivit_linear_plan)p;
    return TW_OK;
}
int ivit_cpu_linear_plan_destroy(ivit_linear_plan p) { free(p); return TW_OK; }
int ivit_cpu_linear_plan_query(ivit_linear_plan p, int *pipelined_ok, int *single_fma_ok) {
    TW_REQ(p);
    if (pipelined_ok) *pipelined_ok = 1;
    if (single_fma_ok) *single_fma_ok = 1;
    return TW_OK;
}
int ivit_cpu_linear_i8_requant_planned(ivit_handle h, ivit_linear_plan p, const int8_t *x, int bits, void *out, int M) {
    const struct ivit_cpu_plan_s *pl = (const struct ivit_cpu_plan_s *)p;
    TW_REQ(pl);
    return ivit_cpu_linear_i8_requant(h, x, pl->w, pl->bias, pl->dy, bits, out, M, pl->N, pl->K);
}
int ivit_cpu_linear_i8_requant_residual_planned(ivit_handle h, ivit_linear_plan p, const int8_t *x, ivit_dyadic dy_main,
                                                ivit_dyadic dy_res, const int16_t *residual, int16_t *out, int M) {
    const struct ivit_cpu_plan_s *pl = (const struct ivit_cpu_plan_s *)p;
    TW_REQ(pl);
    return ivit_cpu_linear_i8_requant_residual(h, x, pl->w, pl->bias, pl->dy, dy_main, dy_res, residual, out, M, pl->N,
                                               pl->K);
}
int ivit_cpu_linear_i8_qkv_planned(ivit_handle h, ivit_linear_plan p, const int8_t *x, int8_t *q, int8_t *k, int8_t *vt,
                                   int B, int T, int H, int dh, int ldv) {
    const struct ivit_cpu_plan_s *pl = (const struct ivit_cpu_plan_s *)p;
    TW_REQ(pl && pl->N == 3 * H * dh && pl->K == H * dh);
    return ivit_cpu_linear_i8_qkv(h, x, pl->w, pl->bias, pl->dy, q, k, vt, B, T, H, dh, ldv);
}

/* ---- a2 */
int ivit_cpu_bmm_nt_i8(ivit_handle h, const int8_t *A, const int8_t *B, int32_t *C, int nb, int M, int N, int K, int lda,
                       int ldb, int ldc, int64_t strideA, int64_t strideB, int64_t strideC) {
    (void)h;
    TW_REQ(A && B && C && nb > 0 && M > 0 && N > 0 && K > 0);
    for (int64_t b = 0; b < nb; ++b)
        for (int64_t i = 0; i < M; ++i)
            for (int64_t j = 0; j < N; ++j) {
                const int8_t *a = A + b * strideA + i * lda, *bb = B + b * strideB + j * ldb;
                int32_t s = 0;
                for (int64_t kk = 0; kk < K; ++kk) s += (int32_t)a[kk] * (int32_t)bb[kk];
                C[b * strideC + i * ldc + j] = s;
            }
    return TW_OK;
}

int ivit_cpu_bmm_nt_u16i8(ivit_handle h, const uint16_t *A, const int8_t *B, int32_t *C, int nb, int M, int N, int K,
                          int lda, int ldb, int ldc, int64_t strideA, int64_t strideB, int64_t strideC) {
    (void)h;
    TW_REQ(A && B && C && nb > 0 && M > 0 && N > 0 && K > 0);
    for (int64_t b = 0; b < nb; ++b)
        for (int64_t i = 0; i < M; ++i)
            for (int64_t j = 0; j < N; ++j) {
                const uint16_t *a = A + b * strideA + i * lda;
                const int8_t *bb = B + b * strideB + j * ldb;
                int32_t s = 0;
                for (int64_t kk = 0; kk < K; ++kk) s += (int32_t)a[kk] * (int32_t)bb[kk];
                C[b * strideC + i * ldc + j] = s;
            }
    return TW_OK;
}

int ivit_cpu_attn_qk_requant(ivit_handle h, const int8_t *q, const int8_t *k, ivit_dyadic dy, int8_t *scores8, int BH,
                             int T, int dh, int lds) {
    TW_REQ(q && k && scores8 && BH > 0 && T > 0 && dh > 0 && lds >= T);
    int32_t *acc = (int32_t *)xmalloc((size_t)BH * T * T * 4);
    ivit_cpu_bmm_nt_i8(h, q, k, acc, BH, T, T, dh, dh, dh, T, (int64_t)T * dh, (int64_t)T * dh, (int64_t)T * T);
    ivit_ref_requant_i32(acc, &dy, 1, NULL, NULL, 8, acc, (int64_t)BH * T, T);
    for (int64_t r = 0; r < (int64_t)BH * T; ++r)
        for (int64_t j = 0; j < T; ++j) scores8[r * lds + j] = (int8_t)acc[r * T + j];
    free(acc);
    return TW_OK;
}

int ivit_cpu_attn_pv_requant(ivit_handle h, const uint16_t *p, const int8_t *vt, ivit_dyadic dy, int8_t *ctx8, int B,
                             int H, int T, int dh, int ldp, int ldv) {
    TW_REQ(p && vt && ctx8 && B > 0 && H > 0 && T > 0 && dh > 0 && ldp >= T && ldv >= T);
    const int64_t BH = (int64_t)B * H;
    int32_t *acc = (int32_t *)xmalloc((size_t)BH * T * dh * 4);
    ivit_cpu_bmm_nt_u16i8(h, p, vt, acc, (int)BH, T, dh, T, ldp, ldv, dh, (int64_t)T * ldp, (int64_t)dh * ldv,
                          (int64_t)T * dh);
    ivit_ref_requant_i32(acc, &dy, 1, NULL, NULL, 8, acc, BH * T, dh);
    /* heads merged: x.transpose(1, 2).reshape(B, N, C)  (vit_quant.py:81) */
    for (int64_t b = 0; b < B; ++b)
        for (int64_t hh = 0; hh < H; ++hh)
            for (int64_t t = 0; t < T; ++t)
                for (int64_t d = 0; d < dh; ++d)
                    ctx8[(b * T + t) * H * dh + hh * dh + d] = (int8_t)acc[((b * H + hh) * T + t) * dh + d];
    free(acc);
    return TW_OK;
}

int ivit_cpu_attention_fused(ivit_handle h, const int8_t *q, const int8_t *k, const int8_t *vt, ivit_dyadic dy_qk,
                             float s_softmax, ivit_dyadic dy_pv, int8_t *ctx8, int B, int H, int T, int dh, int ldv) {
    TW_REQ(q && k && vt && ctx8 && B > 0 && H > 0 && T > 0 && dh > 0 && ldv >= T && s_softmax > 0.f);
    const int64_t BH = (int64_t)B * H;
    int8_t *s8 = (int8_t *)xmalloc((size_t)BH * T * T);
    uint16_t *p16 = (uint16_t *)xmalloc((size_t)BH * T * T * 2);
    int rc = ivit_cpu_attn_qk_requant(h, q, k, dy_qk, s8, (int)BH, T, dh, T);
    if (rc == TW_OK) {
        ivit_ref_shiftmax(s8, BH * T, T, T, s_softmax, 16, p16, T);
        rc = ivit_cpu_attn_pv_requant(h, p16, vt, dy_pv, ctx8, B, H, T, dh, T, ldv);
    }
    free(s8);
    free(p16);
    return rc;
}

/* the table form computes the same integers (the tables are checked exhaustively against the arithmetic when built) */
int ivit_cpu_attention_fused_lut(ivit_handle h, const int8_t *q, const int8_t *k, const int8_t *vt, ivit_dyadic dy_qk,
                                 float s_softmax, const uint16_t *exp_aq, const float *exp_t, const uint8_t *exp_cls,
                                 int nclass, int t_count, int dmin, ivit_dyadic dy_pv, int8_t *ctx8, int B, int H, int T,
                                 int dh, int ldv) {
    (void)exp_aq; (void)exp_t; (void)exp_cls; (void)nclass; (void)t_count; (void)dmin;
    return ivit_cpu_attention_fused(h, q, k, vt, dy_qk, s_softmax, dy_pv, ctx8, B, H, T, dh, ldv);
}

/* ---- a3 */
int ivit_cpu_requant_i32(ivit_handle h, const int32_t *z, const ivit_dyadic *dy, int nch, const int32_t *z_id,
                         const ivit_dyadic *dy_id, int bits, void *out, int64_t rows, int C) {
    (void)h;
    TW_REQ(z && dy && out && (nch == 1 || nch == C) && (bits == 8 || bits == 16 || bits == 32) && rows > 0 && C > 0 &&
           (!z_id || dy_id));
    int32_t *t = (int32_t *)xmalloc((size_t)rows * C * 4);
    ivit_ref_requant_i32(z, dy, nch, z_id, dy_id, bits, t, rows, C);
    store_bits(t, bits, out, rows * C);
    free(t);
    return TW_OK;
}
int ivit_cpu_requant_i16(ivit_handle h, const int16_t *z, const ivit_dyadic *dy, int nch, const int32_t *z_id,
                         const ivit_dyadic *dy_id, int bits, void *out, int64_t rows, int C) {
    TW_REQ(z && rows > 0 && C > 0);
    int32_t *z32 = (int32_t *)xmalloc((size_t)rows * C * 4);
    for (int64_t i = 0; i < rows * C; ++i) z32[i] = z[i];
    const int rc = ivit_cpu_requant_i32(h, z32, dy, nch, z_id, dy_id, bits, out, rows, C);
    free(z32);
    return rc;
}
int ivit_cpu_requant_f32(ivit_handle h, const float *z, const ivit_dyadic *dy, int nch, const int32_t *z_id,
                         const ivit_dyadic *dy_id, int bits, void *out, int64_t rows, int C) {
    (void)h;
    TW_REQ(z && dy && out && (nch == 1 || nch == C) && (bits == 8 || bits == 16 || bits == 32) && rows > 0 && C > 0 &&
           (!z_id || dy_id));
    int32_t *t = (int32_t *)xmalloc((size_t)rows * C * 4);
    float *zid = NULL;
    if (z_id) {
        zid = (float *)xmalloc((size_t)rows * C * 4);
        for (int64_t i = 0; i < rows * C; ++i) zid[i] = (float)z_id[i];     /* exact: |identity| < 2^24 on this path */
    }
    ivit_ref_requant_f32(z, dy, nch, zid, dy_id, bits, t, rows, C);
    store_bits(t, bits, out, rows * C);
    free(zid);
    free(t);
    return TW_OK;
}

/* ---- a5, a6 */
int ivit_cpu_shiftmax(ivit_handle h, const int8_t *x, int64_t rows, int n, int ld_in, float scale, int out_bits,
                      uint16_t *out, int ld_out) {
    (void)h;
    TW_REQ(x && out && rows > 0 && n > 0 && ld_in >= n && ld_out >= n && scale > 0.f && (out_bits == 8 || out_bits == 16));
    ivit_ref_shiftmax(x, rows, n, ld_in, scale, out_bits, out, ld_out);
    return TW_OK;
}
int ivit_cpu_shiftgelu(ivit_handle h, const int8_t *x, int64_t rows, int C, float scale, int16_t *out16) {
    (void)h;
    TW_REQ(x && out16 && rows > 0 && C > 0 && scale > 0.f);
    ivit_ref_shiftgelu(x, rows, C, scale, out16);
    return TW_OK;
}
int ivit_cpu_shiftgelu_requant(ivit_handle h, const int8_t *x, int64_t rows, int C, float scale, ivit_dyadic dy,
                               int8_t *out8) {
    TW_REQ(x && out8 && rows > 0 && C > 0 && scale > 0.f);
    int16_t *g = (int16_t *)xmalloc((size_t)rows * C * 2);
    ivit_ref_shiftgelu(x, rows, C, scale, g);
    const int rc = ivit_cpu_requant_i16(h, g, &dy, 1, NULL, NULL, 8, out8, rows, C);
    free(g);
    return rc;
}
/* table[(qmax + 128) * 256 + (Q + 128)] for Q <= qmax (a row's maximum bounds its elements; the other half of the
 * table is never indexed and left 0): the two-element row {Q, qmax} has the same maximum, hence the same result */
int ivit_cpu_shiftgelu_build_table(ivit_handle h, float scale, ivit_dyadic dy, int8_t *table) {
    TW_REQ(table && scale > 0.f);
    memset(table, 0, 65536);
    for (int qmax = -128; qmax < 128; ++qmax)
        for (int Q = -128; Q <= qmax; ++Q) {
            const int8_t row[2] = {(int8_t)Q, (int8_t)qmax};
            int8_t o[2];
            const int rc = ivit_cpu_shiftgelu_requant(h, row, 1, 2, scale, dy, o);
            if (rc) return rc;
            table[(qmax + 128) * 256 + (Q + 128)] = o[0];
        }
    return TW_OK;
}
int ivit_cpu_shiftgelu_requant_lut(ivit_handle h, const int8_t *x, int64_t rows, int C, const int8_t *table,
                                   int8_t *out8) {
    (void)h;
    TW_REQ(x && table && out8 && rows > 0 && C > 0);
    for (int64_t r = 0; r < rows; ++r) {
        int qmax = -128;
        for (int c = 0; c < C; ++c) qmax = x[r * C + c] > qmax ? x[r * C + c] : qmax;
        for (int c = 0; c < C; ++c) out8[r * C + c] = table[(qmax + 128) * 256 + (x[r * C + c] + 128)];
    }
    return TW_OK;
}

/* ---- a7 */
int ivit_cpu_layernorm(ivit_handle h, const int16_t *x, int64_t rows, int C, float scale, const float *bias_int,
                       const float *sc, float *z) {
    (void)h;
    TW_REQ(x && bias_int && sc && z && rows > 0 && C > 0 && scale > 0.f);
    ivit_ref_layernorm(x, rows, C, scale, bias_int, sc, z);
    return TW_OK;
}
int ivit_cpu_layernorm_requant(ivit_handle h, const int16_t *x, int64_t rows, int C, int64_t row_stride, float scale,
                               const float *bias_int, const float *sc, const ivit_dyadic *dy_ch, int8_t *out8) {
    TW_REQ(x && bias_int && sc && dy_ch && out8 && rows > 0 && C > 0 && row_stride >= C && scale > 0.f);
    int16_t *xc = (int16_t *)xmalloc((size_t)rows * C * 2);
    float *z = (float *)xmalloc((size_t)rows * C * 4);
    for (int64_t r = 0; r < rows; ++r) memcpy(xc + r * C, x + r * row_stride, (size_t)C * 2);
    ivit_ref_layernorm(xc, rows, C, scale, bias_int, sc, z);
    const int rc = ivit_cpu_requant_f32(h, z, dy_ch, C, NULL, NULL, 8, out8, rows, C);
    free(z);
    free(xc);
    return rc;
}

/* ---- a8 */
int ivit_cpu_im2col_patch(ivit_handle h, const int8_t *img, int B, int Cin, int H, int W, int P, int8_t *rows) {
    (void)h;
    TW_REQ(img && rows && B > 0 && Cin > 0 && P > 0 && H % P == 0 && W % P == 0);
    ivit_ref_im2col_patch(img, B, Cin, H, W, P, rows);
    return TW_OK;
}
int ivit_cpu_embed_finish(ivit_handle h, const int16_t *patch16, const int32_t *z_cls, const int16_t *pos,
                          ivit_dyadic dy_x, ivit_dyadic dy_pos, int16_t *x16, int B, int T, int D) {
    (void)h;
    TW_REQ(patch16 && z_cls && pos && x16 && B > 0 && T > 1 && D > 0);
    int32_t *z = (int32_t *)xmalloc((size_t)T * D * 4), *pz = (int32_t *)xmalloc((size_t)T * D * 4);
    for (int64_t i = 0; i < (int64_t)T * D; ++i) pz[i] = pos[i];
    for (int64_t b = 0; b < B; ++b) {
        for (int d = 0; d < D; ++d) z[d] = z_cls[d];
        for (int64_t i = 0; i < (int64_t)(T - 1) * D; ++i) z[D + i] = patch16[b * (int64_t)(T - 1) * D + i];
        ivit_ref_requant_i32(z, &dy_x, 1, pz, &dy_pos, 16, z, T, D);
        store_bits(z, 16, x16 + b * (int64_t)T * D, (int64_t)T * D);
    }
    free(z);
    free(pz);
    return TW_OK;
}

/* ---- a9, the fused form of the narrow Swin stage */
int ivit_cpu_mlp_fused(ivit_handle h, const int8_t *x, const int8_t *w1, const int32_t *b1, const ivit_dyadic *dy1,
                       const int8_t *gelu_table, const int8_t *w2, const int32_t *b2, const ivit_dyadic *dy2,
                       ivit_dyadic dy_main, ivit_dyadic dy_res, const int16_t *residual, int16_t *out, int64_t M, int C,
                       int hidden) {
    TW_REQ(x && w1 && dy1 && gelu_table && w2 && dy2 && residual && out && M > 0 && C > 0 && hidden > 0);
    int8_t *h8 = (int8_t *)xmalloc((size_t)M * hidden), *g8 = (int8_t *)xmalloc((size_t)M * hidden);
    int rc = ivit_cpu_linear_i8_requant(h, x, w1, b1, dy1, 8, h8, (int)M, hidden, C);
    if (rc == TW_OK) rc = ivit_cpu_shiftgelu_requant_lut(h, h8, M, hidden, gelu_table, g8);
    if (rc == TW_OK) rc = ivit_cpu_linear_i8_requant_residual(h, g8, w2, b2, dy2, dy_main, dy_res, residual, out, (int)M, C, hidden);
    free(h8);
    free(g8);
    return rc;
}

/* ==== Swin-specific entry points and the planned fused Mlp (round 3: the rest of the operator-level header) ==== */
void ivit_ref_shiftmax_masked(const int8_t *x, int64_t rows, int64_t n, int64_t ld_in, float s, int out_bits,
                              const float *mask, int64_t nW, int64_t H, uint16_t *out, int64_t ld_out);
void ivit_ref_layernorm_ord(const int16_t *x, int64_t rows, int64_t C, float s, const float *bias_int, const float *sc,
                            float *z, int order, int64_t L);
void ivit_ref_bmm_nt_i8(const int8_t *A, const int8_t *B, int32_t *C, int64_t nb, int64_t M, int64_t N, int64_t K);

static inline int32_t tw_rq(double z, ivit_dyadic d) { return (int32_t)rint((z * d.m) * d.r); }
static inline int32_t tw_clamp(int32_t v, int bits) {
    const int32_t hi = (1 << (bits - 1)) - 1, lo = -hi - 1;
    return v < lo ? lo : (v > hi ? hi : v);
}

int ivit_cpu_shiftmax_masked(ivit_handle h, const int8_t *x, int64_t rows, int n, int ld_in, float scale, int out_bits,
                             const float *mask, int nW, int H, uint16_t *out, int ld_out) {
    (void)h;
    TW_REQ(x && out && rows > 0 && n > 0 && ld_in >= n && ld_out >= n && scale > 0.f && (out_bits == 8 || out_bits == 16));
    TW_REQ(mask == NULL || (nW > 0 && H > 0));
    ivit_ref_shiftmax_masked(x, rows, n, ld_in, scale, out_bits, mask, nW > 0 ? nW : 1, H > 0 ? H : 1, out, ld_out);
    return TW_OK;
}

/* QuantAct with an identity that repeats every id_period elements (swin_quant.py:149) */
int ivit_cpu_requant_i32_bcast(ivit_handle h, const int32_t *z, ivit_dyadic dy, const int32_t *z_id, int64_t id_period,
                               ivit_dyadic dy_id, int bits, void *out, int64_t total) {
    (void)h;
    TW_REQ(z && z_id && out && id_period > 0 && total > 0 && (bits == 8 || bits == 16));
    int32_t *t = (int32_t *)xmalloc((size_t)total * 4);
    for (int64_t i = 0; i < total; ++i) t[i] = tw_clamp(tw_rq((double)z[i], dy) + tw_rq((double)z_id[i % id_period], dy_id), bits);
    store_bits(t, bits, out, total);
    free(t);
    return TW_OK;
}

/* AdaptiveAvgPool1d(1) + QuantAct(8): L odd, so rne(sum(Q) / L) is the reference's round(fl(mean / s)) (ivit_oracle.c
 * ivit_ref_avgpool_z restates the fp32 sequence; tests/test_oracle_golden.py pins the two against each other) */
int ivit_cpu_avgpool_requant(ivit_handle h, const int8_t *x, int B, int L, int C, ivit_dyadic dy, int8_t *out8) {
    (void)h;
    TW_REQ(x && out8 && B > 0 && L > 0 && C > 0);
    for (int64_t b = 0; b < B; ++b)
        for (int64_t c = 0; c < C; ++c) {
            long long sum = 0;
            for (int64_t l = 0; l < L; ++l) sum += x[(b * L + l) * C + c];
            const double z = rint((double)sum / (double)L);
            out8[b * C + c] = (int8_t)tw_clamp(tw_rq(z, dy), 8);
        }
    return TW_OK;
}

int ivit_cpu_layernorm_tokenorder(ivit_handle h, const int16_t *x, int64_t rows, int C, float scale, const float *bias_int,
                                  const float *sc, int tokens_per_image, float *z) {
    (void)h;
    TW_REQ(x && bias_int && sc && z && rows > 0 && C > 0 && scale > 0.f && tokens_per_image > 0);
    ivit_ref_layernorm_ord(x, rows, C, scale, bias_int, sc, z, 1, tokens_per_image);
    return TW_OK;
}
int ivit_cpu_layernorm_tokenorder_requant(ivit_handle h, const int16_t *x, int64_t rows, int C, float scale,
                                          const float *bias_int, const float *sc, const ivit_dyadic *dy,
                                          int tokens_per_image, int8_t *out8) {
    TW_REQ(x && bias_int && sc && dy && out8 && rows > 0 && C > 0 && scale > 0.f && tokens_per_image > 0);
    float *z = (float *)xmalloc((size_t)rows * C * 4);
    ivit_ref_layernorm_ord(x, rows, C, scale, bias_int, sc, z, 1, tokens_per_image);
    const int rc = ivit_cpu_requant_f32(h, z, dy, C, NULL, NULL, 8, out8, rows, C);
    free(z);
    return rc;
}
/* PatchEmbed's tail: int8 conv output -> norm (token-order sums) -> qact (16 bit, per channel) -> qact1 (16 bit, per tensor) */
int ivit_cpu_patch_norm_tokenorder(ivit_handle h, const int8_t *x8, int64_t rows, int C, float scale, const float *bias_int,
                                   const float *sc, const ivit_dyadic *dy_ch, ivit_dyadic dy2, int tokens_per_image,
                                   int16_t *out16) {
    TW_REQ(x8 && bias_int && sc && dy_ch && out16 && rows > 0 && C > 0 && scale > 0.f && tokens_per_image > 0);
    int16_t *x16 = (int16_t *)xmalloc((size_t)rows * C * 2);
    float *z = (float *)xmalloc((size_t)rows * C * 4);
    int16_t *t16 = (int16_t *)xmalloc((size_t)rows * C * 2);
    for (int64_t i = 0; i < rows * C; ++i) x16[i] = x8[i];
    ivit_ref_layernorm_ord(x16, rows, C, scale, bias_int, sc, z, 1, tokens_per_image);
    int rc = ivit_cpu_requant_f32(h, z, dy_ch, C, NULL, NULL, 16, t16, rows, C);
    if (rc == TW_OK) rc = ivit_cpu_requant_i16(h, t16, &dy2, 1, NULL, NULL, 16, out16, rows, C);
    free(x16);
    free(z);
    free(t16);
    return rc;
}

/* PatchMerging's gather (swin_quant.py:336-342): cat([x[0::2,0::2], x[1::2,0::2], x[0::2,1::2], x[1::2,1::2]], -1) */
int ivit_cpu_patch_merge_gather(ivit_handle h, const void *x, int in_bits, int B, int R, int C, int16_t *out) {
    (void)h;
    TW_REQ(x && out && B > 0 && R > 0 && (R % 2) == 0 && C > 0 && (in_bits == 8 || in_bits == 16));
    const int R2 = R / 2;
    for (int64_t b = 0; b < B; ++b)
        for (int y = 0; y < R2; ++y)
            for (int xx = 0; xx < R2; ++xx)
                for (int blk = 0; blk < 4; ++blk) {
                    const int sy = 2 * y + (blk & 1), sx = 2 * xx + (blk >> 1);
                    const int64_t src = ((b * R + sy) * R + sx) * C, dst = (((b * R2 + y) * R2 + xx) * 4 + blk) * (int64_t)C;
                    for (int c = 0; c < C; ++c)
                        out[dst + c] = in_bits == 8 ? (int16_t)((const int8_t *)x)[src + c] : ((const int16_t *)x)[src + c];
                }
    return TW_OK;
}
int ivit_cpu_widen_i8_i16(ivit_handle h, const int8_t *x, int16_t *out, int64_t n) {
    (void)h;
    TW_REQ(x && out && n > 0);
    for (int64_t i = 0; i < n; ++i) out[i] = x[i];
    return TW_OK;
}

/* WindowAttention.forward between the qkv QuantAct and proj (swin_quant.py:121-162) with roll / window partition /
 * reverse as index arithmetic (:18-50, 268-287), one (image, window, head) at a time in the reference's operator order */
int ivit_cpu_window_attention_fused(ivit_handle h, const int8_t *qkv, ivit_dyadic dy_qk, ivit_dyadic dy_a, const int16_t *relb,
                                    float s_softmax, ivit_dyadic dy_pv, int8_t *ctx, int B, int R, int window, int shift,
                                    int heads, int dh) {
    (void)h;
    TW_REQ(qkv && relb && ctx && B > 0 && R > 0 && heads > 0 && window == 7 && dh == 32 && (R % 7) == 0 && shift >= 0 && shift < 7);
    const int nw = R / 7, C = heads * 32, N = 49;
    int8_t q[49 * 32], k[49 * 32], v[49 * 32], a8[49 * 49];
    int32_t S[49 * 49], O[49 * 32];
    uint16_t P[49 * 49];
    float mask[49 * 49];
    int64_t tok[49];
    int reg[49];
    for (int b = 0; b < B; ++b)
        for (int wi = 0; wi < nw; ++wi)
            for (int wj = 0; wj < nw; ++wj) {
                for (int n = 0; n < N; ++n) {
                    const int wy = n / 7, wx = n % 7, ys = wi * 7 + wy, xs = wj * 7 + wx;     /* coordinates in the ROLLED image */
                    tok[n] = ((int64_t)b * R + (ys + shift) % R) * R + (xs + shift) % R;      /* torch.roll(x, -shift): rolled[y] = x[y + shift] */
                    const int ry = ys < R - 7 ? 0 : (ys < R - shift ? 1 : 2), rx = xs < R - 7 ? 0 : (xs < R - shift ? 1 : 2);
                    reg[n] = ry * 3 + rx;                                                      /* img_mask region (:276-287) */
                }
                for (int i = 0; i < N; ++i)
                    for (int j = 0; j < N; ++j) mask[i * N + j] = (shift > 0 && reg[i] != reg[j]) ? -100.0f : 0.0f;
                for (int hd = 0; hd < heads; ++hd) {
                    for (int n = 0; n < N; ++n) {
                        const int8_t *row = qkv + tok[n] * (3 * C) + hd * 32;
                        memcpy(q + n * 32, row, 32);
                        memcpy(k + n * 32, row + C, 32);
                        memcpy(v + n * 32, row + 2 * C, 32);
                    }
                    ivit_ref_bmm_nt_i8(q, k, S, 1, N, N, 32);
                    for (int i = 0; i < N * N; ++i) {
                        const int32_t v1 = tw_clamp(tw_rq((double)S[i], dy_qk), 8);                          /* qact_attn1 */
                        a8[i] = (int8_t)tw_clamp(tw_rq((double)v1, dy_a) + relb[(int64_t)hd * N * N + i], 8);  /* qact2 + bias */
                    }
                    ivit_ref_shiftmax_masked(a8, N, N, N, s_softmax, 8, shift > 0 ? mask : NULL, 1, 1, P, N);
                    for (int i = 0; i < N; ++i)
                        for (int d = 0; d < 32; ++d) {
                            int32_t acc = 0;
                            for (int j = 0; j < N; ++j) acc += (int32_t)P[i * N + j] * (int32_t)v[j * 32 + d];
                            O[i * 32 + d] = acc;
                        }
                    for (int i = 0; i < N; ++i)
                        for (int d = 0; d < 32; ++d)
                            ctx[tok[i] * C + hd * 32 + d] = (int8_t)tw_clamp(tw_rq((double)O[i * 32 + d], dy_pv), 8);   /* qact3 */
                }
            }
    return TW_OK;
}

/* the table form computes the same integers (the tables are checked exhaustively against the arithmetic when built) */
int ivit_cpu_window_attention_fused_lut(ivit_handle h, const int8_t *qkv, ivit_dyadic dy_qk, ivit_dyadic dy_a, const int16_t *relb,
                                        float s_softmax, const uint16_t *exp_aq, const float *exp_t, const uint8_t *exp_cls,
                                        int nclass, int t_count, int dmin, ivit_dyadic dy_pv, int8_t *ctx, int B, int R,
                                        int window, int shift, int heads, int dh) {
    TW_REQ(exp_aq && exp_t && exp_cls && nclass >= 1 && t_count >= 1 && dmin <= 0);
    return ivit_cpu_window_attention_fused(h, qkv, dy_qk, dy_a, relb, s_softmax, dy_pv, ctx, B, R, window, shift, heads, dh);
}

/* planned fused Mlp: the twin keeps the two linear plans; the planned call is the unplanned chain */
struct ivit_cpu_mlp_plan_s { const struct ivit_cpu_plan_s *fc1, *fc2; };
int ivit_cpu_mlp_plan_create(ivit_handle h, ivit_linear_plan fc1, ivit_linear_plan fc2, ivit_mlp_plan *out) {
    (void)h;
    const struct ivit_cpu_plan_s *p1 = (const struct ivit_cpu_plan_s *)fc1, *p2 = (const struct ivit_cpu_plan_s *)fc2;
    TW_REQ(p1 && p2 && out);
    if (p1->K != 384 || p1->N != 1536 || p2->K != 1536 || p2->N != 384) return 3;      /* IVIT_ERR_UNSUPPORTED, like the HIP side */
    struct ivit_cpu_mlp_plan_s *p = (struct ivit_cpu_mlp_plan_s *)xmalloc(sizeof(*p));
    p->fc1 = p1; p->fc2 = p2;
    *out = (ivit_mlp_plan)p;
    return TW_OK;
}
int ivit_cpu_mlp_plan_destroy(ivit_mlp_plan p) { free(p); return TW_OK; }
int ivit_cpu_mlp_fused_planned(ivit_handle h, ivit_mlp_plan p, const int8_t *x, const int8_t *gelu_table, ivit_dyadic dy_main,
                               ivit_dyadic dy_res, const int16_t *residual, int16_t *out, int64_t M) {
    const struct ivit_cpu_mlp_plan_s *pl = (const struct ivit_cpu_mlp_plan_s *)p;
    TW_REQ(pl && x && gelu_table && residual && out && M > 0);
    return ivit_cpu_mlp_fused(h, x, pl->fc1->w, pl->fc1->bias, pl->fc1->dy, gelu_table, pl->fc2->w, pl->fc2->bias, pl->fc2->dy,
                              dy_main, dy_res, residual, out, M, pl->fc1->K, pl->fc1->N);
}

/* ---- N3: the uint8 front end (utils/data_utils.py:82-91).  Same fp32 operation order as the device kernels and as
 * oracle/oracle.py::resize_center_crop_u8 (which tests/golden/resize.npz pins against torch's antialiased bicubic). */
int ivit_cpu_normalize_quantize_u8(ivit_handle h, const uint8_t *hwc, int B, int H, int W, const float mean_host[3],
                                   const float std_host[3], float scale, int8_t *nchw) {
    (void)h;
    TW_REQ(hwc && nchw && mean_host && std_host && B > 0 && H > 0 && W > 0 && scale > 0.f);
    TW_REQ(std_host[0] != 0.f && std_host[1] != 0.f && std_host[2] != 0.f);
    const volatile float inv = 1.0f / scale;
    const int64_t HW = (int64_t)H * W;
    for (int64_t b = 0; b < B; ++b)
        for (int64_t p = 0; p < HW; ++p)
            for (int c = 0; c < 3; ++c) {
                volatile float v = (float)hwc[(b * HW + p) * 3 + c] / 255.0f;     /* ToTensor */
                v = v - mean_host[c];                                             /* Normalize */
                v = v / std_host[c];
                float r = rintf(inv * v);                                         /* input QuantAct (quant_utils.py:12-48) */
                r = r < -128.f ? -128.f : (r > 127.f ? 127.f : r);
                nchw[(b * 3 + c) * HW + p] = (int8_t)(int)r;
            }
    return TW_OK;
}

static float tw_cubic_aa(float x) {
    const float a = -0.5f;
    x = fabsf(x);
    if (x < 1.0f) return ((a + 2.0f) * x - (a + 3.0f)) * x * x + 1.0f;
    if (x < 2.0f) return (((a * x) - (5.0f * a)) * x + (8.0f * a)) * x - (4.0f * a);
    return 0.0f;
}
typedef struct { int xmin, xsize; float center, invscale, total; } tw_taps;
static tw_taps tw_aa_taps(int i, int in_size, int out_size) {
    tw_taps t;
    const float scale = (float)in_size / (float)out_size;
    const float support = scale >= 1.0f ? 2.0f * scale : 2.0f;
    t.invscale = scale >= 1.0f ? 1.0f / scale : 1.0f;
    t.center = scale * ((float)i + 0.5f);
    int lo = (int)(t.center - support + 0.5f), hi = (int)(t.center + support + 0.5f);
    t.xmin = lo > 0 ? lo : 0;
    t.xsize = (hi < in_size ? hi : in_size) - t.xmin;
    float tot = 0.0f;
    for (int j = 0; j < t.xsize; ++j) tot += tw_cubic_aa(((float)(j + t.xmin) - t.center + 0.5f) * t.invscale);
    t.total = tot;
    return t;
}
static float tw_tap_w(const tw_taps *t, int j) {
    return tw_cubic_aa(((float)(j + t->xmin) - t->center + 0.5f) * t->invscale) / t->total;
}
int ivit_cpu_resize_center_crop_u8(ivit_handle h, const uint8_t *hwc, int B, int H0, int W0, int size, int crop,
                                   float *workspace, uint8_t *out_hwc) {
    (void)h;
    TW_REQ(hwc && workspace && out_hwc && B > 0 && H0 > 0 && W0 > 0 && size > 0 && crop > 0);
    int Hr, Wr;
    if (H0 <= W0) { Hr = size; Wr = (int)((int64_t)size * W0 / H0); }
    else { Wr = size; Hr = (int)((int64_t)size * H0 / W0); }
    TW_REQ(crop <= Hr && crop <= Wr);
    const int top = (int)rint((Hr - crop) / 2.0), left = (int)rint((Wr - crop) / 2.0);
    for (int xo = 0; xo < crop; ++xo) {                       /* horizontal pass, cropped columns only */
        const tw_taps t = tw_aa_taps(xo + left, W0, Wr);
        for (int64_t by = 0; by < (int64_t)B * H0; ++by) {
            const uint8_t *row = hwc + (by * W0 + t.xmin) * 3;
            float w = tw_tap_w(&t, 0);
            float a0 = (float)row[0] * w, a1 = (float)row[1] * w, a2 = (float)row[2] * w;
            for (int j = 1; j < t.xsize; ++j) {
                w = tw_tap_w(&t, j);
                a0 += (float)row[j * 3] * w; a1 += (float)row[j * 3 + 1] * w; a2 += (float)row[j * 3 + 2] * w;
            }
            float *o = workspace + (by * crop + xo) * 3;
            o[0] = a0; o[1] = a1; o[2] = a2;
        }
    }
    const int64_t rs = (int64_t)crop * 3;
    for (int yo = 0; yo < crop; ++yo) {                       /* vertical pass, rne, clamp */
        const tw_taps t = tw_aa_taps(yo + top, H0, Hr);
        for (int64_t b = 0; b < B; ++b)
            for (int xo = 0; xo < crop; ++xo) {
                const float *col = workspace + ((b * H0 + t.xmin) * crop + xo) * 3;
                float w = tw_tap_w(&t, 0);
                float a0 = col[0] * w, a1 = col[1] * w, a2 = col[2] * w;
                for (int j = 1; j < t.xsize; ++j) {
                    w = tw_tap_w(&t, j);
                    a0 += col[j * rs] * w; a1 += col[j * rs + 1] * w; a2 += col[j * rs + 2] * w;
                }
                uint8_t *o = out_hwc + ((b * crop + yo) * crop + xo) * 3;
                const float r0 = rintf(a0), r1 = rintf(a1), r2 = rintf(a2);
                o[0] = (uint8_t)(int)(r0 < 0.f ? 0.f : (r0 > 255.f ? 255.f : r0));
                o[1] = (uint8_t)(int)(r1 < 0.f ? 0.f : (r1 > 255.f ? 255.f : r1));
                o[2] = (uint8_t)(int)(r2 < 0.f ? 0.f : (r2 > 255.f ? 255.f : r2));
            }
    }
    return TW_OK;
}

/* ---- a9-a11: the whole-model runner (VisionTransformer.forward, vit_quant.py:254-282), as the chain of twinned operators
 * in the order csrc/ivit_model.h::run_slice issues them.  Host pointers in ivit_vit_params / ivit_vit_block; the Shiftmax
 * tables of a block are ignored (the arithmetic Shiftmax is what they are checked against); slices are a device notion:
 * any nslices gives the same integers; the workspace is not used. */
struct ivit_vit_s { ivit_vit_config cfg; ivit_vit_params prm; ivit_vit_block *blocks; };
int ivit_cpu_vit_create(ivit_handle h, const ivit_vit_config *cfg, const ivit_vit_params *params, int max_slices, ivit_vit *out) {
    (void)h;
    TW_REQ(cfg && params && out && params->blocks_host && max_slices >= 1 && cfg->depth > 0 && cfg->num_heads > 0);
    TW_REQ(cfg->img_size % cfg->patch_size == 0 && cfg->embed_dim % cfg->num_heads == 0);
    struct ivit_vit_s *m = (struct ivit_vit_s *)xmalloc(sizeof *m);
    m->cfg = *cfg; m->prm = *params;
    m->blocks = (ivit_vit_block *)xmalloc(sizeof(ivit_vit_block) * cfg->depth);
    memcpy(m->blocks, params->blocks_host, sizeof(ivit_vit_block) * cfg->depth);
    *out = m;
    return TW_OK;
}
int ivit_cpu_vit_destroy(ivit_vit m) { if (!m) return TW_INVALID; free(m->blocks); free(m); return TW_OK; }
int ivit_cpu_vit_workspace_bytes(ivit_vit m, int batch, int nslices, size_t *bytes) {
    TW_REQ(m && bytes && batch > 0 && nslices >= 1 && nslices <= batch);
    *bytes = 256;
    return TW_OK;
}
int ivit_cpu_vit_workspace_init(ivit_vit m, void *workspace, size_t bytes, int batch, int nslices) {
    TW_REQ(m && workspace && bytes >= 256 && batch > 0 && nslices >= 1);
    return TW_OK;
}
int ivit_cpu_vit_forward(ivit_vit m, const int8_t *images, int batch, int nslices, void *workspace, size_t bytes, int32_t *logits) {
    (void)workspace; (void)bytes;
    TW_REQ(m && images && logits && batch > 0 && nslices >= 1 && nslices <= batch);
    const ivit_vit_config *c = &m->cfg;
    const ivit_vit_params *P = &m->prm;
    const int g = c->img_size / c->patch_size, np_ = g * g, T = np_ + 1, D = c->embed_dim, H = c->num_heads, dh = D / H;
    const int Hd = c->hidden_dim, ld = (T + 15) / 16 * 16, Kp = c->in_chans * c->patch_size * c->patch_size, B = batch;
    const int64_t M = (int64_t)B * T;
    int8_t *patches = (int8_t *)xmalloc((size_t)B * np_ * Kp), *a8 = (int8_t *)xmalloc((size_t)M * D);
    int8_t *q = (int8_t *)xmalloc((size_t)M * D), *k = (int8_t *)xmalloc((size_t)M * D);
    int8_t *vt = (int8_t *)calloc((size_t)B * H * dh * ld, 1), *ctx8 = (int8_t *)xmalloc((size_t)M * D);
    int8_t *h8 = (int8_t *)xmalloc((size_t)M * Hd), *g8 = (int8_t *)xmalloc((size_t)M * Hd), *cls8 = (int8_t *)xmalloc((size_t)B * D);
    int16_t *patch16 = (int16_t *)xmalloc((size_t)B * np_ * D * 2), *x = (int16_t *)xmalloc((size_t)M * D * 2), *y = (int16_t *)xmalloc((size_t)M * D * 2);
    int rc = TW_OK;
#define TW_RUN(call) do { if (rc == TW_OK) rc = (call); } while (0)
    TW_RUN(ivit_cpu_im2col_patch(NULL, images, B, c->in_chans, c->img_size, c->img_size, c->patch_size, patches));
    TW_RUN(ivit_cpu_linear_i8_requant(NULL, patches, P->pe_w, P->pe_b, P->pe_dy, 16, patch16, B * np_, D, Kp));
    TW_RUN(ivit_cpu_embed_finish(NULL, patch16, P->z_cls, P->pos, P->dy_x, P->dy_pos, x, B, T, D));
    for (int i = 0; i < c->depth; ++i) {
        const ivit_vit_block *b = &m->blocks[i];
        TW_RUN(ivit_cpu_layernorm_requant(NULL, x, M, D, D, b->s_ln1, b->n1_bias_int, b->n1_sc, b->n1_dy, a8));
        TW_RUN(ivit_cpu_linear_i8_qkv(NULL, a8, b->qkv_w, b->qkv_b, b->qkv_dy, q, k, vt, B, T, H, dh, ld));
        TW_RUN(ivit_cpu_attention_fused(NULL, q, k, vt, b->dy_qk, b->s_softmax, b->dy_pv, ctx8, B, H, T, dh, ld));
        TW_RUN(ivit_cpu_linear_i8_requant_residual(NULL, ctx8, b->proj_w, b->proj_b, b->proj_dy, b->res1_main, b->res1_res, x, y, (int)M, D, D));
        { int16_t *t = x; x = y; y = t; }
        TW_RUN(ivit_cpu_layernorm_requant(NULL, x, M, D, D, b->s_ln2, b->n2_bias_int, b->n2_sc, b->n2_dy, a8));
        TW_RUN(ivit_cpu_linear_i8_requant(NULL, a8, b->fc1_w, b->fc1_b, b->fc1_dy, 8, h8, (int)M, Hd, D));
        TW_RUN(ivit_cpu_shiftgelu_requant(NULL, h8, M, Hd, b->s_gelu, b->dy_gelu, g8));
        TW_RUN(ivit_cpu_linear_i8_requant_residual(NULL, g8, b->fc2_w, b->fc2_b, b->fc2_dy, b->res2_main, b->res2_res, x, y, (int)M, D, Hd));
        { int16_t *t = x; x = y; y = t; }
    }
    /* final norm on the class-token rows only (row stride T*D), then the head's int32 accumulators */
    TW_RUN(ivit_cpu_layernorm_requant(NULL, x, B, D, (int64_t)T * D, P->s_ln, P->n_bias_int, P->n_sc, P->n_dy, cls8));
    TW_RUN(ivit_cpu_linear_i8(NULL, cls8, P->head_w, P->head_b, logits, B, c->num_classes, D));
    free(patches); free(a8); free(q); free(k); free(vt); free(ctx8); free(h8); free(g8); free(cls8); free(patch16); free(x); free(y);
    return rc;
}

/* ---- a12: SwinTransformer.forward (swin_quant.py:539-564) as csrc/ivit_model.h::swin_run_slice chains the operators */
struct ivit_swin_s { ivit_swin_config cfg; ivit_swin_params prm; ivit_swin_block *blocks; ivit_swin_merge *merges; int nblocks; ivit_dyadic dy_qact1; };
int ivit_cpu_swin_create(ivit_handle h, const ivit_swin_config *cfg, const ivit_swin_params *params, int max_slices, ivit_swin *out) {
    (void)h;
    TW_REQ(cfg && params && out && params->blocks_host && params->dy_qact1 && max_slices >= 1);
    TW_REQ(cfg->num_layers >= 1 && cfg->num_layers <= 4 && cfg->window_size == 7);
    int nb = 0;
    for (int li = 0; li < cfg->num_layers; ++li) {
        if (cfg->num_heads[li] <= 0 || ((cfg->embed_dim << li) / cfg->num_heads[li]) != 32) return 3;     /* IVIT_ERR_UNSUPPORTED */
        nb += cfg->depths[li];
    }
    struct ivit_swin_s *m = (struct ivit_swin_s *)xmalloc(sizeof *m);
    m->cfg = *cfg; m->prm = *params; m->nblocks = nb; m->dy_qact1 = params->dy_qact1[0];
    m->blocks = (ivit_swin_block *)xmalloc(sizeof(ivit_swin_block) * nb);
    memcpy(m->blocks, params->blocks_host, sizeof(ivit_swin_block) * nb);
    m->merges = (ivit_swin_merge *)xmalloc(sizeof(ivit_swin_merge) * (cfg->num_layers > 1 ? cfg->num_layers - 1 : 1));
    if (cfg->num_layers > 1) memcpy(m->merges, params->merges_host, sizeof(ivit_swin_merge) * (cfg->num_layers - 1));
    *out = m;
    return TW_OK;
}
int ivit_cpu_swin_destroy(ivit_swin m) { if (!m) return TW_INVALID; free(m->blocks); free(m->merges); free(m); return TW_OK; }
int ivit_cpu_swin_workspace_bytes(ivit_swin m, int batch, int nslices, size_t *bytes) {
    TW_REQ(m && bytes && batch > 0 && nslices >= 1 && nslices <= batch);
    *bytes = 256;
    return TW_OK;
}
int ivit_cpu_swin_forward(ivit_swin m, const int8_t *images, int batch, int nslices, void *workspace, size_t bytes, int32_t *logits) {
    (void)workspace; (void)bytes;
    TW_REQ(m && images && logits && batch > 0 && nslices >= 1 && nslices <= batch);
    const ivit_swin_config *c = &m->cfg;
    const ivit_swin_params *P = &m->prm;
    const int E = c->embed_dim, B = batch, Kp = c->in_chans * c->patch_size * c->patch_size;
    int res = c->img_size / c->patch_size, L = res * res;
    int64_t M = (int64_t)B * L;
    const size_t M0 = (size_t)M;
    int8_t *patches = (int8_t *)xmalloc(M0 * Kp), *a8 = (int8_t *)xmalloc(M0 * E), *qkv = (int8_t *)xmalloc(M0 * 3 * E);
    int8_t *ctx = (int8_t *)xmalloc(M0 * E), *h8 = (int8_t *)xmalloc(M0 * c->mlp_ratio * E), *g8 = (int8_t *)xmalloc(M0 * c->mlp_ratio * E);
    int8_t *pool = (int8_t *)xmalloc((size_t)B * (E << (c->num_layers - 1)));
    int16_t *x = (int16_t *)xmalloc(M0 * E * 2), *y = (int16_t *)xmalloc(M0 * E * 2), *t16 = (int16_t *)xmalloc(M0 * E * 2);
    int rc = TW_OK;
    TW_RUN(ivit_cpu_im2col_patch(NULL, images, B, c->in_chans, c->img_size, c->img_size, c->patch_size, patches));
    TW_RUN(ivit_cpu_linear_i8_requant(NULL, patches, P->pe.w, P->pe.b, P->pe.dy, 8, a8, (int)M, E, Kp));
    TW_RUN(ivit_cpu_patch_norm_tokenorder(NULL, a8, M, E, P->s_bn, P->pn.bias_int, P->pn.sc, P->pn.dy, m->dy_qact1, L, x));
    int bi = 0;
    for (int li = 0; li < c->num_layers; ++li) {
        const int C = E << li, heads = c->num_heads[li];
        for (int bj = 0; bj < c->depths[li]; ++bj, ++bi) {
            const ivit_swin_block *b = &m->blocks[bi];
            const int shift = (bj % 2 == 0 || res <= c->window_size) ? 0 : c->window_size / 2;
            if (li == 0) TW_RUN(ivit_cpu_layernorm_tokenorder_requant(NULL, x, M, C, b->s_in, b->n1.bias_int, b->n1.sc, b->n1.dy, L, a8));
            else TW_RUN(ivit_cpu_layernorm_requant(NULL, x, M, C, C, b->s_in, b->n1.bias_int, b->n1.sc, b->n1.dy, a8));
            TW_RUN(ivit_cpu_linear_i8_requant(NULL, a8, b->qkv.w, b->qkv.b, b->qkv.dy, 8, qkv, (int)M, 3 * C, C));
            if (b->exp_aq)
                TW_RUN(ivit_cpu_window_attention_fused_lut(NULL, qkv, b->dy_qk, b->dy_a, b->relb, b->s_softmax, b->exp_aq, b->exp_t, b->exp_cls, b->exp_nc,
                                                           b->exp_tcount, b->exp_dmin, b->dy_pv, ctx, B, res, c->window_size, shift, heads, C / heads));
            else
                TW_RUN(ivit_cpu_window_attention_fused(NULL, qkv, b->dy_qk, b->dy_a, b->relb, b->s_softmax, b->dy_pv, ctx, B, res, c->window_size, shift, heads, C / heads));
            TW_RUN(ivit_cpu_linear_i8_requant_residual(NULL, ctx, b->proj.w, b->proj.b, b->proj.dy, b->res1_main, b->res1_res, x, y, (int)M, C, C));
            { int16_t *t = x; x = y; y = t; }
            if (li == 0) TW_RUN(ivit_cpu_layernorm_tokenorder_requant(NULL, x, M, C, b->s_mid, b->n2.bias_int, b->n2.sc, b->n2.dy, L, a8));
            else TW_RUN(ivit_cpu_layernorm_requant(NULL, x, M, C, C, b->s_mid, b->n2.bias_int, b->n2.sc, b->n2.dy, a8));
            TW_RUN(ivit_cpu_linear_i8_requant(NULL, a8, b->fc1.w, b->fc1.b, b->fc1.dy, 8, h8, (int)M, c->mlp_ratio * C, C));
            TW_RUN(ivit_cpu_shiftgelu_requant(NULL, h8, M, c->mlp_ratio * C, b->s_gelu, b->dy_gelu, g8));
            TW_RUN(ivit_cpu_linear_i8_requant_residual(NULL, g8, b->fc2.w, b->fc2.b, b->fc2.dy, b->res2_main, b->res2_res, x, y, (int)M, C, c->mlp_ratio * C));
            { int16_t *t = x; x = y; y = t; }
        }
        if (li < c->num_layers - 1) {     /* PatchMerging: gather -> LN(4C) -> qact1(8) -> reduction -> qact2(8) */
            const ivit_swin_merge *g = &m->merges[li];
            TW_RUN(ivit_cpu_patch_merge_gather(NULL, x, 16, B, res, C, t16));
            res /= 2; L = res * res; M = (int64_t)B * L;
            TW_RUN(ivit_cpu_layernorm_requant(NULL, t16, M, 4 * C, 4 * C, g->s_in, g->n.bias_int, g->n.sc, g->n.dy, a8));
            TW_RUN(ivit_cpu_linear_i8_requant(NULL, a8, g->red.w, NULL, g->red.dy, 8, ctx, (int)M, 2 * C, 4 * C));
            TW_RUN(ivit_cpu_widen_i8_i16(NULL, ctx, x, M * 2 * C));
        }
    }
    const int Cl = E << (c->num_layers - 1);
    TW_RUN(ivit_cpu_layernorm_requant(NULL, x, M, Cl, Cl, P->s_norm_in, P->n.bias_int, P->n.sc, P->n.dy, a8));
    TW_RUN(ivit_cpu_avgpool_requant(NULL, a8, B, L, Cl, P->dy_pool, pool));
    TW_RUN(ivit_cpu_linear_i8(NULL, pool, P->head_w, P->head_b, logits, B, c->num_classes, Cl));
    free(patches); free(a8); free(qkv); free(ctx); free(h8); free(g8); free(pool); free(x); free(y); free(t16);
    return rc;
}
#undef TW_RUN
