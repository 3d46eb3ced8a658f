// ivit_model.h — native runner for the frozen integer DeiT/ViT forward (include/ivit.h,
// "whole-model runner").  Chains the C-ABI entry points of ivit_hip.hip in the order of the
// reference's VisionTransformer.forward (vit_quant.py:254-282); owns the ShiftGELU tables, the
// slice streams/events and the optional hipGraph.  Included at the end of ivit_hip.hip.
#pragma once
#include <vector>

struct ivit_vit_s {
    ivit_handle h;                    // the caller's handle (its stream is the parent stream)
    ivit_vit_config cfg;
    ivit_vit_params prm;
    std::vector<ivit_vit_block> blocks;
    int T, ld, Kp, num_patches;
    bool fused_attention;
    int8_t *gelu_tab;                 // [depth][65536]
    float *rowtab;                    // [depth][256][64] Shiftmax row tables (ivit_shiftmax_rowtable) or null
    std::vector<char> has_rowtab;     // per block: its scale's table lines fit 64 entries and the multipliers are in the fast range
    std::vector<ivit_linear_plan> plans;   // per block: qkv, proj, fc1, fc2 (frozen QuantLinear plans, ivit_linear_plan_create)
    std::vector<ivit_mlp_plan> mlp_plans;  // per block: fused Mlp plan (D = 384), or null -> fc1 / ShiftGELU / fc2 launches
    int max_slices;
    std::vector<ivit_handle> slice_h; // one handle per internal stream
    std::vector<hipStream_t> streams;
    std::vector<hipEvent_t> done;
    hipEvent_t fork;
};

struct ivit_graph_s {
    ivit_handle h;
    hipGraph_t graph;
    hipGraphExec_t exec;
};

#ifndef IVIT_OPT_V_ROWMAJOR
#define IVIT_OPT_V_ROWMAJOR 1          // A/B: v row-major between the qkv GEMM and the row-table attention
#endif
#ifndef IVIT_OPT_LN_QKV
#define IVIT_OPT_LN_QKV 1              // A/B: norm1 inside the qkv GEMM's prologue (ivit_layernorm_linear_i8_qkv_planned) where v is row-major
#endif
#ifndef IVIT_OPT_PROJ_WS
#define IVIT_OPT_PROJ_WS 1             // A/B: attn.proj + qact2 of a D = 384 block on gemm_ws_qkv_kernel (prepared plan)
#endif
#ifndef IVIT_OPT_PROJ_LN
#define IVIT_OPT_PROJ_LN 0              // A/B: norm2 in the tail of the attn.proj launch (ivit_linear_i8_requant_residual_layernorm_planned): same-box
                                       // 2.785 ms against 2.657 with norm2 as its own launch (the tail runs behind a barrier, two waves per SIMD)
#endif
#ifndef IVIT_OPT_LN_MLP
#define IVIT_OPT_LN_MLP 1              // A/B: norm2 in the head of the fused Mlp's launch (ivit_layernorm_mlp_fused_planned)
#endif
#ifndef IVIT_OPT_ATTN_ROWTAB
#define IVIT_OPT_ATTN_ROWTAB 1         // A/B: Shiftmax by row tables (one gather per score) where a layer's table lines fit
#endif

namespace {

inline size_t al256(size_t n) { return (n + 255) & ~(size_t)255; }

// byte offsets of the per-slice buffers for `B` images
struct SliceLayout {
    size_t patches, patch16, xa, xb, a8, q, k, vt, ctx8, h8, g8, cls8, s8, p16, total;
};

SliceLayout slice_layout(const ivit_vit_s *m, int B) {
    const ivit_vit_config &c = m->cfg;
    const size_t M = (size_t)B * m->T, D = c.embed_dim, H = c.num_heads, dh = D / H, Hd = c.hidden_dim;
    SliceLayout L;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o = al256(o + bytes); return at; };
    L.patches = take((size_t)B * m->num_patches * m->Kp);
    L.patch16 = take((size_t)B * m->num_patches * D * 2);
    L.xa = take(M * D * 2);
    L.xb = take(M * D * 2);
    L.a8 = take(M * D);
    L.q = take((size_t)B * H * m->T * dh);
    L.k = take((size_t)B * H * m->T * dh);
    L.vt = take((size_t)B * H * dh * m->ld);
    L.ctx8 = take(M * D);
    L.h8 = take(M * Hd);
    L.g8 = take(M * Hd);
    L.cls8 = take((size_t)B * D);
    L.s8 = L.p16 = 0;
    if (!m->fused_attention) {
        L.s8 = take((size_t)B * H * m->T * m->ld);
        L.p16 = take((size_t)B * H * m->T * m->ld * 2);
    }
    L.total = o;
    return L;
}

#ifndef IVIT_OPT_SLICE_CU_SHARE
#define IVIT_OPT_SLICE_CU_SHARE 1       // A/B: a slice's persistent D = 384 kernels sized for num_cu / slices (ivit_ctx::cu_share)
#endif
inline int slice_begin(int batch, int nslices, int i) { return (int)(((long long)batch * i) / nslices); }
inline int max_slice(int batch, int nslices) { return (batch + nslices - 1) / nslices; }

// one slice on handle `h`
// `Bmax`: images of the LARGEST slice of this forward — every slice uses that slice's buffer layout, so the regions
// zeroed by ivit_vit_workspace_init are the ones the kernels see whatever the (ragged) slice sizes are
int run_slice(const ivit_vit_s *m, ivit_handle h, const int8_t *images, int B, int Bmax, char *ws, int32_t *logits) {
    const ivit_vit_config &c = m->cfg;
    const ivit_vit_params &P = m->prm;
    const int T = m->T, D = c.embed_dim, H = c.num_heads, dh = D / H, Hd = c.hidden_dim, ld = m->ld;
    const int M = B * T;
    const SliceLayout L = slice_layout(m, Bmax);
    int8_t *patches = (int8_t *)(ws + L.patches), *a8 = (int8_t *)(ws + L.a8), *q = (int8_t *)(ws + L.q),
           *k = (int8_t *)(ws + L.k), *vt = (int8_t *)(ws + L.vt), *ctx8 = (int8_t *)(ws + L.ctx8),
           *h8 = (int8_t *)(ws + L.h8), *g8 = (int8_t *)(ws + L.g8), *cls8 = (int8_t *)(ws + L.cls8);
    int16_t *patch16 = (int16_t *)(ws + L.patch16), *x = (int16_t *)(ws + L.xa), *y = (int16_t *)(ws + L.xb);
    int rc;
#define RUN(call) do { rc = (call); if (rc != IVIT_OK) { if (h != m->h) snprintf(m->h->err, sizeof(m->h->err), "%s", h->err); return rc; } } while (0)
    // PatchEmbed + class token + position embedding: one GEMM launch that gathers its A rows from the images and finishes the rows in its
    // epilogue (round 6), or im2col -> GEMM -> embed_finish where that form does not apply
    rc = (m->Kp == c.in_chans * c.patch_size * c.patch_size)
             ? ivit_patch_embed(h, images, B, c.in_chans, c.img_size, c.img_size, c.patch_size, P.pe_w, P.pe_b, P.pe_dy, P.z_cls, P.pos, P.dy_x, P.dy_pos, x, D)
             : IVIT_ERR_UNSUPPORTED;
    if (rc == IVIT_ERR_UNSUPPORTED) {
        RUN(ivit_im2col_patch(h, images, B, c.in_chans, c.img_size, c.img_size, c.patch_size, patches));
        RUN(ivit_linear_i8_requant(h, patches, P.pe_w, P.pe_b, P.pe_dy, 16, patch16, B * m->num_patches, D, m->Kp));
        RUN(ivit_embed_finish(h, patch16, P.z_cls, P.pos, P.dy_x, P.dy_pos, x, B, T, D));
    } else {
        RUN(rc);
    }
    for (int i = 0; i < c.depth; ++i) {
        const ivit_vit_block &b = m->blocks[i];
        // a layer on the row-table attention takes v ROW-major (ldv = 0: the qkv GEMM stores 16 bytes per lane instead of 16 byte
        // stores, the attention kernel transposes on its way into the LDS); the other attention forms read v^T
        const int ldv = (m->fused_attention && m->has_rowtab[i] && IVIT_OPT_V_ROWMAJOR) ? 0 : ld;
        // norm1's 8-bit output has one consumer: where the qkv GEMM keeps a CU's tokens in LDS it is computed there (round 6)
        rc = (IVIT_OPT_LN_QKV && ldv == 0) ? ivit_layernorm_linear_i8_qkv_planned(h, m->plans[4 * i], x, b.s_ln1, b.n1_bias_int, b.n1_sc, b.n1_dy,
                                                                                  q, k, vt, B, T, H, dh)
                                           : IVIT_ERR_UNSUPPORTED;
        if (rc == IVIT_ERR_UNSUPPORTED) {
            RUN(ivit_layernorm_requant(h, x, M, D, D, b.s_ln1, b.n1_bias_int, b.n1_sc, b.n1_dy, a8));
            RUN(ivit_linear_i8_qkv_planned(h, m->plans[4 * i], a8, q, k, vt, B, T, H, dh, ldv));
        } else {
            RUN(rc);
        }
        if (m->fused_attention) {
            if (m->has_rowtab[i])      // one gather per score (round 6)
                RUN(ivit_attention_fused_rowlut(h, q, k, vt, b.dy_qk, b.s_softmax, m->rowtab + (size_t)i * 256 * 64, b.exp_dmin, b.dy_pv,
                                                ctx8, B, H, T, dh, ldv));
            else if (b.exp_aq)
                RUN(ivit_attention_fused_lut(h, q, k, vt, b.dy_qk, b.s_softmax, b.exp_aq, b.exp_t, b.exp_cls, b.exp_nc,
                                             b.exp_tcount, b.exp_dmin, b.dy_pv, ctx8, B, H, T, dh, ld));
            else
                RUN(ivit_attention_fused(h, q, k, vt, b.dy_qk, b.s_softmax, b.dy_pv, ctx8, B, H, T, dh, ld));
        } else {
            int8_t *s8 = (int8_t *)(ws + L.s8);
            uint16_t *p16 = (uint16_t *)(ws + L.p16);
            RUN(ivit_attn_qk_requant(h, q, k, b.dy_qk, s8, B * H, T, dh, ld));
            RUN(ivit_shiftmax(h, s8, (int64_t)B * H * T, T, ld, b.s_softmax, 16, p16, ld));
            RUN(ivit_attn_pv_requant(h, p16, vt, b.dy_pv, ctx8, B, H, T, dh, ld, ld));
        }
        // attn.proj + qact2 with the identity branch, then norm2 + qact3 and the Mlp.  norm2 can ride in the TAIL of the proj launch (the kernel
        // owns whole rows per workgroup: built, bit-exact, slower, IVIT_OPT_PROJ_LN = 0) or in the HEAD of the fused Mlp's launch
        // (ivit_layernorm_mlp_fused_planned, IVIT_OPT_LN_MLP)
        const bool mlp_fast = m->mlp_plans[i] && fabs(b.res2_main.m * b.res2_main.r) < RQ_FAST_CLIM &&
                              fabs(b.res2_res.m * b.res2_res.r) < RQ_FAST_CLIM;
        rc = IVIT_OPT_PROJ_LN ? ivit_linear_i8_requant_residual_layernorm_planned(h, m->plans[4 * i + 1], ctx8, b.res1_main, b.res1_res, x, y, M,
                                                                                b.s_ln2, b.n2_bias_int, b.n2_sc, b.n2_dy, a8)
                              : IVIT_ERR_UNSUPPORTED;
        bool ln2_done = rc != IVIT_ERR_UNSUPPORTED;
        if (!ln2_done) RUN(ivit_linear_i8_requant_residual_planned(h, m->plans[4 * i + 1], ctx8, b.res1_main, b.res1_res, x, y, M));
        else RUN(rc);
        { int16_t *t = x; x = y; y = t; }
        bool mlp_done = false;
        if (!ln2_done && mlp_fast && IVIT_OPT_LN_MLP) {
            rc = ivit_layernorm_mlp_fused_planned(h, m->mlp_plans[i], x, b.s_ln2, b.n2_bias_int, b.n2_sc, b.n2_dy, a8, m->gelu_tab + (size_t)i * 65536,
                                                  b.res2_main, b.res2_res, y, M);
            if (rc != IVIT_ERR_UNSUPPORTED) { RUN(rc); ln2_done = mlp_done = true; }
        }
        if (!ln2_done) RUN(ivit_layernorm_requant(h, x, M, D, D, b.s_ln2, b.n2_bias_int, b.n2_sc, b.n2_dy, a8));
        if (mlp_done) {
        } else if (mlp_fast) {     // hidden tensor stays in LDS
            RUN(ivit_mlp_fused_planned(h, m->mlp_plans[i], a8, m->gelu_tab + (size_t)i * 65536, b.res2_main, b.res2_res, x, y, M));
        } else {
            RUN(ivit_linear_i8_requant_planned(h, m->plans[4 * i + 2], a8, 8, h8, M));
            RUN(ivit_shiftgelu_requant_lut(h, h8, M, Hd, m->gelu_tab + (size_t)i * 65536, g8));
            RUN(ivit_linear_i8_requant_residual_planned(h, m->plans[4 * i + 3], g8, b.res2_main, b.res2_res, x, y, M));
        }
        { int16_t *t = x; x = y; y = t; }
    }
    // final norm on the class-token rows only (row stride T*D), then the head's int32 accumulators
    RUN(ivit_layernorm_requant(h, x, B, D, (int64_t)T * D, P.s_ln, P.n_bias_int, P.n_sc, P.n_dy, cls8));
    RUN(ivit_linear_i8(h, cls8, P.head_w, P.head_b, logits, B, c.num_classes, D));
#undef RUN
    return IVIT_OK;
}

}  // namespace

extern "C" {

int ivit_vit_create(ivit_handle h, const ivit_vit_config *cfg, const ivit_vit_params *params, int max_slices,
                    ivit_vit *out) {
    CHECK_H(h);
    REQUIRE(h, cfg && params && out && params->blocks_host, "null argument");
    REQUIRE(h, cfg->depth > 0 && cfg->embed_dim > 0 && cfg->num_heads > 0 && cfg->embed_dim % cfg->num_heads == 0 &&
                   cfg->patch_size > 0 && cfg->img_size % cfg->patch_size == 0 && cfg->hidden_dim > 0 &&
                   cfg->num_classes > 0 && cfg->in_chans > 0,
            "bad model configuration");
    REQUIRE(h, max_slices >= 1 && max_slices <= 16, "max_slices must be in [1, 16]");
    ivit_vit_s *m = new (std::nothrow) ivit_vit_s();
    if (!m) return IVIT_ERR_HIP;
    m->h = h;
    m->cfg = *cfg;
    m->prm = *params;
    m->blocks.assign(params->blocks_host, params->blocks_host + cfg->depth);
    m->prm.blocks_host = m->blocks.data();
    const int g = cfg->img_size / cfg->patch_size;
    m->num_patches = g * g;
    m->T = m->num_patches + 1;
    m->ld = (m->T + 15) / 16 * 16;
    m->Kp = cfg->in_chans * cfg->patch_size * cfg->patch_size;
    m->fused_attention = (cfg->embed_dim / cfg->num_heads == 64) && m->T <= 640;
    m->gelu_tab = nullptr;
    m->rowtab = nullptr;
    m->has_rowtab.assign(cfg->depth, 0);
    m->max_slices = max_slices;
    m->fork = nullptr;
    hipError_t e = hipMalloc((void **)&m->gelu_tab, (size_t)cfg->depth * 65536);
    if (e == hipSuccess && m->fused_attention && IVIT_OPT_ATTN_ROWTAB) e = hipMalloc((void **)&m->rowtab, (size_t)cfg->depth * 256 * 64 * sizeof(float));
    if (e != hipSuccess) {
        snprintf(h->err, sizeof(h->err), "ivit_vit_create: hipMalloc: %s", hipGetErrorString(e));
        delete m;
        return IVIT_ERR_HIP;
    }
    for (int i = 0; i < cfg->depth; ++i) {
        int rc = ivit_shiftgelu_build_table(h, m->blocks[i].s_gelu, m->blocks[i].dy_gelu, m->gelu_tab + (size_t)i * 65536);
        if (rc != IVIT_OK) { ivit_vit_destroy(m); return rc; }
        // frozen QuantLinear plans: per-channel multipliers and the exactness bounds of the pipelined GEMMs
        const ivit_vit_block &b = m->blocks[i];
        if (m->rowtab && b.exp_aq && 1 - b.exp_dmin <= 64 && fabs(b.dy_qk.m * b.dy_qk.r) < 512.0 && fabs(b.dy_pv.m * b.dy_pv.r) < 512.0) {
            rc = ivit_shiftmax_rowtable(h, b.exp_aq, b.exp_t, b.exp_cls, b.exp_nc, b.exp_tcount, b.exp_dmin, m->rowtab + (size_t)i * 256 * 64);
            if (rc != IVIT_OK) { ivit_vit_destroy(m); return rc; }
            m->has_rowtab[i] = 1;
        }
        const int D = cfg->embed_dim, Hd = cfg->hidden_dim;
        const struct { const int8_t *w; const int32_t *bias; const ivit_dyadic *dy; int N, K; } lin[4] = {
            {b.qkv_w, b.qkv_b, b.qkv_dy, 3 * D, D}, {b.proj_w, b.proj_b, b.proj_dy, D, D},
            {b.fc1_w, b.fc1_b, b.fc1_dy, Hd, D}, {b.fc2_w, b.fc2_b, b.fc2_dy, D, Hd}};
        for (int k = 0; k < 4; ++k) {
            ivit_linear_plan pl = nullptr;
            rc = ivit_linear_plan_create(h, lin[k].w, lin[k].bias, lin[k].dy, lin[k].N, lin[k].K, &pl);
            if (rc != IVIT_OK) { ivit_vit_destroy(m); return rc; }
            // the qkv layer of a D = 384, dh = 64 model also runs on gemm_ws_qkv_kernel (weights in its fragment order)
            if (k == 0 && IVIT_OPT_LN_QKV && D == WS_K && D / cfg->num_heads == 64) (void)ivit_linear_plan_prepare_ws(h, pl);
            if (k == 1 && IVIT_OPT_PROJ_WS && D == WS_K) (void)ivit_linear_plan_prepare_ws(h, pl);      // attn.proj + residual on the same kernel
            m->plans.push_back(pl);
        }
        ivit_mlp_plan mp = nullptr;
        if (D == MLP_C && Hd == MLP_HD && ivit_mlp_plan_create(h, m->plans[4 * i + 2], m->plans[4 * i + 3], &mp) != IVIT_OK) mp = nullptr;
        m->mlp_plans.push_back(mp);
    }
    if (max_slices > 1) {
        bool ok = hipEventCreateWithFlags(&m->fork, hipEventDisableTiming) == hipSuccess;
        for (int i = 0; ok && i < max_slices; ++i) {
            // each resource is owned by `m` as soon as it exists, so the destroy on the error path releases it
            hipStream_t st = nullptr;
            hipEvent_t ev = nullptr;
            ivit_handle sh = nullptr;
            ok = hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess;
            if (ok) m->streams.push_back(st);
            ok = ok && hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess;
            if (ok) m->done.push_back(ev);
            ok = ok && ivit_create(&sh, h->device, st) == IVIT_OK;
            if (ok) m->slice_h.push_back(sh);
        }
        if (!ok) {
            snprintf(h->err, sizeof(h->err), "ivit_vit_create: stream/event creation failed");
            ivit_vit_destroy(m);
            return IVIT_ERR_HIP;
        }
    }
    *out = m;
    return IVIT_OK;
}

int ivit_vit_destroy(ivit_vit m) {
    if (!m) return IVIT_ERR_INVALID;
    for (auto sh : m->slice_h) ivit_destroy(sh);
    for (auto ev : m->done) (void)hipEventDestroy(ev);
    for (auto st : m->streams) (void)hipStreamDestroy(st);
    if (m->fork) (void)hipEventDestroy(m->fork);
    if (m->gelu_tab) (void)hipFree(m->gelu_tab);
    if (m->rowtab) (void)hipFree(m->rowtab);
    for (auto mp : m->mlp_plans) if (mp) (void)ivit_mlp_plan_destroy(mp);
    for (auto pl : m->plans) (void)ivit_linear_plan_destroy(pl);
    delete m;
    return IVIT_OK;
}

int ivit_vit_workspace_bytes(ivit_vit m, int batch, int nslices, size_t *bytes) {
    if (!m) return IVIT_ERR_INVALID;
    REQUIRE(m->h, bytes && batch > 0 && nslices >= 1 && nslices <= m->max_slices && nslices <= batch, "bad arguments");
    *bytes = slice_layout(m, max_slice(batch, nslices)).total * (size_t)nslices;
    return IVIT_OK;
}

int ivit_vit_workspace_init(ivit_vit m, void *workspace, size_t bytes, int batch, int nslices) {
    if (!m) return IVIT_ERR_INVALID;
    size_t need = 0;
    int rc = ivit_vit_workspace_bytes(m, batch, nslices, &need);
    if (rc != IVIT_OK) return rc;
    REQUIRE(m->h, workspace && bytes >= need, "workspace too small");
    REQUIRE(m->h, ((uintptr_t)workspace & 255) == 0, "workspace must be 256-byte aligned");
    const SliceLayout L = slice_layout(m, max_slice(batch, nslices));
    for (int i = 0; i < nslices; ++i) {
        char *ws = (char *)workspace + L.total * (size_t)i;
        const size_t vt_bytes = (size_t)max_slice(batch, nslices) * m->cfg.num_heads * (m->cfg.embed_dim / m->cfg.num_heads) * m->ld;
        if (hipMemsetAsync(ws + L.vt, 0, vt_bytes, m->h->stream) != hipSuccess) return IVIT_ERR_HIP;
        if (!m->fused_attention) {
            const size_t pb = (size_t)max_slice(batch, nslices) * m->cfg.num_heads * m->T * m->ld * 2;
            if (hipMemsetAsync(ws + L.p16, 0, pb, m->h->stream) != hipSuccess) return IVIT_ERR_HIP;
        }
    }
    return IVIT_OK;
}

int ivit_vit_forward(ivit_vit m, const int8_t *images, int batch, int nslices, void *workspace, size_t bytes,
                     int32_t *logits) {
    if (!m) return IVIT_ERR_INVALID;
    ivit_handle h = m->h;
    size_t need = 0;
    int rc = ivit_vit_workspace_bytes(m, batch, nslices, &need);
    if (rc != IVIT_OK) return rc;
    REQUIRE(h, images && logits && workspace && bytes >= need, "bad arguments / workspace too small");
    REQUIRE(h, ((uintptr_t)workspace & 255) == 0, "workspace must be 256-byte aligned");
    const size_t img_bytes = (size_t)m->cfg.in_chans * m->cfg.img_size * m->cfg.img_size;
    const size_t stride = slice_layout(m, max_slice(batch, nslices)).total;
    if (nslices == 1) return run_slice(m, h, images, batch, batch, (char *)workspace, logits);
    if (hipEventRecord(m->fork, h->stream) != hipSuccess) return IVIT_ERR_HIP;
    for (int i = 0; i < nslices; ++i) {
        const int b0 = slice_begin(batch, nslices, i), b1 = slice_begin(batch, nslices, i + 1);
        if (hipStreamWaitEvent(m->streams[i], m->fork, 0) != hipSuccess) return IVIT_ERR_HIP;
        m->slice_h[i]->cu_share = IVIT_OPT_SLICE_CU_SHARE ? std::max(1, persistent_cus(h) / nslices) : 0;      // a share of the caller's own share
        rc = run_slice(m, m->slice_h[i], images + (size_t)b0 * img_bytes, b1 - b0, max_slice(batch, nslices),
                       (char *)workspace + stride * (size_t)i,
                       logits + (size_t)b0 * m->cfg.num_classes);
        if (rc != IVIT_OK) return rc;
        if (hipEventRecord(m->done[i], m->streams[i]) != hipSuccess) return IVIT_ERR_HIP;
    }
    for (int i = 0; i < nslices; ++i)
        if (hipStreamWaitEvent(h->stream, m->done[i], 0) != hipSuccess) return IVIT_ERR_HIP;
    return IVIT_OK;
}

int ivit_vit_graph_create(ivit_vit m, const int8_t *images, int batch, int nslices, void *workspace, size_t bytes,
                          int32_t *logits, ivit_graph *out) {
    if (!m) return IVIT_ERR_INVALID;
    ivit_handle h = m->h;
    REQUIRE(h, out, "null argument");
    REQUIRE(h, h->stream != nullptr, "graph capture needs a non-default stream on the handle");
    hipError_t e = hipStreamBeginCapture(h->stream, hipStreamCaptureModeRelaxed);
    if (e != hipSuccess) { snprintf(h->err, sizeof(h->err), "begin capture: %s", hipGetErrorString(e)); return IVIT_ERR_HIP; }
    int rc = ivit_vit_forward(m, images, batch, nslices, workspace, bytes, logits);
    hipGraph_t graph = nullptr;
    e = hipStreamEndCapture(h->stream, &graph);
    if (rc != IVIT_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    if (e != hipSuccess || !graph) { snprintf(h->err, sizeof(h->err), "end capture: %s", hipGetErrorString(e)); return IVIT_ERR_HIP; }
    hipGraphExec_t exec = nullptr;
    e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    if (e != hipSuccess) { (void)hipGraphDestroy(graph); snprintf(h->err, sizeof(h->err), "instantiate: %s", hipGetErrorString(e)); return IVIT_ERR_HIP; }
    ivit_graph_s *g = new (std::nothrow) ivit_graph_s();
    if (!g) { (void)hipGraphExecDestroy(exec); (void)hipGraphDestroy(graph); return IVIT_ERR_HIP; }
    g->h = h; g->graph = graph; g->exec = exec;
    *out = g;
    return IVIT_OK;
}

int ivit_graph_launch(ivit_graph g) {
    if (!g) return IVIT_ERR_INVALID;
    hipError_t e = hipGraphLaunch(g->exec, g->h->stream);
    if (e != hipSuccess) { snprintf(g->h->err, sizeof(g->h->err), "graph launch: %s", hipGetErrorString(e)); return IVIT_ERR_HIP; }
    return IVIT_OK;
}

int ivit_graph_destroy(ivit_graph g) {
    if (!g) return IVIT_ERR_INVALID;
    (void)hipGraphExecDestroy(g->exec);
    (void)hipGraphDestroy(g->graph);
    delete g;
    return IVIT_OK;
}

}  // extern "C"

// =====================================================================================================
// Swin runner
struct ivit_swin_s {
    ivit_handle h;
    ivit_swin_config cfg;
    ivit_swin_params prm;
    std::vector<ivit_swin_block> blocks;
    std::vector<ivit_swin_merge> merges;
    int grid, nblocks;
    ivit_dyadic dy_qact1_host;        // host copy of prm.dy_qact1[0]
    bool fused_mlp;                   // stage-0 Mlp in one kernel (ivit_mlp_fused)
    std::vector<ivit_linear_plan> mlp_lin;   // per block: fc1, fc2 plans of the C = 384 stage (null elsewhere)
    std::vector<ivit_mlp_plan> mlp_plans;    // per block: fused Mlp plan (C = 384, hidden 1536) or null
    std::vector<ivit_linear_plan> lin_plans; // per block: qkv, proj, fc1, fc2 plans where C % 384 == 0 (IVIT_OPT_SWIN_PLANS), else null
    int8_t *gelu_tab;                 // [nblocks][65536]
    int max_slices;
    std::vector<ivit_handle> slice_h;
    std::vector<hipStream_t> streams;
    std::vector<hipEvent_t> done;
    hipEvent_t fork;
};

#ifndef IVIT_OPT_MERGE_LN
#define IVIT_OPT_MERGE_LN 1            // A/B: PatchMerging's gather folded into its LayerNorm (ivit_patch_merge_layernorm_requant)
#endif
#ifndef IVIT_OPT_SWIN_WS
#define IVIT_OPT_SWIN_WS 1             // A/B: the C = 384 stage's qkv (+ norm1) and proj layers on gemm_ws_qkv_kernel
#endif
#ifndef IVIT_OPT_MERGE_W16
#define IVIT_OPT_MERGE_W16 1            // A/B: PatchMerging's reduction stores its 8-bit QuantAct as int16 (no ivit_widen_i8_i16 pass)
#endif
#ifndef IVIT_OPT_SWIN_LN_MLP
#define IVIT_OPT_SWIN_LN_MLP 0
#endif
#ifndef IVIT_OPT_SWIN_PLANS
#define IVIT_OPT_SWIN_PLANS 0          // A/B: the C = 384 / 768 stages' QuantLinear layers on the planned (persistent) kernels
#endif

namespace {

struct SwinLayout { size_t patches, a8, xa, xb, xc, zf, qkv, ctx, h8, g8, pool, total; };

SwinLayout swin_layout(const ivit_swin_s *m, int B) {
    const ivit_swin_config &c = m->cfg;
    const size_t M0 = (size_t)B * m->grid * m->grid, E = c.embed_dim;
    SwinLayout L;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o = al256(o + bytes + 64); return at; };
    L.patches = take(M0 * c.in_chans * c.patch_size * c.patch_size);
    L.a8 = take(M0 * E);
    L.xa = take(M0 * E * 2);
    L.xb = take(M0 * E * 2);
    L.xc = take(M0 * E * 2);
    L.zf = take(M0 * E * 4);
    L.qkv = take(M0 * 3 * E);
    L.ctx = take(M0 * E);
    L.h8 = take(M0 * c.mlp_ratio * E);
    L.g8 = take(M0 * c.mlp_ratio * E);
    L.pool = take((size_t)B * (E << (c.num_layers - 1)));
    L.total = o;
    return L;
}

int swin_ln(const ivit_swin_s *m, ivit_handle h, const int16_t *x, long long M, int C, float s_in, const ivit_ln_params &n,
            int L, bool token_order, int8_t *out8) {
    if (token_order) return ivit_layernorm_tokenorder_requant(h, x, M, C, s_in, n.bias_int, n.sc, n.dy, L, out8);
    return ivit_layernorm_requant(h, x, M, C, C, s_in, n.bias_int, n.sc, n.dy, out8);
}

int swin_run_slice(const ivit_swin_s *m, ivit_handle h, const int8_t *images, int B, char *ws, int32_t *logits) {
    const ivit_swin_config &c = m->cfg;
    const ivit_swin_params &P = m->prm;
    const SwinLayout Lw = swin_layout(m, B);
    int8_t *patches = (int8_t *)(ws + Lw.patches), *a8 = (int8_t *)(ws + Lw.a8), *qkv = (int8_t *)(ws + Lw.qkv),
           *ctx = (int8_t *)(ws + Lw.ctx), *h8 = (int8_t *)(ws + Lw.h8), *g8 = (int8_t *)(ws + Lw.g8),
           *pool = (int8_t *)(ws + Lw.pool);
    int16_t *x = (int16_t *)(ws + Lw.xa), *y = (int16_t *)(ws + Lw.xb), *t16 = (int16_t *)(ws + Lw.xc);
    float *zf = (float *)(ws + Lw.zf);
    const int E = c.embed_dim;
    int res = m->grid, L = res * res;
    long long M = (long long)B * L;
    const int Kp = c.in_chans * c.patch_size * c.patch_size;
    int rc;
#define RUN(call) do { rc = (call); if (rc != IVIT_OK) { if (h != m->h) snprintf(m->h->err, sizeof(m->h->err), "%s", h->err); return rc; } } while (0)
    // PatchEmbed: conv -> qact_before_norm(8) -> norm (token-order sums) -> qact(16) -> qact1(16)
    RUN(ivit_im2col_patch(h, images, B, c.in_chans, c.img_size, c.img_size, c.patch_size, patches));
    RUN(ivit_linear_i8_requant(h, patches, P.pe.w, P.pe.b, P.pe.dy, 8, a8, (int)M, E, Kp));
    RUN(ivit_patch_norm_tokenorder(h, a8, M, E, P.s_bn, P.pn.bias_int, P.pn.sc, P.pn.dy, m->dy_qact1_host, L, x));
    int bi = 0;
    for (int li = 0; li < c.num_layers; ++li) {
        const int C = E << li, heads = c.num_heads[li];
        for (int bj = 0; bj < c.depths[li]; ++bj, ++bi) {
            const ivit_swin_block &b = m->blocks[bi];
            const int shift = (bj % 2 == 0 || res <= c.window_size) ? 0 : c.window_size / 2;
            const ivit_linear_plan *lp = m->lin_plans.empty() ? nullptr : &m->lin_plans[4 * bi];
            // norm1 inside the qkv launch where that layer runs on gemm_ws_qkv_kernel (C = 384, activations in natural token order)
            rc = (IVIT_OPT_SWIN_WS && lp && lp[0] && li != 0) ? ivit_layernorm_linear_i8_requant_planned(h, lp[0], x, b.s_in, b.n1.bias_int, b.n1.sc, b.n1.dy, qkv, (int)M)
                                                              : IVIT_ERR_UNSUPPORTED;
            if (rc == IVIT_ERR_UNSUPPORTED) {
                RUN(swin_ln(m, h, x, M, C, b.s_in, b.n1, L, li == 0, a8));
                if (lp && lp[0]) RUN(ivit_linear_i8_requant_planned(h, lp[0], a8, 8, qkv, (int)M));
                else RUN(ivit_linear_i8_requant(h, a8, b.qkv.w, b.qkv.b, b.qkv.dy, 8, qkv, (int)M, 3 * C, C));
            } else {
                RUN(rc);
            }
            if (b.exp_aq)
                RUN(ivit_window_attention_fused_lut(h, qkv, b.dy_qk, b.dy_a, b.relb, b.s_softmax, b.exp_aq, b.exp_t, b.exp_cls,
                                                    b.exp_nc, b.exp_tcount, b.exp_dmin, b.dy_pv, ctx, B, res, c.window_size,
                                                    shift, heads, C / heads));
            else
                RUN(ivit_window_attention_fused(h, qkv, b.dy_qk, b.dy_a, b.relb, b.s_softmax, b.dy_pv, ctx, B, res,
                                                c.window_size, shift, heads, C / heads));
            if (lp && lp[1]) RUN(ivit_linear_i8_requant_residual_planned(h, lp[1], ctx, b.res1_main, b.res1_res, x, y, (int)M));
            else RUN(ivit_linear_i8_requant_residual(h, ctx, b.proj.w, b.proj.b, b.proj.dy, b.res1_main, b.res1_res, x, y, (int)M, C, C));
            { int16_t *t = x; x = y; y = t; }
            const bool mlp384 = m->mlp_plans[bi] && fabs(b.res2_main.m * b.res2_main.r) < RQ_FAST_CLIM &&
                                fabs(b.res2_res.m * b.res2_res.r) < RQ_FAST_CLIM;
            // C = 384 stage: norm2 can ride in the head of the fused Mlp's launch (natural token order) — measured SLOWER here (Swin-T b256,
            // two slices: 4.76 against 4.70 ms same-box; the LayerNorm launch of one slice overlaps the other slice's kernels) and off
            rc = (IVIT_OPT_SWIN_LN_MLP && mlp384 && li != 0)
                     ? ivit_layernorm_mlp_fused_planned(h, m->mlp_plans[bi], x, b.s_mid, b.n2.bias_int, b.n2.sc, b.n2.dy, a8,
                                                        m->gelu_tab + (size_t)bi * 65536, b.res2_main, b.res2_res, y, M)
                     : IVIT_ERR_UNSUPPORTED;
            const bool ln_mlp = rc != IVIT_ERR_UNSUPPORTED;
            if (ln_mlp) RUN(rc);
            else RUN(swin_ln(m, h, x, M, C, b.s_mid, b.n2, L, li == 0, a8));
            if (ln_mlp) {
            } else if (C == 96 && c.mlp_ratio == 4 && m->fused_mlp) {     // narrow stage: hidden tensor stays in LDS
                RUN(ivit_mlp_fused(h, a8, b.fc1.w, b.fc1.b, b.fc1.dy, m->gelu_tab + (size_t)bi * 65536, b.fc2.w, b.fc2.b,
                                   b.fc2.dy, b.res2_main, b.res2_res, x, y, M, C, 4 * C));
            } else if (mlp384) {                                    // C = 384 stage: weights streamed, hidden tile in LDS
                RUN(ivit_mlp_fused_planned(h, m->mlp_plans[bi], a8, m->gelu_tab + (size_t)bi * 65536, b.res2_main, b.res2_res, x, y, M));
            } else {
                if (lp && lp[2]) RUN(ivit_linear_i8_requant_planned(h, lp[2], a8, 8, h8, (int)M));
                else RUN(ivit_linear_i8_requant(h, a8, b.fc1.w, b.fc1.b, b.fc1.dy, 8, h8, (int)M, c.mlp_ratio * C, C));
                RUN(ivit_shiftgelu_requant_lut(h, h8, M, c.mlp_ratio * C, m->gelu_tab + (size_t)bi * 65536, g8));
                if (lp && lp[3]) RUN(ivit_linear_i8_requant_residual_planned(h, lp[3], g8, b.res2_main, b.res2_res, x, y, (int)M));
                else RUN(ivit_linear_i8_requant_residual(h, g8, b.fc2.w, b.fc2.b, b.fc2.dy, b.res2_main, b.res2_res, x, y, (int)M, C, c.mlp_ratio * C));
            }
            { int16_t *t = x; x = y; y = t; }
        }
        if (li < c.num_layers - 1) {     // PatchMerging: gather -> LN(4C) -> qact1(8) -> reduction -> qact2(8)
            const ivit_swin_merge &g = m->merges[li];
            // the 2 x 2 gather rides in the LayerNorm's loads (round 6: as a pass of its own it was 32 us per merge at Swin-T b256)
            rc = IVIT_OPT_MERGE_LN ? ivit_patch_merge_layernorm_requant(h, x, B, res, C, g.s_in, g.n.bias_int, g.n.sc, g.n.dy, a8) : IVIT_ERR_UNSUPPORTED;
            const bool merged = rc == IVIT_OK;
            if (!merged && rc != IVIT_ERR_UNSUPPORTED) RUN(rc);
            if (!merged) RUN(ivit_patch_merge_gather(h, x, 16, B, res, C, t16));
            res /= 2;
            L = res * res;
            M = (long long)B * L;
            if (!merged) RUN(swin_ln(m, h, t16, M, 4 * C, g.s_in, g.n, L, false, a8));
            // reduction -> qact2(8), stored as the 16-bit stream the next stage reads (round 6: the widening pass was 16 us per merge)
            rc = IVIT_OPT_MERGE_W16 ? ivit_linear_i8_requant8_store16(h, a8, g.red.w, nullptr, g.red.dy, x, (int)M, 2 * C, 4 * C) : IVIT_ERR_UNSUPPORTED;
            if (rc == IVIT_ERR_UNSUPPORTED) {
                RUN(ivit_linear_i8_requant(h, a8, g.red.w, nullptr, g.red.dy, 8, ctx, (int)M, 2 * C, 4 * C));
                RUN(ivit_widen_i8_i16(h, ctx, x, M * 2 * C));
            } else {
                RUN(rc);
            }
        }
    }
    const int C = E << (c.num_layers - 1);
    RUN(swin_ln(m, h, x, M, C, P.s_norm_in, P.n, L, false, a8));
    RUN(ivit_avgpool_requant(h, a8, B, L, C, P.dy_pool, pool));
    RUN(ivit_linear_i8(h, pool, P.head_w, P.head_b, logits, B, c.num_classes, C));
#undef RUN
    return IVIT_OK;
}

}  // namespace

extern "C" {

int ivit_swin_destroy(ivit_swin m) {
    if (!m) return IVIT_ERR_INVALID;
    for (auto sh : m->slice_h) ivit_destroy(sh);
    for (auto ev : m->done) (void)hipEventDestroy(ev);
    for (auto st : m->streams) (void)hipStreamDestroy(st);
    if (m->fork) (void)hipEventDestroy(m->fork);
    if (m->gelu_tab) (void)hipFree(m->gelu_tab);
    for (auto mp : m->mlp_plans) if (mp) (void)ivit_mlp_plan_destroy(mp);
    for (auto pl : m->mlp_lin) if (pl) (void)ivit_linear_plan_destroy(pl);
    for (auto pl : m->lin_plans) if (pl) (void)ivit_linear_plan_destroy(pl);
    delete m;
    return IVIT_OK;
}

int ivit_swin_create(ivit_handle h, const ivit_swin_config *cfg, const ivit_swin_params *params, int max_slices,
                     ivit_swin *out) {
    CHECK_H(h);
    REQUIRE(h, cfg && params && out && params->blocks_host && params->dy_qact1, "null argument");
    REQUIRE(h, cfg->num_layers >= 1 && cfg->num_layers <= 4 && cfg->embed_dim > 0 && cfg->patch_size > 0 &&
                   cfg->img_size % cfg->patch_size == 0 && cfg->mlp_ratio > 0 && cfg->num_classes > 0,
            "bad model configuration");
    REQUIRE(h, cfg->num_layers == 1 || params->merges_host, "merges_host missing");
    REQUIRE(h, max_slices >= 1 && max_slices <= 16, "max_slices must be in [1, 16]");
    int nb = 0;
    for (int li = 0; li < cfg->num_layers; ++li) {
        if (cfg->window_size != 7 || cfg->num_heads[li] <= 0 || ((cfg->embed_dim << li) / cfg->num_heads[li]) != 32) {
            snprintf(h->err, sizeof(h->err), "ivit_swin_create: built for window 7 and head dim 32");
            return IVIT_ERR_UNSUPPORTED;
        }
        nb += cfg->depths[li];
    }
    const int grid = cfg->img_size / cfg->patch_size;
    REQUIRE(h, (grid >> (cfg->num_layers - 1)) % 7 == 0 && grid % (7 << (cfg->num_layers - 1)) == 0,
            "every stage resolution must be a multiple of the window");
    ivit_swin_s *m = new (std::nothrow) ivit_swin_s();
    if (!m) return IVIT_ERR_HIP;
    m->h = h; m->cfg = *cfg; m->prm = *params;
    m->blocks.assign(params->blocks_host, params->blocks_host + nb);
    if (cfg->num_layers > 1) m->merges.assign(params->merges_host, params->merges_host + cfg->num_layers - 1);
    m->grid = grid; m->nblocks = nb; m->gelu_tab = nullptr; m->max_slices = max_slices; m->fork = nullptr;
    m->fused_mlp = true;
    if (hipMemcpy(&m->dy_qact1_host, params->dy_qact1, sizeof(ivit_dyadic), hipMemcpyDeviceToHost) != hipSuccess) {
        snprintf(h->err, sizeof(h->err), "ivit_swin_create: cannot read dy_qact1");
        delete m;
        return IVIT_ERR_HIP;
    }
    if (hipMalloc((void **)&m->gelu_tab, (size_t)nb * 65536) != hipSuccess) {
        snprintf(h->err, sizeof(h->err), "ivit_swin_create: hipMalloc failed");
        delete m;
        return IVIT_ERR_HIP;
    }
    for (int i = 0; i < nb; ++i) {
        int rc = ivit_shiftgelu_build_table(h, m->blocks[i].s_gelu, m->blocks[i].dy_gelu, m->gelu_tab + (size_t)i * 65536);
        if (rc != IVIT_OK) { ivit_swin_destroy(m); return rc; }
    }
    {   // fused Mlp plans for the C = 384 stage (hidden 1536)
        int bi = 0;
        for (int li = 0; li < cfg->num_layers; ++li)
            for (int bj = 0; bj < cfg->depths[li]; ++bj, ++bi) {
                const int C = cfg->embed_dim << li;
                ivit_linear_plan p1 = nullptr, p2 = nullptr;
                ivit_mlp_plan mp = nullptr;
                const ivit_swin_block &b = m->blocks[bi];
                if (C == MLP_C && cfg->mlp_ratio * C == MLP_HD &&
                    ivit_linear_plan_create(h, b.fc1.w, b.fc1.b, b.fc1.dy, MLP_HD, MLP_C, &p1) == IVIT_OK &&
                    ivit_linear_plan_create(h, b.fc2.w, b.fc2.b, b.fc2.dy, MLP_C, MLP_HD, &p2) == IVIT_OK) {
                    if (ivit_mlp_plan_create(h, p1, p2, &mp) != IVIT_OK) mp = nullptr;
                }
                m->mlp_lin.push_back(p1);
                m->mlp_lin.push_back(p2);
                m->mlp_plans.push_back(mp);
                if (IVIT_OPT_SWIN_PLANS || IVIT_OPT_SWIN_WS) {
                    ivit_linear_plan q[4] = {nullptr, nullptr, nullptr, nullptr};
                    if (!IVIT_OPT_SWIN_PLANS) {
                        // round 6: the C = 384 stage's qkv and proj layers on gemm_ws_qkv_kernel (prepared plans), norm1 inside the qkv launch
                        if (C == WS_K) {
                            if (ivit_linear_plan_create(h, b.qkv.w, b.qkv.b, b.qkv.dy, 3 * C, C, &q[0]) != IVIT_OK) q[0] = nullptr;
                            if (q[0] && ivit_linear_plan_prepare_ws(h, q[0]) != IVIT_OK) { (void)ivit_linear_plan_destroy(q[0]); q[0] = nullptr; }
                            if (ivit_linear_plan_create(h, b.proj.w, b.proj.b, b.proj.dy, C, C, &q[1]) != IVIT_OK) q[1] = nullptr;
                            if (q[1] && ivit_linear_plan_prepare_ws(h, q[1]) != IVIT_OK) { (void)ivit_linear_plan_destroy(q[1]); q[1] = nullptr; }
                        }
                    } else if (C % 384 == 0) {
                        if (ivit_linear_plan_create(h, b.qkv.w, b.qkv.b, b.qkv.dy, 3 * C, C, &q[0]) != IVIT_OK) q[0] = nullptr;
                        if (ivit_linear_plan_create(h, b.proj.w, b.proj.b, b.proj.dy, C, C, &q[1]) != IVIT_OK) q[1] = nullptr;
                        if (!mp) {
                            if (ivit_linear_plan_create(h, b.fc1.w, b.fc1.b, b.fc1.dy, cfg->mlp_ratio * C, C, &q[2]) != IVIT_OK) q[2] = nullptr;
                            if (ivit_linear_plan_create(h, b.fc2.w, b.fc2.b, b.fc2.dy, C, cfg->mlp_ratio * C, &q[3]) != IVIT_OK) q[3] = nullptr;
                        }
                    }
                    for (int k = 0; k < 4; ++k) m->lin_plans.push_back(q[k]);
                }
            }
    }
    if (max_slices > 1) {
        bool ok = hipEventCreateWithFlags(&m->fork, hipEventDisableTiming) == hipSuccess;
        for (int i = 0; ok && i < max_slices; ++i) {
            // each resource is owned by `m` as soon as it exists, so the destroy on the error path releases it
            hipStream_t st = nullptr;
            hipEvent_t ev = nullptr;
            ivit_handle sh = nullptr;
            ok = hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess;
            if (ok) m->streams.push_back(st);
            ok = ok && hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess;
            if (ok) m->done.push_back(ev);
            ok = ok && ivit_create(&sh, h->device, st) == IVIT_OK;
            if (ok) m->slice_h.push_back(sh);
        }
        if (!ok) {
            snprintf(h->err, sizeof(h->err), "ivit_swin_create: stream/event creation failed");
            ivit_swin_destroy(m);
            return IVIT_ERR_HIP;
        }
    }
    *out = m;
    return IVIT_OK;
}

int ivit_swin_workspace_bytes(ivit_swin m, int batch, int nslices, size_t *bytes) {
    if (!m) return IVIT_ERR_INVALID;
    REQUIRE(m->h, bytes && batch > 0 && nslices >= 1 && nslices <= m->max_slices && nslices <= batch, "bad arguments");
    *bytes = swin_layout(m, max_slice(batch, nslices)).total * (size_t)nslices;
    return IVIT_OK;
}

int ivit_swin_forward(ivit_swin m, const int8_t *images, int batch, int nslices, void *workspace, size_t bytes,
                      int32_t *logits) {
    if (!m) return IVIT_ERR_INVALID;
    ivit_handle h = m->h;
    size_t need = 0;
    int rc = ivit_swin_workspace_bytes(m, batch, nslices, &need);
    if (rc != IVIT_OK) return rc;
    REQUIRE(h, images && logits && workspace && bytes >= need, "bad arguments / workspace too small");
    REQUIRE(h, ((uintptr_t)workspace & 255) == 0, "workspace must be 256-byte aligned");
    const size_t img_bytes = (size_t)m->cfg.in_chans * m->cfg.img_size * m->cfg.img_size;
    const size_t stride = swin_layout(m, max_slice(batch, nslices)).total;
    if (nslices == 1) return swin_run_slice(m, h, images, batch, (char *)workspace, logits);
    if (hipEventRecord(m->fork, h->stream) != hipSuccess) return IVIT_ERR_HIP;
    for (int i = 0; i < nslices; ++i) {
        const int b0 = slice_begin(batch, nslices, i), b1 = slice_begin(batch, nslices, i + 1);
        if (hipStreamWaitEvent(m->streams[i], m->fork, 0) != hipSuccess) return IVIT_ERR_HIP;
        m->slice_h[i]->cu_share = IVIT_OPT_SLICE_CU_SHARE ? std::max(1, persistent_cus(h) / nslices) : 0;      // a share of the caller's own share
        rc = swin_run_slice(m, m->slice_h[i], images + (size_t)b0 * img_bytes, b1 - b0, (char *)workspace + stride * (size_t)i,
                            logits + (size_t)b0 * m->cfg.num_classes);
        if (rc != IVIT_OK) return rc;
        if (hipEventRecord(m->done[i], m->streams[i]) != hipSuccess) return IVIT_ERR_HIP;
    }
    for (int i = 0; i < nslices; ++i)
        if (hipStreamWaitEvent(h->stream, m->done[i], 0) != hipSuccess) return IVIT_ERR_HIP;
    return IVIT_OK;
}

int ivit_swin_graph_create(ivit_swin m, const int8_t *images, int batch, int nslices, void *workspace, size_t bytes,
                           int32_t *logits, ivit_graph *out) {
    if (!m) return IVIT_ERR_INVALID;
    ivit_handle h = m->h;
    REQUIRE(h, out, "null argument");
    REQUIRE(h, h->stream != nullptr, "graph capture needs a non-default stream on the handle");
    hipError_t e = hipStreamBeginCapture(h->stream, hipStreamCaptureModeRelaxed);
    if (e != hipSuccess) { snprintf(h->err, sizeof(h->err), "begin capture: %s", hipGetErrorString(e)); return IVIT_ERR_HIP; }
    int rc = ivit_swin_forward(m, images, batch, nslices, workspace, bytes, logits);
    hipGraph_t graph = nullptr;
    e = hipStreamEndCapture(h->stream, &graph);
    if (rc != IVIT_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    if (e != hipSuccess || !graph) { snprintf(h->err, sizeof(h->err), "end capture: %s", hipGetErrorString(e)); return IVIT_ERR_HIP; }
    hipGraphExec_t exec = nullptr;
    e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    if (e != hipSuccess) { (void)hipGraphDestroy(graph); snprintf(h->err, sizeof(h->err), "instantiate: %s", hipGetErrorString(e)); return IVIT_ERR_HIP; }
    ivit_graph_s *g = new (std::nothrow) ivit_graph_s();
    if (!g) { (void)hipGraphExecDestroy(exec); (void)hipGraphDestroy(graph); return IVIT_ERR_HIP; }
    g->h = h; g->graph = graph; g->exec = exec;
    *out = g;
    return IVIT_OK;
}

}  // extern "C"
