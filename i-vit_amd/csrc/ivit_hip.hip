// ivit_hip.hip — C-ABI (include/ivit.h) over the gfx950 kernels.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -shared -fPIC
#include <hip/hip_runtime.h>
#include <algorithm>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <new>
#include <dlfcn.h>
#include <mutex>
#include <atomic>
#include "ivit_device.h"
#include "ivit_elementwise.h"
#include "ivit_layernorm.h"
#include "ivit_gemm.h"
#include "ivit_attention.h"
#include "ivit_gemm2.h"
#include "ivit_gemm3.h"
#include "ivit_gemm_wreg.h"
#include "ivit_swin.h"
#include "ivit_mlp.h"
#include "ivit_mlp_rs.h"
#include "ivit_swin_mlp_rs.h"
#include "ivit_gemm_ws.h"

#define IVIT_MAX_DEVICES 64     // per-device caches of launch attributes (larger ordinals simply do not cache)
struct ivit_ctx {
    int device;
    hipStream_t stream;
    int num_cu;
    // CUs the one-workgroup-per-CU kernels (gemm_ws_qkv_kernel, mlp384rs_kernel / mlp384_kernel) size their grids for; 0 = num_cu.  The
    // whole-model runners give the handle of a slice stream its share of the device (num_cu / slices): a slice's launch then has the
    // per-workgroup geometry of the unsliced one (DeiT-S b256: one image per workgroup) and two slices' kernels run side by side on
    // disjoint CUs instead of as two half-filled grids of 256 (profiles/README.md, round 6, last session)
    int cu_share;
    char err[256];
};
static inline int persistent_cus(const ivit_ctx *h) { return h->cu_share > 0 ? h->cu_share : h->num_cu; }

// every entry point binds the calling thread to the handle's device for the duration of the call and puts the caller's
// current device back on the way out: one process may drive several GPUs through several handles, and a torch (or any
// other HIP) user of the same thread keeps allocating / launching where it was
struct ivit_device_guard {
    int prev = -1;
    bool restore = false;
    bool enter(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev == dev) return true;
        if (hipSetDevice(dev) != hipSuccess) return false;
        restore = prev >= 0;
        return true;
    }
    ~ivit_device_guard() { if (restore) (void)hipSetDevice(prev); }
};
#define CHECK_H(h) if (!(h)) return IVIT_ERR_INVALID; ivit_device_guard ivit_dev_guard_; if (!ivit_dev_guard_.enter((h)->device)) return IVIT_ERR_HIP
#define REQUIRE(h, cond, msg) do { if (!(cond)) { snprintf((h)->err, sizeof((h)->err), "%s: %s", __func__, msg); return IVIT_ERR_INVALID; } } while (0)
#define LAUNCH_CHECK(h) do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) { snprintf((h)->err, sizeof((h)->err), "%s: %s", __func__, hipGetErrorString(e_)); return IVIT_ERR_HIP; } } while (0)

extern "C" {

int ivit_version(void) { return IVIT_VERSION; }

const char *ivit_status_string(int s) {
    switch (s) {
        case IVIT_OK: return "ok";
        case IVIT_ERR_INVALID: return "invalid argument";
        case IVIT_ERR_HIP: return "HIP runtime error";
        case IVIT_ERR_UNSUPPORTED: return "unsupported shape";
        case IVIT_ERR_NO_DEVICE: return "no HIP device";
        default: return "unknown status";
    }
}

int ivit_create(ivit_handle *out, int device, void *hip_stream) {
    if (!out) return IVIT_ERR_INVALID;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return IVIT_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return IVIT_ERR_HIP;
    ivit_ctx *c = new (std::nothrow) ivit_ctx();
    if (!c) return IVIT_ERR_HIP;
    c->device = device;
    c->stream = (hipStream_t)hip_stream;
    c->err[0] = 0;
    c->cu_share = 0;
    hipDeviceProp_t prop;
    c->num_cu = (hipGetDeviceProperties(&prop, device) == hipSuccess) ? prop.multiProcessorCount : 256;
    *out = c;
    return IVIT_OK;
}

int ivit_destroy(ivit_handle h) {
    CHECK_H(h);
    delete h;
    return IVIT_OK;
}

int ivit_set_stream(ivit_handle h, void *hip_stream) {
    CHECK_H(h);
    h->stream = (hipStream_t)hip_stream;
    return IVIT_OK;
}

int ivit_set_cu_share(ivit_handle h, int cus) {
    CHECK_H(h);
    REQUIRE(h, cus >= 0, "cus must be >= 0 (0 = the whole device)");
    h->cu_share = cus > h->num_cu ? h->num_cu : cus;
    return IVIT_OK;
}

const char *ivit_last_error(ivit_handle h) { return h ? h->err : "null handle"; }

// tuning / ablation switches are read once per process (thread-safe static initialisation at the call site)
// Experiment switches of the dispatch are COMPILE-TIME options (build a scratch library with -DIVIT_OPT_...=n and point
// IVIT_LIB at it): a stray environment variable must not change which kernel a user of the library gets.
#ifndef IVIT_OPT_GEMM3
#define IVIT_OPT_GEMM3 7                // epilogues on the persistent pipelined kernels: 1 requant, 2 qkv scatter, 4 requant + residual
#endif
#ifndef IVIT_OPT_GEMM3_RES_MIN_N
#define IVIT_OPT_GEMM3_RES_MIN_N 512    // residual flavour on the persistent kernel from this output width on
#endif
#ifndef IVIT_OPT_GEMM3_FMA
#define IVIT_OPT_GEMM3_FMA (-1)         // -1: as the plan proves; 0 / 1: force the two-rounding / single-FMA requant (1 only where proven)
#endif
#ifndef G2_DBG
#define G2_DBG 0
#endif
#ifndef IVIT_OPT_GEMM_BM
#define IVIT_OPT_GEMM_BM 0              // 0: tile height by the occupancy estimate; 128 / 256: forced
#endif
#ifndef IVIT_OPT_ATTN_GENERIC
#define IVIT_OPT_ATTN_GENERIC 0         // 1: run-time token count; 2: arithmetic Shiftmax even when tables are given
#endif

static inline int grid_for(ivit_handle h, long long work_items, int per_block) {
    long long g = (work_items + per_block - 1) / per_block;
    long long cap = (long long)h->num_cu * 16;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

int ivit_quantize_input_f32(ivit_handle h, const float *x, float scale, int8_t *q, int64_t n) {
    CHECK_H(h);
    REQUIRE(h, x && q && n >= 0 && scale > 0.f, "bad arguments");
    if (n == 0) return IVIT_OK;
    quantize_input_kernel<<<grid_for(h, n, 1024), 256, 0, h->stream>>>(x, scale, q, n);
    LAUNCH_CHECK(h);
    return IVIT_OK;
}

int ivit_resize_center_crop_u8(ivit_handle h, const uint8_t *hwc, int B, int H0, int W0, int size, int crop,
                               float *workspace, uint8_t *out_hwc) {
    CHECK_H(h);
    REQUIRE(h, hwc && workspace && out_hwc && B > 0 && H0 > 0 && W0 > 0 && size > 0 && crop > 0, "bad arguments");
    // torchvision Resize(int): the SHORTER side becomes `size`, the longer int(size * long / short); CenterCrop offsets
    // int(round((dim - crop) / 2.0))
    int Hr, Wr;
    if (H0 <= W0) { Hr = size; Wr = (int)((long long)size * W0 / H0); }
    else { Wr = size; Hr = (int)((long long)size * H0 / W0); }
    REQUIRE(h, crop <= Hr && crop <= Wr, "crop larger than the resized image");
    const int top = (int)__builtin_rint((Hr - crop) / 2.0), left = (int)__builtin_rint((Wr - crop) / 2.0);
    resize_h_kernel<<<grid_for(h, (long long)B * H0 * crop, 256), 256, 0, h->stream>>>(hwc, B, H0, W0, Wr, left, crop, workspace);
    resize_v_kernel<<<grid_for(h, (long long)B * crop * crop, 256), 256, 0, h->stream>>>(workspace, B, H0, Hr, top, crop, out_hwc);
    LAUNCH_CHECK(h);
    return IVIT_OK;
}

int ivit_normalize_quantize_u8(ivit_handle h, const uint8_t *hwc, int B, int H, int W, const float mean[3],
                               const float std_[3], float scale, int8_t *nchw) {
    CHECK_H(h);
    REQUIRE(h, hwc && nchw && mean && std_ && B > 0 && H > 0 && W > 0 && scale > 0.f, "bad arguments");
    REQUIRE(h, std_[0] != 0.f && std_[1] != 0.f && std_[2] != 0.f, "zero std");
    normalize_quantize_u8_kernel<<<grid_for(h, (long long)B * H * W, 256), 256, 0, h->stream>>>(
        hwc, B, H, W, mean[0], mean[1], mean[2], std_[0], std_[1], std_[2], scale, nchw);
    LAUNCH_CHECK(h);
    return IVIT_OK;
}

}  // extern "C"

// ---------------------------------------------------------------- GEMM launchers
template <bool A16, int EPI>
static int launch_gemm(ivit_handle h, GemmArgs &a, int nb) {
    const int tm = (a.M + GEMM_BM - 1) / GEMM_BM;
    a.tiles_n = (a.N + GEMM_BN - 1) / GEMM_BN;
    dim3 grid((unsigned)(tm * a.tiles_n), (unsigned)nb, 1);
    gemm_nt_kernel<A16, EPI><<<grid, 256, 0, h->stream>>>(a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(h->err, sizeof(h->err), "gemm launch: %s", hipGetErrorString(e));
        return IVIT_ERR_HIP;
    }
    return IVIT_OK;
}

// production path for the QuantLinear GEMMs: K % 64 == 0, int8 A, requant epilogues
template <int EPI>
static int launch_gemm2(ivit_handle h, GemmArgs &a) {
    a.tiles_n = (a.N + G2_BN - 1) / G2_BN;
    constexpr int force_bm = IVIT_OPT_GEMM_BM;
    a.dbg = G2_DBG;                      // 0; timing probes: 1 main loop only, 2 no row stores, 3 no requant arithmetic
    // tile height: estimated time ~ ceil(tiles / resident slots) * rows per tile; 256-row tiles run
    // 2 per CU, 128-row tiles 3 per CU.  Ties go to the larger tile (better operand reuse).
    const long long t256 = (long long)((a.M + 255) / 256) * a.tiles_n, t128 = (long long)((a.M + 127) / 128) * a.tiles_n;
    const long long s256 = 2LL * h->num_cu, s128 = (G2_NSTAGE128 == 2 ? 4LL : 3LL) * h->num_cu;
    const long long c256 = ((t256 + s256 - 1) / s256) * 256, c128 = ((t128 + s128 - 1) / s128) * 128;
    const bool use128 = force_bm ? (force_bm == 128) : (c128 < c256);
    if (use128) gemm_glds_kernel<EPI, 128><<<dim3((unsigned)t128), 256, 0, h->stream>>>(a);
    else gemm_glds_kernel<EPI, 256><<<dim3((unsigned)t256), 512, 0, h->stream>>>(a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(h->err, sizeof(h->err), "gemm2 launch: %s", hipGetErrorString(e));
        return IVIT_ERR_HIP;
    }
    return IVIT_OK;
}
static inline bool use_gemm2(const GemmArgs &a) { return (a.K % 32) == 0 && a.K >= 64 && (a.lda % 16) == 0 && (a.ldb % 16) == 0; }

// short-K streaming kernel (ivit_gemm_wreg.h): K = 96 / 128 / 192 with whole 32-channel tiles in groups of 3 or 2, plain row-major
// operands, enough rows to keep every CU busy
#ifndef IVIT_OPT_GEMM_WREG
#define IVIT_OPT_GEMM_WREG 1
#endif
static inline int wreg_nct(const GemmArgs &a) {
    if (!IVIT_OPT_GEMM_WREG || a.K % 32 || !(a.K == 96 || a.K == 128 || a.K == 192) || a.N % 32 || a.lda != a.K || a.ldb != a.K ||
        a.ldc != a.N || a.M < 8192 || a.inner != 1)
        return 0;
    const int nt = a.N / 32;
    return nt % 3 == 0 ? 3 : (nt % 2 == 0 ? 2 : 0);
}
template <int EPI, int KS, int NCT>
static int launch_wreg_k(ivit_handle h, const GemmArgs &a) {
    const int ncg = a.N / (32 * NCT);
    // resident workgroups per CU of this instantiation (registers, LDS): asked once PER DEVICE (a process-wide
    // static reused the first device's answer on every other one); a race between host threads stores the same value twice
    static std::atomic<int> wpc_dev[IVIT_MAX_DEVICES];
    const bool cached = h->device >= 0 && h->device < IVIT_MAX_DEVICES;
    int wpc = cached ? wpc_dev[h->device].load(std::memory_order_relaxed) : 0;
    if (!wpc) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gemm_wreg_kernel<EPI, KS, NCT>, GW_THREADS, 0) != hipSuccess || n < 1) n = 1;
        wpc = n;
        if (cached) wpc_dev[h->device].store(n, std::memory_order_relaxed);
    }
    int per = (h->num_cu * wpc) / (8 * ncg);
    if (per < 1) per = 1;
    gemm_wreg_kernel<EPI, KS, NCT><<<dim3((unsigned)(8 * ncg * per)), GW_THREADS, 0, h->stream>>>(a);
    LAUNCH_CHECK(h);
    return IVIT_OK;
}
template <int EPI>
static int launch_wreg(ivit_handle h, const GemmArgs &a, int nct) {
    const int ks = a.K / 32;
    if (nct == 3) return ks == 3 ? launch_wreg_k<EPI, 3, 3>(h, a) : (ks == 4 ? launch_wreg_k<EPI, 4, 3>(h, a) : launch_wreg_k<EPI, 6, 3>(h, a));
    return ks == 3 ? launch_wreg_k<EPI, 3, 2>(h, a) : (ks == 4 ? launch_wreg_k<EPI, 4, 2>(h, a) : launch_wreg_k<EPI, 6, 2>(h, a));
}

static GemmArgs linear_args(const int8_t *x, const int8_t *w, const int32_t *bias, int M, int N, int K) {
    GemmArgs a;
    memset(&a, 0, sizeof(a));
    a.A = x; a.B = w; a.M = M; a.N = N; a.K = K;
    a.lda = K; a.ldb = K; a.ldc = N;
    a.inner = 1; a.bias = bias;
    return a;
}

extern "C" {

int ivit_linear_i8(ivit_handle h, const int8_t *x, const int8_t *w, const int32_t *bias, int32_t *acc,
                   int M, int N, int K) {
    CHECK_H(h);
    REQUIRE(h, x && w && acc && M > 0 && N > 0 && K > 0, "bad arguments");
    REQUIRE(h, (K % 16) == 0, "K must be a multiple of 16");
    GemmArgs a = linear_args(x, w, bias, M, N, K);
    a.out = acc;
    return launch_gemm<false, EPI_RAW32>(h, a, 1);
}

int ivit_linear_i8_requant(ivit_handle h, const int8_t *x, const int8_t *w, const int32_t *bias,
                           const ivit_dyadic *dy_ch, int bits, void *out, int M, int N, int K) {
    CHECK_H(h);
    REQUIRE(h, x && w && out && dy_ch && M > 0 && N > 0 && K > 0, "bad arguments");
    REQUIRE(h, (K % 16) == 0, "K must be a multiple of 16");
    REQUIRE(h, bits == 8 || bits == 16, "bits must be 8 or 16");
    GemmArgs a = linear_args(x, w, bias, M, N, K);
    a.out = out; a.dy_ch = dy_ch;
    if (const int nct = wreg_nct(a)) return bits == 8 ? launch_wreg<EPI_RQ8_CH>(h, a, nct) : launch_wreg<EPI_RQ16_CH>(h, a, nct);
    if (use_gemm2(a)) return bits == 8 ? launch_gemm2<EPI_RQ8_CH>(h, a) : launch_gemm2<EPI_RQ16_CH>(h, a);
    return bits == 8 ? launch_gemm<false, EPI_RQ8_CH>(h, a, 1) : launch_gemm<false, EPI_RQ16_CH>(h, a, 1);
}

// QuantLinear -> QuantAct(8) whose consumer is the 16-bit stream: clamp to 8 bits, store int16 (the widening pass ivit_widen_i8_i16 folded
// into the GEMM's stores): the Swin runner's PatchMerging reduction
int ivit_linear_i8_requant8_store16(ivit_handle h, const int8_t *x, const int8_t *w, const int32_t *bias, const ivit_dyadic *dy_ch,
                                    int16_t *out, int M, int N, int K) {
    CHECK_H(h);
    REQUIRE(h, x && w && out && dy_ch && M > 0 && N > 0 && K > 0, "bad arguments");
    REQUIRE(h, (K % 16) == 0, "K must be a multiple of 16");
    GemmArgs a = linear_args(x, w, bias, M, N, K);
    a.out = out; a.dy_ch = dy_ch;
    if (wreg_nct(a) || !use_gemm2(a)) {
        snprintf(h->err, sizeof(h->err), "%s: built for gemm_glds_kernel's shapes (use ivit_linear_i8_requant + ivit_widen_i8_i16)", __func__);
        return IVIT_ERR_UNSUPPORTED;
    }
    return launch_gemm2<EPI_RQ8W16_CH>(h, a);
}

int ivit_linear_i8_requant_residual(ivit_handle h, const int8_t *x, const int8_t *w, const int32_t *bias,
                                    const ivit_dyadic *dy_ch, ivit_dyadic dy_main, ivit_dyadic dy_res,
                                    const int16_t *residual, int16_t *out, int M, int N, int K) {
    CHECK_H(h);
    REQUIRE(h, x && w && out && dy_ch && residual && M > 0 && N > 0 && K > 0, "bad arguments");
    REQUIRE(h, (K % 16) == 0, "K must be a multiple of 16");
    GemmArgs a = linear_args(x, w, bias, M, N, K);
    a.out = out; a.dy_ch = dy_ch; a.dy_main = dy_main; a.dy_res = dy_res; a.residual = residual;
    if (const int nct = wreg_nct(a)) return launch_wreg<EPI_RQ16_CH_RES>(h, a, nct);
    if (use_gemm2(a)) return launch_gemm2<EPI_RQ16_CH_RES>(h, a);
    return launch_gemm<false, EPI_RQ16_CH_RES>(h, a, 1);
}

int ivit_linear_i8_qkv(ivit_handle h, const int8_t *x, const int8_t *w, const int32_t *bias,
                       const ivit_dyadic *dy_ch, int8_t *q, int8_t *k, int8_t *vt, int B, int T, int H,
                       int dh, int ldv) {
    CHECK_H(h);
    REQUIRE(h, x && w && dy_ch && q && k && vt && B > 0 && T > 0 && H > 0 && dh > 0, "bad arguments");
    REQUIRE(h, (dh % 16) == 0, "head dim must be a multiple of 16");
    REQUIRE(h, ldv == 0 || ldv >= T, "ldv < T (0 = v row-major [B*H, T, dh])");
    const int D = H * dh;
    GemmArgs a = linear_args(x, w, bias, B * T, 3 * D, D);
    a.dy_ch = dy_ch; a.q = q; a.k = k; a.vt = vt;
    a.T = T; a.H = H; a.dh = dh; a.ldv = ldv; a.D = D;
    if (use_gemm2(a) && (dh % 16) == 0)
        return launch_gemm2<EPI_QKV>(h, a);
    return launch_gemm<false, EPI_QKV>(h, a, 1);
}

int ivit_bmm_nt_i8(ivit_handle h, const int8_t *A, const int8_t *B, int32_t *C, int nb, int M, int N, int K,
                   int lda, int ldb, int ldc, int64_t strideA, int64_t strideB, int64_t strideC) {
    CHECK_H(h);
    REQUIRE(h, A && B && C && nb > 0 && M > 0 && N > 0 && K > 0, "bad arguments");
    REQUIRE(h, (lda % 16) == 0 && (ldb % 16) == 0 && (strideA % 16) == 0 && (strideB % 16) == 0,
            "lda/ldb/strides must be multiples of 16");
    GemmArgs a;
    memset(&a, 0, sizeof(a));
    a.A = A; a.B = B; a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc;
    a.strideA = strideA; a.strideB = strideB; a.inner = 1; a.sC_outer = strideC; a.out = C;
    return launch_gemm<false, EPI_RAW32>(h, a, nb);
}

int ivit_bmm_nt_u16i8(ivit_handle h, const uint16_t *A, const int8_t *B, int32_t *C, int nb, int M, int N,
                      int K, int lda, int ldb, int ldc, int64_t strideA, int64_t strideB, int64_t strideC) {
    CHECK_H(h);
    REQUIRE(h, A && B && C && nb > 0 && M > 0 && N > 0 && K > 0, "bad arguments");
    REQUIRE(h, (lda % 8) == 0 && (ldb % 16) == 0 && (strideA % 8) == 0 && (strideB % 16) == 0,
            "lda must be a multiple of 8, ldb of 16");
    GemmArgs a;
    memset(&a, 0, sizeof(a));
    a.A = A; a.B = B; a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc;
    a.strideA = strideA; a.strideB = strideB; a.inner = 1; a.sC_outer = strideC; a.out = C;
    return launch_gemm<true, EPI_RAW32>(h, a, nb);
}

int ivit_attn_qk_requant(ivit_handle h, const int8_t *q, const int8_t *k, ivit_dyadic dy, int8_t *scores8,
                         int BH, int T, int dh, int lds) {
    CHECK_H(h);
    REQUIRE(h, q && k && scores8 && BH > 0 && T > 0 && dh > 0, "bad arguments");
    REQUIRE(h, (dh % 16) == 0 && lds >= T, "dh %16, lds >= T");
    GemmArgs a;
    memset(&a, 0, sizeof(a));
    a.A = q; a.B = k; a.M = T; a.N = T; a.K = dh; a.lda = dh; a.ldb = dh; a.ldc = lds;
    a.strideA = (long long)T * dh; a.strideB = (long long)T * dh;
    a.inner = 1; a.sC_outer = (long long)T * lds; a.out = scores8; a.dy_main = dy;
    return launch_gemm<false, EPI_RQ8_S>(h, a, BH);
}

int ivit_attn_pv_requant(ivit_handle h, const uint16_t *p, const int8_t *vt, ivit_dyadic dy, int8_t *ctx8,
                         int B, int H, int T, int dh, int ldp, int ldv) {
    CHECK_H(h);
    REQUIRE(h, p && vt && ctx8 && B > 0 && H > 0 && T > 0 && dh > 0, "bad arguments");
    REQUIRE(h, (ldp % 8) == 0 && (ldv % 16) == 0 && ldp >= T && ldv >= T, "ldp %8, ldv %16, >= T");
    GemmArgs a;
    memset(&a, 0, sizeof(a));
    a.A = p; a.B = vt; a.M = T; a.N = dh; a.K = T; a.lda = ldp; a.ldb = ldv; a.ldc = H * dh;
    a.strideA = (long long)T * ldp; a.strideB = (long long)dh * ldv;
    a.inner = H; a.sC_outer = (long long)T * H * dh; a.sC_inner = dh; a.out = ctx8; a.dy_main = dy;
    return launch_gemm<true, EPI_RQ8_S>(h, a, B * H);
}

}  // extern "C"



// ---------------------------------------------------------------- constants: upload / RCCL broadcast
extern "C" {

int ivit_constants_upload(ivit_handle h, const void *host_blob, size_t bytes, void *device_blob) {
    CHECK_H(h);
    REQUIRE(h, host_blob && device_blob && bytes > 0, "bad arguments");
    hipError_t e = hipMemcpyAsync(device_blob, host_blob, bytes, hipMemcpyHostToDevice, h->stream);
    if (e != hipSuccess) { snprintf(h->err, sizeof(h->err), "ivit_constants_upload: %s", hipGetErrorString(e)); return IVIT_ERR_HIP; }
    return IVIT_OK;
}

// ncclBroadcast(sendbuff, recvbuff, count, datatype, root, comm, stream) resolved from librccl.so at first use, so that
// libivit_hip.so itself carries no link-time dependency on RCCL (single-GPU users never load it)
int ivit_constants_broadcast(ivit_handle h, void *device_blob, size_t bytes, int root, void *rccl_comm) {
    CHECK_H(h);
    REQUIRE(h, device_blob && rccl_comm && bytes > 0 && root >= 0, "bad arguments");
    typedef int (*bcast_fn)(const void *, void *, size_t, int, int, void *, hipStream_t);
    static bcast_fn fn = nullptr;
    static std::once_flag once;
    std::call_once(once, [] {
        void *lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (lib) fn = (bcast_fn)dlsym(lib, "ncclBroadcast");
    });
    if (!fn) { snprintf(h->err, sizeof(h->err), "ivit_constants_broadcast: librccl.so / ncclBroadcast not found"); return IVIT_ERR_UNSUPPORTED; }
    const int rc = fn(device_blob, device_blob, bytes, /* ncclUint8 */ 1, root, rccl_comm, h->stream);   // in place: one ring over xGMI
    if (rc != 0) { snprintf(h->err, sizeof(h->err), "ivit_constants_broadcast: ncclBroadcast returned %d", rc); return IVIT_ERR_HIP; }
    return IVIT_OK;
}

}  // extern "C"

// ---------------------------------------------------------------- linear plans (frozen QuantLinear)
// One wavefront per output channel: c[n] = m*2^-e, zmax[n] = 128 * sum_k |W[n,k]| + |bias[n]| >= |acc + bias|.
// bad bit 0: |zmax * c| >= 2^31 (the magic-number rounding of the pipelined epilogue could wrap);
// bad bit 1: zmax * |m| >= 2^53 (z*m inexact in fp64: the single-FMA form is not the reference's two roundings).
__global__ __launch_bounds__(256) void linear_plan_kernel(const int8_t *__restrict__ w, const int32_t *__restrict__ bias,
                                                          const ivit_dyadic *__restrict__ dy, int N, int K,
                                                          double *__restrict__ cq, int *__restrict__ bad) {
    const int lane = threadIdx.x & 63, n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const int8_t *wp = w + (long long)n * K;
    int l1 = 0;
    for (int k = lane; k < K; k += 64) l1 += abs((int)wp[k]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) l1 += __shfl_xor(l1, o);
    if (lane == 0) {
        const double c = dy[n].m * dy[n].r;
        cq[n] = c;
        const double zmax = 128.0 * (double)l1 + fabs((double)(bias ? bias[n] : 0));
        int b = 0;
        if (!(fabs(c) * zmax < 2147483000.0)) b |= 1;
        if (!(fabs(dy[n].m) * zmax < 9007199254740992.0)) b |= 4;      // bound fails: linear_plan_fma_kernel decides
        if (b) atomicOr(bad, b);
    }
}

// Channels whose |z * m| may exceed 2^53 (the cheap bound above failed, flag 4): is the one-FMA requant
//   f1(z) = lo32(fma(double(z), c, 1.5 * 2^52)) = RNE(z * c)            (one rounding)
// still the reference's  f2(z) = RNE(fl64(z * m) * 2^-e)                (two roundings, quant_utils.py:229-231)  on [-zmax, zmax]?
// Both are non-decreasing step functions of the integer z; they agree on the clamped 16-bit range (and so on the 8-bit one)
// iff every step of f1 into a value k in (-32768, 32768] is a step of f2 at the same z.  One block per channel, 65536 steps
// over 256 threads; steps whose |z * m| < 2^53 are exact in both and skipped.  A failing channel sets flag 2.
__global__ __launch_bounds__(256) void linear_plan_fma_kernel(const int8_t *__restrict__ w, const int32_t *__restrict__ bias,
                                                              const ivit_dyadic *__restrict__ dy, int N, int K, int *__restrict__ bad) {
    __shared__ int s_l1, s_fail;
    const int n = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) { s_l1 = 0; s_fail = 0; }
    __syncthreads();
    int l1 = 0;
    for (int k = tid; k < K; k += 256) l1 += abs((int)w[(long long)n * K + k]);
    atomicAdd(&s_l1, l1);
    __syncthreads();
    const double m = dy[n].m, r = dy[n].r, c = m * r;
    const double zmax = 128.0 * (double)s_l1 + fabs((double)(bias ? bias[n] : 0));
    if (fabs(m) * zmax < 9007199254740992.0) return;               // exact products: nothing to prove
    if (!(c > 0.0) || !(zmax < 2147483000.0)) { if (tid == 0) atomicOr(bad, 2); return; }
    auto f1 = [&](double z) { return __double2loint(__builtin_fma(z, c, 6755399441055744.0)); };
    auto f2 = [&](double z) { return (int)__builtin_rint((z * m) * r); };
    bool fail = false;
    for (int i = 0; i < 256 && !fail; ++i) {
        const int k = -32767 + tid + 256 * i;                        // step INTO k
        double z0 = __builtin_ceil(((double)k - 0.5) / c);
        for (int it = 0; it < 4 && z0 - 1.0 >= -zmax && f1(z0 - 1.0) >= k; ++it) z0 -= 1.0;
        for (int it = 0; it < 4 && z0 <= zmax && f1(z0) < k; ++it) z0 += 1.0;
        if (fabs(z0) * fabs(m) < 9.0e15) continue;                   // |z * m| < 2^53 around this step
        if (z0 <= zmax && z0 >= -zmax && (f1(z0) < k || f2(z0) < k)) fail = true;
        if (z0 - 1.0 <= zmax && z0 - 1.0 >= -zmax && (f1(z0 - 1.0) >= k || f2(z0 - 1.0) >= k)) fail = true;
    }
    if (fail) atomicOr(&s_fail, 1);
    __syncthreads();
    if (tid == 0 && s_fail) atomicOr(bad, 2);
}

struct ivit_linear_plan_s {
    const int8_t *w;
    const int32_t *bias;        // caller's pointer (may be null)
    const int32_t *bias_eff;    // never null: zeros inside `dev` when the layer has no bias
    const ivit_dyadic *dy;
    int N, K;
    char *dev;                  // one allocation: cq[N] doubles | zero bias (optional) | flag word | 1 KB store scratch
    double *cq;
    void *dummy;
    int pipelined_ok, single_fma_ok;
    v4i *wf;                    // ivit_linear_plan_prepare_ws: the weights in gemm_ws_qkv_kernel's fragment order (own allocation), or null
    int device;                 // where `dev` lives: destroy / debug reads run there whatever the caller's current device is
};

extern "C" {

int ivit_linear_plan_create(ivit_handle h, const int8_t *w, const int32_t *bias, const ivit_dyadic *dy_ch, int N, int K,
                            ivit_linear_plan *out) {
    CHECK_H(h);
    REQUIRE(h, w && dy_ch && out && N > 0 && K > 0, "bad arguments");
    const size_t cq_bytes = ((size_t)N * 8 + 255) & ~(size_t)255, b_bytes = ((size_t)N * 4 + 255) & ~(size_t)255;
    char *dev = nullptr;
    hipError_t e = hipMalloc((void **)&dev, cq_bytes + b_bytes + 256 + 1024 + 8192);
    if (e != hipSuccess) { snprintf(h->err, sizeof(h->err), "ivit_linear_plan_create: hipMalloc: %s", hipGetErrorString(e)); return IVIT_ERR_HIP; }
    int *flag = (int *)(dev + cq_bytes + b_bytes);
    int host_bad = 3;
    bool ok = hipMemsetAsync(dev + cq_bytes, 0, b_bytes + 256 + 1024 + 8192, h->stream) == hipSuccess;
    if (ok) {
        linear_plan_kernel<<<(N + 3) / 4, 256, 0, h->stream>>>(w, bias, dy_ch, N, K, (double *)dev, flag);
        ok = hipGetLastError() == hipSuccess;
    }
    ok = ok && hipMemcpyAsync(&host_bad, flag, sizeof(int), hipMemcpyDeviceToHost, h->stream) == hipSuccess;
    ok = ok && hipStreamSynchronize(h->stream) == hipSuccess;      // plan creation is a build-time call
    if (ok && (host_bad & 4) && !(host_bad & 1)) {                 // the cheap one-FMA bound failed somewhere: exact proof per channel
        linear_plan_fma_kernel<<<N, 256, 0, h->stream>>>(w, bias, dy_ch, N, K, flag);
        ok = hipGetLastError() == hipSuccess;
        ok = ok && hipMemcpyAsync(&host_bad, flag, sizeof(int), hipMemcpyDeviceToHost, h->stream) == hipSuccess;
        ok = ok && hipStreamSynchronize(h->stream) == hipSuccess;
    } else if (host_bad & 4) {
        host_bad |= 2;
    }
    if (!ok) {
        snprintf(h->err, sizeof(h->err), "ivit_linear_plan_create: HIP error");
        (void)hipFree(dev);
        return IVIT_ERR_HIP;
    }
    ivit_linear_plan_s *p = new (std::nothrow) ivit_linear_plan_s();
    if (!p) { (void)hipFree(dev); return IVIT_ERR_HIP; }
    p->w = w; p->bias = bias; p->dy = dy_ch; p->N = N; p->K = K; p->dev = dev; p->cq = (double *)dev;
    p->bias_eff = bias ? bias : (const int32_t *)(dev + cq_bytes);
    p->dummy = dev + cq_bytes + b_bytes + 256;
    p->pipelined_ok = !(host_bad & 1);
    p->single_fma_ok = !(host_bad & 2);
    p->device = h->device;
    p->wf = nullptr;
    *out = p;
    return IVIT_OK;
}

int ivit_linear_plan_destroy(ivit_linear_plan p) {
    if (!p) return IVIT_ERR_INVALID;
    ivit_device_guard g;
    if (!g.enter(p->device)) return IVIT_ERR_HIP;
    (void)hipFree(p->dev);
    if (p->wf) (void)hipFree(p->wf);
    delete p;
    return IVIT_OK;
}

int ivit_debug_plan_scratch(ivit_linear_plan p, void *host_dst, int nbytes) {
    if (!p || !host_dst || nbytes <= 0 || nbytes > 8192) return IVIT_ERR_INVALID;
    ivit_device_guard g;
    if (!g.enter(p->device)) return IVIT_ERR_HIP;
    if (hipDeviceSynchronize() != hipSuccess) return IVIT_ERR_HIP;
    return hipMemcpy(host_dst, (char *)p->dummy + 1024, (size_t)nbytes, hipMemcpyDeviceToHost) == hipSuccess ? IVIT_OK : IVIT_ERR_HIP;
}

int ivit_linear_plan_query(ivit_linear_plan p, int *pipelined_ok, int *single_fma_ok) {
    if (!p) return IVIT_ERR_INVALID;
    if (pipelined_ok) *pipelined_ok = p->pipelined_ok;
    if (single_fma_ok) *single_fma_ok = p->single_fma_ok;
    return IVIT_OK;
}

}  // extern "C"

// persistent pipelined kernel: shapes it is built for (anything else runs on gemm_glds_kernel / gemm_nt_kernel)
// IVIT_GEMM3: bit mask of the epilogues that run on the persistent pipelined kernels — 1 requant (8/16-bit), 2 qkv
// scatter, 4 requant + residual; the others stay on the launch-per-tile kernels (A/B and fallback).
static inline bool use_gemm3(const ivit_linear_plan_s *pl, const GemmArgs &a, int epi_bit) {
    constexpr int on = IVIT_OPT_GEMM3;
    // the residual flavour on a narrow output (N = 384: three channel tiles per 256-token panel, 77 % balance, and its
    // 32 resident residual registers) measured no better in-model than the launch-per-tile kernel: N >= 512 only
    constexpr int res_min_n = IVIT_OPT_GEMM3_RES_MIN_N;
    if (epi_bit == 4 && a.N < res_min_n) return false;
    return (on & epi_bit) && pl->pipelined_ok && (a.K % 64) == 0 && a.K >= 320 && (a.N % 16) == 0 && (a.ldc % 16) == 0 &&
           (a.lda % 16) == 0 && (a.ldb % 16) == 0 && a.M >= 128;
}

// ---- D = 384 qkv on the register-resident-weights kernel (ivit_gemm_ws.h), with or without norm1 in its prologue
#ifndef IVIT_OPT_QKV_WS
#define IVIT_OPT_QKV_WS 1               // A/B: ivit_linear_i8_qkv_planned(ldv = 0) on gemm_ws_qkv_kernel where the plan is prepared
#endif
static inline bool qkv_ws_ok(const ivit_linear_plan_s *pl, int B, int T, int H, int dh) {
    return pl->wf && dh == 64 && pl->K == WS_K && pl->N == 3 * H * dh && pl->N % 192 == 0 && (long long)B * H * T * 64 < (1ll << 31) && (long long)B * T < (1ll << 26);
}
static int launch_qkv_ws(ivit_handle h, const ivit_linear_plan_s *pl, const int8_t *x8, const int16_t *x16, float ln_s,
                         const float *ln_bias_int, const float *ln_sc, const ivit_dyadic *ln_dy, int8_t *q, int8_t *k, int8_t *v,
                         int B, int T, int H) {
    if (h->device != pl->device) { snprintf(h->err, sizeof(h->err), "qkv: plan and handle live on different devices"); return IVIT_ERR_INVALID; }
    WsArgs a;
    a.x = x8; a.wf = pl->wf; a.bias = pl->bias_eff; a.cq = pl->cq; a.q = q; a.k = k; a.v = v;
    a.M = B * T; a.N = pl->N; a.T = T; a.H = H; a.dummy = pl->dummy;
    a.x16 = x16; a.ln_s = ln_s; a.ln_bias_int = ln_bias_int; a.ln_sc = ln_sc; a.ln_dy = ln_dy; a.trace = nullptr;
    a.residual = nullptr; a.out16 = nullptr; a.cm = a.cr = 0.0; a.ln_out8 = nullptr;
    static std::atomic<bool> attr_dev[IVIT_MAX_DEVICES];
    const bool cached = h->device >= 0 && h->device < IVIT_MAX_DEVICES;
    if (!cached || !attr_dev[h->device].load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute((const void *)gemm_ws_qkv_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, WS_SMEM);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)gemm_ws_qkv_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, WS_SMEM);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)gemm_ws_qkv_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, WS_SMEM);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)gemm_ws_qkv_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, WS_SMEM);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)gemm_ws_qkv_kernel<true, true, WS_EPI_RQ8>, hipFuncAttributeMaxDynamicSharedMemorySize, WS_SMEM);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)gemm_ws_qkv_kernel<true, false, WS_EPI_RQ8>, hipFuncAttributeMaxDynamicSharedMemorySize, WS_SMEM);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)gemm_ws_qkv_kernel<false, true, WS_EPI_RQ8>, hipFuncAttributeMaxDynamicSharedMemorySize, WS_SMEM);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)gemm_ws_qkv_kernel<false, false, WS_EPI_RQ8>, hipFuncAttributeMaxDynamicSharedMemorySize, WS_SMEM);
        if (e != hipSuccess) { snprintf(h->err, sizeof(h->err), "qkv attr: %s", hipGetErrorString(e)); return IVIT_ERR_HIP; }
        if (cached) attr_dev[h->device].store(true, std::memory_order_release);
    }
    // one workgroup per CU, each with a contiguous range of 32-token tiles
    const int ntt = (a.M + 31) / 32;
    const unsigned grid = (unsigned)(ntt < persistent_cus(h) ? ntt : persistent_cus(h));
    const bool fma = pl->single_fma_ok;
    if (H == 0) {       // plain 8-bit output [M][N] (q = out8; T = 1)
        a.T = 1; a.H = 1;
        if (x16) {
            if (fma) gemm_ws_qkv_kernel<true, true, WS_EPI_RQ8><<<grid, WS_THREADS, WS_SMEM, h->stream>>>(a);
            else gemm_ws_qkv_kernel<false, true, WS_EPI_RQ8><<<grid, WS_THREADS, WS_SMEM, h->stream>>>(a);
        } else {
            if (fma) gemm_ws_qkv_kernel<true, false, WS_EPI_RQ8><<<grid, WS_THREADS, WS_SMEM, h->stream>>>(a);
            else gemm_ws_qkv_kernel<false, false, WS_EPI_RQ8><<<grid, WS_THREADS, WS_SMEM, h->stream>>>(a);
        }
    } else if (x16) {
        if (fma) gemm_ws_qkv_kernel<true, true><<<grid, WS_THREADS, WS_SMEM, h->stream>>>(a);
        else gemm_ws_qkv_kernel<false, true><<<grid, WS_THREADS, WS_SMEM, h->stream>>>(a);
    } else {
        if (fma) gemm_ws_qkv_kernel<true, false><<<grid, WS_THREADS, WS_SMEM, h->stream>>>(a);
        else gemm_ws_qkv_kernel<false, false><<<grid, WS_THREADS, WS_SMEM, h->stream>>>(a);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(h->err, sizeof(h->err), "qkv launch: %s", hipGetErrorString(e)); return IVIT_ERR_HIP; }
    return IVIT_OK;
}

#ifndef IVIT_OPT_RES_WS
#define IVIT_OPT_RES_WS 1               // A/B: ivit_linear_i8_requant_residual_planned on gemm_ws_qkv_kernel<.., EPI_RES16> where the plan is prepared
#endif
static int launch_res_ws(ivit_handle h, const ivit_linear_plan_s *pl, const int8_t *x8, double cm, double cr, const int16_t *residual,
                         int16_t *out, int M, float ln_s = 0.f, const float *ln_bias_int = nullptr, const float *ln_sc = nullptr,
                         const ivit_dyadic *ln_dy = nullptr, int8_t *ln_out8 = nullptr) {
    if (h->device != pl->device) { snprintf(h->err, sizeof(h->err), "linear: plan and handle live on different devices"); return IVIT_ERR_INVALID; }
    WsArgs a;
    a.x = x8; a.wf = pl->wf; a.bias = pl->bias_eff; a.cq = pl->cq; a.q = a.k = a.v = nullptr;
    a.M = M; a.N = pl->N; a.T = 1; a.H = 1; a.dummy = pl->dummy;
    a.x16 = nullptr; a.ln_s = ln_s; a.ln_bias_int = ln_bias_int; a.ln_sc = ln_sc; a.ln_dy = ln_dy; a.trace = nullptr;
    a.residual = residual; a.out16 = out; a.cm = cm; a.cr = cr; a.ln_out8 = ln_out8;
    static std::atomic<bool> attr_dev[IVIT_MAX_DEVICES];
    const bool cached = h->device >= 0 && h->device < IVIT_MAX_DEVICES;
    if (!cached || !attr_dev[h->device].load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute((const void *)gemm_ws_qkv_kernel<true, false, WS_EPI_RES16>, hipFuncAttributeMaxDynamicSharedMemorySize, WS_SMEM);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)gemm_ws_qkv_kernel<false, false, WS_EPI_RES16>, hipFuncAttributeMaxDynamicSharedMemorySize, WS_SMEM);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)gemm_ws_qkv_kernel<true, true, WS_EPI_RES16>, hipFuncAttributeMaxDynamicSharedMemorySize, WS_SMEM);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)gemm_ws_qkv_kernel<false, true, WS_EPI_RES16>, hipFuncAttributeMaxDynamicSharedMemorySize, WS_SMEM);
        if (e != hipSuccess) { snprintf(h->err, sizeof(h->err), "linear attr: %s", hipGetErrorString(e)); return IVIT_ERR_HIP; }
        if (cached) attr_dev[h->device].store(true, std::memory_order_release);
    }
    const int ntt = (M + 31) / 32;
    const unsigned grid = (unsigned)(ntt < persistent_cus(h) ? ntt : persistent_cus(h));
    if (ln_out8) {
        if (pl->single_fma_ok) gemm_ws_qkv_kernel<true, true, WS_EPI_RES16><<<grid, WS_THREADS, WS_SMEM, h->stream>>>(a);
        else gemm_ws_qkv_kernel<false, true, WS_EPI_RES16><<<grid, WS_THREADS, WS_SMEM, h->stream>>>(a);
    } else if (pl->single_fma_ok) gemm_ws_qkv_kernel<true, false, WS_EPI_RES16><<<grid, WS_THREADS, WS_SMEM, h->stream>>>(a);
    else gemm_ws_qkv_kernel<false, false, WS_EPI_RES16><<<grid, WS_THREADS, WS_SMEM, h->stream>>>(a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(h->err, sizeof(h->err), "linear launch: %s", hipGetErrorString(e)); return IVIT_ERR_HIP; }
    return IVIT_OK;
}

template <int EPI>
static int launch_gemm3(ivit_handle h, const ivit_linear_plan_s *pl, GemmArgs &a) {
    a.tiles_n = (a.N + 127) / 128;
    a.cq = pl->cq;
    a.bias = pl->bias_eff;
    a.dummy = pl->dummy;
    constexpr int force_fma = IVIT_OPT_GEMM3_FMA, wg_per_cu = 2;
    constexpr bool astat_on = true;
    a.dbg = 0;
    const bool fma = force_fma >= 0 ? (force_fma != 0 && pl->single_fma_ok) : (pl->single_fma_ok != 0);
    // gemm_as_kernel: K = n * 384; the qkv scatter additionally needs whole units inside one of q / k / v and whole
    // 32-channel groups inside one head.  Its operand offsets are 32-bit and its epilogue goes through buffer
    // resources (offsets < 2^31, out-of-range lanes parked at 0x80000000): larger tensors take the 64-bit kernels.
    const long long out_bytes = (EPI == EPI_QKV) ? std::max((long long)a.M * a.D, (long long)(a.M / (a.T > 0 ? a.T : 1) + 1) * a.D * a.ldv)
                                                 : (long long)a.M * a.ldc * ((EPI == EPI_RQ8_CH) ? 1 : 2);
    const bool astat = astat_on && (a.K % (GA_BK * GA_NK)) == 0 && a.M >= 256 && (a.N % 32) == 0 &&
                       (long long)a.M * a.lda < (1LL << 32) && (long long)a.N * a.ldb < (1LL << 32) &&
                       out_bytes < (1LL << 31) &&
                       (EPI != EPI_QKV || ((a.D % 128) == 0 && (a.dh % 32) == 0));
    if (astat) {
        const long long nunits = (long long)((a.M + 255) / 256) * a.tiles_n;
        long long grid = h->num_cu;
        if (grid > nunits) grid = nunits;
        const dim3 g((unsigned)grid);
        if (a.K == GA_BK * GA_NK) {
            if (fma) gemm_as_kernel<EPI, false, true><<<g, 512, 0, h->stream>>>(a);
            else gemm_as_kernel<EPI, false, false><<<g, 512, 0, h->stream>>>(a);
        } else {
            if (fma) gemm_as_kernel<EPI, true, true><<<g, 512, 0, h->stream>>>(a);
            else gemm_as_kernel<EPI, true, false><<<g, 512, 0, h->stream>>>(a);
        }
    } else {
        const long long nunits = (long long)((a.M + 127) / 128) * a.tiles_n;
        long long grid = (long long)h->num_cu * wg_per_cu;
        if (grid > nunits) grid = nunits;
        if (fma) gemm_ps_kernel<EPI, true, false><<<dim3((unsigned)grid), 256, 0, h->stream>>>(a);
        else gemm_ps_kernel<EPI, false, false><<<dim3((unsigned)grid), 256, 0, h->stream>>>(a);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(h->err, sizeof(h->err), "gemm3 launch: %s", hipGetErrorString(e));
        return IVIT_ERR_HIP;
    }
    return IVIT_OK;
}

extern "C" {

int ivit_linear_i8_requant_planned(ivit_handle h, ivit_linear_plan pl, const int8_t *x, int bits, void *out, int M) {
    CHECK_H(h);
    REQUIRE(h, pl && x && out && M > 0, "bad arguments");
    REQUIRE(h, bits == 8 || bits == 16, "bits must be 8 or 16");
    GemmArgs a = linear_args(x, pl->w, pl->bias, M, pl->N, pl->K);
    a.out = out; a.dy_ch = pl->dy;
    if (IVIT_OPT_QKV_WS && bits == 8 && pl->wf && pl->K == WS_K && M < (1 << 26))      // prepared plan: tokens of a CU in LDS, weight slabs in registers
        return launch_qkv_ws(h, pl, x, nullptr, 0.f, nullptr, nullptr, nullptr, (int8_t *)out, nullptr, nullptr, M, 1, 0);
    if (use_gemm3(pl, a, 1)) return bits == 8 ? launch_gemm3<EPI_RQ8_CH>(h, pl, a) : launch_gemm3<EPI_RQ16_CH>(h, pl, a);
    return ivit_linear_i8_requant(h, x, pl->w, pl->bias, pl->dy, bits, out, M, pl->N, pl->K);
}

int ivit_layernorm_linear_i8_requant_planned(ivit_handle h, ivit_linear_plan pl, const int16_t *x16, float scale, const float *bias_int,
                                             const float *sc, const ivit_dyadic *ln_dy, int8_t *out8, int M) {
    CHECK_H(h);
    REQUIRE(h, pl && x16 && bias_int && sc && ln_dy && out8 && M > 0, "bad arguments");
    if (!(pl->wf && pl->K == WS_K && M < (1 << 26))) {
        snprintf(h->err, sizeof(h->err), "%s: needs ivit_linear_plan_prepare_ws on a K = 384 plan", __func__);
        return IVIT_ERR_UNSUPPORTED;
    }
    return launch_qkv_ws(h, pl, nullptr, x16, scale, bias_int, sc, ln_dy, out8, nullptr, nullptr, M, 1, 0);
}

int ivit_linear_i8_requant_residual_planned(ivit_handle h, ivit_linear_plan pl, const int8_t *x, ivit_dyadic dy_main,
                                            ivit_dyadic dy_res, const int16_t *residual, int16_t *out, int M) {
    CHECK_H(h);
    REQUIRE(h, pl && x && out && residual && M > 0, "bad arguments");
    GemmArgs a = linear_args(x, pl->w, pl->bias, M, pl->N, pl->K);
    a.out = out; a.dy_ch = pl->dy; a.dy_main = dy_main; a.dy_res = dy_res; a.residual = residual;
    const bool res_fast = fabs(dy_main.m * dy_main.r) < RQ_FAST_CLIM && fabs(dy_res.m * dy_res.r) < RQ_FAST_CLIM;
    if (IVIT_OPT_RES_WS && pl->wf && res_fast && pl->K == WS_K && M < (1 << 26))
        return launch_res_ws(h, pl, x, dy_main.m * dy_main.r, dy_res.m * dy_res.r, residual, out, M);
    if (use_gemm3(pl, a, 4) && res_fast) return launch_gemm3<EPI_RQ16_CH_RES>(h, pl, a);
    return ivit_linear_i8_requant_residual(h, x, pl->w, pl->bias, pl->dy, dy_main, dy_res, residual, out, M, pl->N, pl->K);
}

int ivit_linear_i8_requant_residual_layernorm_planned(ivit_handle h, ivit_linear_plan pl, const int8_t *x, ivit_dyadic dy_main,
                                                      ivit_dyadic dy_res, const int16_t *residual, int16_t *out, int M, float ln_scale,
                                                      const float *ln_bias_int, const float *ln_sc, const ivit_dyadic *ln_dy,
                                                      int8_t *ln_out8) {
    CHECK_H(h);
    REQUIRE(h, pl && x && out && residual && M > 0 && ln_bias_int && ln_sc && ln_dy && ln_out8, "bad arguments");
    const bool res_fast = fabs(dy_main.m * dy_main.r) < RQ_FAST_CLIM && fabs(dy_res.m * dy_res.r) < RQ_FAST_CLIM;
    if (!(pl->wf && res_fast && pl->K == WS_K && pl->N == WS_K && M < (1 << 26))) {
        snprintf(h->err, sizeof(h->err), "%s: needs ivit_linear_plan_prepare_ws on a 384 x 384 plan and residual multipliers in the fast range", __func__);
        return IVIT_ERR_UNSUPPORTED;
    }
    return launch_res_ws(h, pl, x, dy_main.m * dy_main.r, dy_res.m * dy_res.r, residual, out, M, ln_scale, ln_bias_int, ln_sc, ln_dy, ln_out8);
}

int ivit_linear_i8_qkv_planned(ivit_handle h, ivit_linear_plan pl, const int8_t *x, int8_t *q, int8_t *k, int8_t *vt,
                               int B, int T, int H, int dh, int ldv) {
    CHECK_H(h);
    REQUIRE(h, pl && x && q && k && vt && B > 0 && T > 0 && H > 0 && dh > 0, "bad arguments");
    REQUIRE(h, (dh % 16) == 0, "head dim must be a multiple of 16");
    REQUIRE(h, ldv == 0 || ldv >= T, "ldv < T (0 = v row-major [B*H, T, dh])");
    const int D = H * dh;
    REQUIRE(h, pl->N == 3 * D && pl->K == D, "plan shape is not [3*H*dh, H*dh]");
    GemmArgs a = linear_args(x, pl->w, pl->bias, B * T, 3 * D, D);
    a.dy_ch = pl->dy; a.q = q; a.k = k; a.vt = vt;
    a.T = T; a.H = H; a.dh = dh; a.ldv = ldv; a.D = D;
    if (IVIT_OPT_QKV_WS && ldv == 0 && qkv_ws_ok(pl, B, T, H, dh)) return launch_qkv_ws(h, pl, x, nullptr, 0.f, nullptr, nullptr, nullptr, q, k, vt, B, T, H);
    if (use_gemm3(pl, a, 2) && (long long)B * T < (1 << 23)) return launch_gemm3<EPI_QKV>(h, pl, a);
    return ivit_linear_i8_qkv(h, x, pl->w, pl->bias, pl->dy, q, k, vt, B, T, H, dh, ldv);
}

int ivit_linear_plan_prepare_ws(ivit_handle h, ivit_linear_plan pl) {
    CHECK_H(h);
    REQUIRE(h, pl, "null plan");
    REQUIRE(h, h->device == pl->device, "plan and handle live on different devices");
    if (pl->wf) return IVIT_OK;
    if (pl->K != WS_K || pl->N % 64 != 0 || pl->N > WS_MAXN || !pl->pipelined_ok) {
        snprintf(h->err, sizeof(h->err), "%s: built for K = 384, N a multiple of 64 up to %d, |(acc + bias) * c| < 2^31", __func__, WS_MAXN);
        return IVIT_ERR_UNSUPPORTED;
    }
    v4i *wf = nullptr;
    hipError_t e = hipMalloc((void **)&wf, (size_t)pl->N * WS_K);
    if (e != hipSuccess) { snprintf(h->err, sizeof(h->err), "%s: hipMalloc: %s", __func__, hipGetErrorString(e)); return IVIT_ERR_HIP; }
    ws_swizzle_kernel<<<64, 256, 0, h->stream>>>(pl->w, wf, pl->N);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);       // plan preparation is a build-time call
    if (e != hipSuccess) { (void)hipFree(wf); snprintf(h->err, sizeof(h->err), "%s: %s", __func__, hipGetErrorString(e)); return IVIT_ERR_HIP; }
    pl->wf = wf;
    return IVIT_OK;
}

int ivit_layernorm_linear_i8_qkv_planned(ivit_handle h, ivit_linear_plan pl, const int16_t *x16, float scale, const float *bias_int,
                                         const float *sc, const ivit_dyadic *ln_dy, int8_t *q, int8_t *k, int8_t *v, int B, int T,
                                         int H, int dh) {
    CHECK_H(h);
    REQUIRE(h, pl && x16 && bias_int && sc && ln_dy && q && k && v && B > 0 && T > 0 && H > 0 && dh > 0, "bad arguments");
    REQUIRE(h, pl->N == 3 * H * dh && pl->K == H * dh, "plan shape is not [3*H*dh, H*dh]");
    if (!qkv_ws_ok(pl, B, T, H, dh)) {
        snprintf(h->err, sizeof(h->err), "%s: needs ivit_linear_plan_prepare_ws on a K = 384, dh = 64 plan and B*H*T*64 < 2^31", __func__);
        return IVIT_ERR_UNSUPPORTED;
    }
    return launch_qkv_ws(h, pl, nullptr, x16, scale, bias_int, sc, ln_dy, q, k, v, B, T, H);
}

}  // extern "C"

#ifndef IVIT_OPT_SWIN_MLP_RS
#define IVIT_OPT_SWIN_MLP_RS 1          // A/B builds: 0 = the narrow-stage fused Mlp always on the phase-by-phase kernel
#endif
// ---- fused Mlp (+ residual QuantAct) for D = 384, hidden = 1536 (ivit_mlp.h)
#ifndef IVIT_OPT_MLP_RS
#define IVIT_OPT_MLP_RS 1               // A/B builds: 0 = the shape-based default never picks the role-split kernel
#endif
#ifndef IVIT_OPT_MLP_LNH
#define IVIT_OPT_MLP_LNH 1               // ivit_layernorm_mlp_fused_planned: 1 = the LayerNorm of every row first, 2 = the first unit's rows first, the rest
#endif                                   // by the consumer waves beside the producers' first fc1 (bit-exact, 2.69 against 2.65 ms per forward: not the default)
struct ivit_mlp_plan_s {
    ivit_linear_plan fc1, fc2;      // borrowed: must outlive this plan
    v4i *w1f, *w2f;                 // fragment-ordered copies of the two weight matrices (one allocation)
    v4i *w1r, *w2r;                 // the same in the role-split kernel's order (ivit_mlp_rs.h), same allocation
    int fma;                        // both layers: one fused rounding == the reference's two
    int kernel;                     // 0 = by shape, 1 = lock-step (mlp384_kernel), 2 = role-split (mlp384rs_kernel): ivit_mlp_plan_select
    int device;
};

extern "C" {

int ivit_mlp_plan_create(ivit_handle h, ivit_linear_plan fc1, ivit_linear_plan fc2, ivit_mlp_plan *out) {
    CHECK_H(h);
    REQUIRE(h, fc1 && fc2 && out, "null argument");
    if (fc1->K != MLP_C || fc1->N != MLP_HD || fc2->K != MLP_HD || fc2->N != MLP_C) {
        snprintf(h->err, sizeof(h->err), "%s: built for 384 -> 1536 -> 384", __func__);
        return IVIT_ERR_UNSUPPORTED;
    }
    if (!fc1->pipelined_ok || !fc2->pipelined_ok) {
        snprintf(h->err, sizeof(h->err), "%s: |(acc + bias) * c| < 2^31 not provable for these weights", __func__);
        return IVIT_ERR_UNSUPPORTED;
    }
    const size_t wbytes = (size_t)MLP_C * MLP_HD;
    char *dev = nullptr;
    if (hipMalloc((void **)&dev, 4 * wbytes) != hipSuccess) { snprintf(h->err, sizeof(h->err), "%s: hipMalloc failed", __func__); return IVIT_ERR_HIP; }
    mlp_swizzle_kernel<<<256, 256, 0, h->stream>>>(fc1->w, MLP_HD, MLP_C, (v4i *)dev);
    mlp_swizzle_kernel<<<256, 256, 0, h->stream>>>(fc2->w, MLP_C, MLP_HD, (v4i *)(dev + wbytes));
    rs_swizzle_w1_kernel<<<144, 256, 0, h->stream>>>(fc1->w, (v4i *)(dev + 2 * wbytes));
    rs_swizzle_w2_kernel<<<144, 256, 0, h->stream>>>(fc2->w, (v4i *)(dev + 3 * wbytes));
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess) {     // plan creation is a build-time call
        (void)hipFree(dev);
        snprintf(h->err, sizeof(h->err), "%s: HIP error", __func__);
        return IVIT_ERR_HIP;
    }
    ivit_mlp_plan_s *p = new (std::nothrow) ivit_mlp_plan_s();
    if (!p) { (void)hipFree(dev); return IVIT_ERR_HIP; }
    p->fc1 = fc1; p->fc2 = fc2; p->w1f = (v4i *)dev; p->w2f = (v4i *)(dev + wbytes);
    p->w1r = (v4i *)(dev + 2 * wbytes); p->w2r = (v4i *)(dev + 3 * wbytes);
    p->fma = fc1->single_fma_ok && fc2->single_fma_ok;
    p->kernel = 0;
    p->device = h->device;
    {   // the dynamic-LDS attributes of the kernels this plan will launch: once, here
        const void *fn = p->fma ? (const void *)mlp384_kernel<true> : (const void *)mlp384_kernel<false>;
        const void *fr = p->fma ? (const void *)mlp384rs_kernel<true> : (const void *)mlp384rs_kernel<false>;
        const void *fl = p->fma ? (const void *)mlp384rs_kernel<true, IVIT_OPT_MLP_LNH> : (const void *)mlp384rs_kernel<false, IVIT_OPT_MLP_LNH>;
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, MLP_SMEM);
        if (e == hipSuccess) e = hipFuncSetAttribute(fr, hipFuncAttributeMaxDynamicSharedMemorySize, RS_SMEM);
        if (e == hipSuccess) e = hipFuncSetAttribute(fl, hipFuncAttributeMaxDynamicSharedMemorySize, RS_SMEM);
        if (e != hipSuccess) {
            snprintf(h->err, sizeof(h->err), "%s: attr: %s", __func__, hipGetErrorString(e));
            (void)hipFree(dev);
            delete p;
            return IVIT_ERR_HIP;
        }
    }
    *out = p;
    return IVIT_OK;
}

int ivit_mlp_plan_select(ivit_mlp_plan p, int kernel) {
    if (!p || kernel < 0 || kernel > 2) return IVIT_ERR_INVALID;
    p->kernel = kernel;
    return IVIT_OK;
}

int ivit_mlp_plan_destroy(ivit_mlp_plan p) {
    if (!p) return IVIT_ERR_INVALID;
    ivit_device_guard g;
    if (!g.enter(p->device)) return IVIT_ERR_HIP;
    (void)hipFree(p->w1f);
    delete p;
    return IVIT_OK;
}

static int mlp_fused_launch(ivit_handle h, ivit_mlp_plan p, const int8_t *x, const int8_t *gelu_table, ivit_dyadic dy_main,
                            ivit_dyadic dy_res, const int16_t *residual, int16_t *out, int64_t M, float ln_s, const float *ln_bias_int,
                            const float *ln_sc, const ivit_dyadic *ln_dy) {
    MlpArgs a;
    a.ln_s = ln_s; a.ln_bias_int = ln_bias_int; a.ln_sc = ln_sc; a.ln_dy = ln_dy;
    a.x = x; a.w1f = p->w1f; a.w2f = p->w2f; a.b1 = p->fc1->bias_eff; a.b2 = p->fc2->bias_eff;
    a.cq1 = p->fc1->cq; a.cq2 = p->fc2->cq; a.tab = gelu_table; a.residual = residual; a.out = out;
    a.cm = dy_main.m * dy_main.r; a.cr = dy_res.m * dy_res.r; a.M = M; a.trace = nullptr;
    if (!(fabs(a.cm) < RQ_FAST_CLIM && fabs(a.cr) < RQ_FAST_CLIM)) {
        snprintf(h->err, sizeof(h->err), "%s: residual multipliers out of the fast range", __func__);
        return IVIT_ERR_UNSUPPORTED;
    }
    REQUIRE(h, h->device == p->device, "plan and handle live on different devices");
    // one workgroup per CU.  64-token units round-robin unless cutting contiguous tile ranges into units of <= 5 tiles
    // saves a whole round (a unit costs a pass over both weight matrices whatever its size)
    const long long ntiles = (M + 15) / 16, nunits = (ntiles + MLP_TT - 2) / (MLP_TT - 1);
    const unsigned grid = (unsigned)(nunits < persistent_cus(h) ? nunits : persistent_cus(h));
    const long long rounds_fixed = (nunits + grid - 1) / grid, rounds_bal = (ntiles + (long long)MLP_TT * grid - 1) / ((long long)MLP_TT * grid);
    a.balanced = rounds_bal < rounds_fixed;
    // two kernels, the same integers.  The role-split one (producer waves on fc1 of unit u + 1 beside consumer waves on
    // ShiftGELU / fc2 / epilogue of unit u) needs a second unit per workgroup to overlap anything: with one unit per CU it only
    // ties with the lock-step kernel (45.0 vs 44.6 us at M = 20480), from two units on it wins (profiles/README.md, round 5)
    const bool role_split = p->kernel == 2 || (p->kernel == 0 && IVIT_OPT_MLP_RS && nunits > (long long)grid);
    if (ln_dy && !role_split) {
        snprintf(h->err, sizeof(h->err), "%s: the LayerNorm prologue exists in the role-split kernel only (two units per CU or more)", __func__);
        return IVIT_ERR_UNSUPPORTED;
    }
    if (role_split) {
        a.w1f = p->w1r; a.w2f = p->w2r;
        if (ln_dy) {
            if (p->fma) mlp384rs_kernel<true, IVIT_OPT_MLP_LNH><<<grid, RS_THREADS, RS_SMEM, h->stream>>>(a);
            else mlp384rs_kernel<false, IVIT_OPT_MLP_LNH><<<grid, RS_THREADS, RS_SMEM, h->stream>>>(a);
        } else if (p->fma) mlp384rs_kernel<true><<<grid, RS_THREADS, RS_SMEM, h->stream>>>(a);
        else mlp384rs_kernel<false><<<grid, RS_THREADS, RS_SMEM, h->stream>>>(a);
    } else if (p->fma) mlp384_kernel<true><<<grid, MLP_THREADS, MLP_SMEM, h->stream>>>(a);
    else mlp384_kernel<false><<<grid, MLP_THREADS, MLP_SMEM, h->stream>>>(a);
    LAUNCH_CHECK(h);
    return IVIT_OK;
}

int ivit_mlp_fused_planned(ivit_handle h, ivit_mlp_plan p, const int8_t *x, const int8_t *gelu_table, ivit_dyadic dy_main,
                           ivit_dyadic dy_res, const int16_t *residual, int16_t *out, int64_t M) {
    CHECK_H(h);
    REQUIRE(h, p && x && gelu_table && residual && out && M > 0, "bad arguments");
    return mlp_fused_launch(h, p, x, gelu_table, dy_main, dy_res, residual, out, M, 0.f, nullptr, nullptr, nullptr);
}

int ivit_layernorm_mlp_fused_planned(ivit_handle h, ivit_mlp_plan p, const int16_t *x16, float scale, const float *bias_int, const float *sc,
                                     const ivit_dyadic *ln_dy, int8_t *scratch8, const int8_t *gelu_table, ivit_dyadic dy_main,
                                     ivit_dyadic dy_res, int16_t *out, int64_t M) {
    CHECK_H(h);
    REQUIRE(h, p && x16 && bias_int && sc && ln_dy && scratch8 && gelu_table && out && M > 0, "bad arguments");
    return mlp_fused_launch(h, p, scratch8, gelu_table, dy_main, dy_res, x16, out, M, scale, bias_int, sc, ln_dy);
}

}  // extern "C"

template <int NB, bool FAST, int TT = 0, int LUT = 0, bool VROW = false>
static int launch_attn2(ivit_handle h, const AttnArgs &a, int BH) {
    const size_t lds = AttCfg<NB>::SMEM + (LUT == 2 ? (size_t)ATT_ROWLINE_BYTES
                                                   : (LUT == 1 ? (size_t)((a.t_count + 3) & ~3) * 4 + (size_t)a.nc * 512 + 256 : 0));
    if (lds > 65536) {
        // once per device and instantiation for the largest size seen (the attribute is a maximum): since round 6 every launch of
        // the row-table form is above 64 KB, and the eager single-stream mode pays host calls
        static std::atomic<int> set_dev[IVIT_MAX_DEVICES];
        const bool cached = h->device >= 0 && h->device < IVIT_MAX_DEVICES;
        if (!cached || set_dev[h->device].load(std::memory_order_acquire) < (int)lds) {
            hipError_t e = hipFuncSetAttribute((const void *)attn_fused_kernel<NB, FAST, TT, LUT, VROW>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) { snprintf(h->err, sizeof(h->err), "attn attr: %s", hipGetErrorString(e)); return IVIT_ERR_HIP; }
            if (cached) set_dev[h->device].store((int)lds, std::memory_order_release);
        }
    }
    attn_fused_kernel<NB, FAST, TT, LUT, VROW><<<BH, ATT_WAVES * 64, lds, h->stream>>>(a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(h->err, sizeof(h->err), "attn launch: %s", hipGetErrorString(e)); return IVIT_ERR_HIP; }
    return IVIT_OK;
}

template <int NB>
static int launch_attn(ivit_handle h, const AttnArgs &a, int BH) {
    const double cq = a.dy_qk.m * a.dy_qk.r, cp = a.dy_pv.m * a.dy_pv.r;
    const bool fast = (cq < 512.0 && cq > -512.0 && cp < 512.0 && cp > -512.0);
    constexpr bool dyn_t = (IVIT_OPT_ATTN_GENERIC & 1) != 0, no_lut = (IVIT_OPT_ATTN_GENERIC & 2) != 0;
    const bool lut = a.aq && a.et && a.cls && !no_lut;
    if (a.rowtab) {     // row-line tables (ivit_attention_fused_rowlut): the multipliers were checked by the caller
        if (a.ldv == 0) {   // v row-major
            if (NB == 4 && a.T == 197 && !dyn_t) return launch_attn2<NB, true, 197, 2, true>(h, a, BH);
            if (NB == 10 && a.T == 577 && !dyn_t) return launch_attn2<NB, true, 577, 2, true>(h, a, BH);
            return launch_attn2<NB, true, 0, 2, true>(h, a, BH);
        }
        if (NB == 4 && a.T == 197 && !dyn_t) return launch_attn2<NB, true, 197, 2>(h, a, BH);
        if (NB == 10 && a.T == 577 && !dyn_t) return launch_attn2<NB, true, 577, 2>(h, a, BH);
        return launch_attn2<NB, true, 0, 2>(h, a, BH);
    }
    if (fast && !dyn_t) {
        if (NB == 4 && a.T == 197) return lut ? launch_attn2<NB, true, 197, true>(h, a, BH) : launch_attn2<NB, true, 197>(h, a, BH);
        if (NB == 10 && a.T == 577) return lut ? launch_attn2<NB, true, 577, true>(h, a, BH) : launch_attn2<NB, true, 577>(h, a, BH);
    }
    if (fast && lut) return launch_attn2<NB, true, 0, true>(h, a, BH);
    return fast ? launch_attn2<NB, true>(h, a, BH) : launch_attn2<NB, false>(h, a, BH);
}

static int attention_fused_impl(ivit_handle h, const int8_t *q, const int8_t *k, const int8_t *vt, ivit_dyadic dy_qk,
                                float s_softmax, ivit_dyadic dy_pv, int8_t *ctx8, int B, int H, int T, int dh, int ldv,
                                const uint16_t *aq, const float *et, const uint8_t *cls, int nc, int t_count, int dmin,
                                const float *rowtab = nullptr) {
    CHECK_H(h);
    REQUIRE(h, q && k && vt && ctx8 && B > 0 && H > 0 && T > 0 && s_softmax > 0.f, "bad arguments");
    REQUIRE(h, ((ldv % 16) == 0 && ldv >= T) || (ldv == 0 && rowtab), "ldv must be a multiple of 16 and >= T (0 = v row-major: ivit_attention_fused_rowlut only)");
    if (dh != 64 || T > 640) {
        snprintf(h->err, sizeof(h->err), "ivit_attention_fused: built for dh == 64, T <= 640");
        return IVIT_ERR_UNSUPPORTED;
    }
    if (aq) REQUIRE(h, et && cls && nc >= 1 && nc <= 64 && t_count >= 1 && t_count <= 16384 && dmin <= 0 && dmin >= -255,
                    "bad Shiftmax tables");
    if (aq) REQUIRE(h, (((uintptr_t)aq | (uintptr_t)et) & 15) == 0 && ((uintptr_t)cls & 3) == 0,
                    "Shiftmax tables: exp_aq and exp_t must be 16-byte aligned, exp_cls 4-byte aligned (copied in 16-byte pieces)");
    AttnArgs a;
    a.q = q; a.k = k; a.vt = vt; a.ctx = ctx8; a.T = T; a.H = H; a.ldv = ldv;
    a.s_softmax = s_softmax; a.dy_qk = dy_qk; a.dy_pv = dy_pv;
    a.aq = aq; a.et = et; a.cls = cls; a.nc = nc; a.t_count = t_count; a.dmin = dmin; a.rowtab = rowtab;
    if (T <= 64) return launch_attn<1>(h, a, B * H);
    if (T <= 256) return launch_attn<4>(h, a, B * H);
    return launch_attn<10>(h, a, B * H);
}

extern "C" int ivit_attention_fused(ivit_handle h, const int8_t *q, const int8_t *k, const int8_t *vt,
                                    ivit_dyadic dy_qk, float s_softmax, ivit_dyadic dy_pv, int8_t *ctx8, int B,
                                    int H, int T, int dh, int ldv) {
    return attention_fused_impl(h, q, k, vt, dy_qk, s_softmax, dy_pv, ctx8, B, H, T, dh, ldv, nullptr, nullptr, nullptr, 0, 0, 0);
}

extern "C" int ivit_attention_fused_lut(ivit_handle h, const int8_t *q, const int8_t *k, const int8_t *vt,
                                        ivit_dyadic dy_qk, float s_softmax, const uint16_t *exp_aq, const float *exp_t,
                                        const uint8_t *exp_cls, int nclass, int t_count, int dmin, ivit_dyadic dy_pv,
                                        int8_t *ctx8, int B, int H, int T, int dh, int ldv) {
    if (h && !(exp_aq && exp_t && exp_cls)) { snprintf(h->err, sizeof(h->err), "ivit_attention_fused_lut: null table"); return IVIT_ERR_INVALID; }
    return attention_fused_impl(h, q, k, vt, dy_qk, s_softmax, dy_pv, ctx8, B, H, T, dh, ldv, exp_aq, exp_t, exp_cls, nclass, t_count, dmin);
}

// rowtab[vmax + 128][dd] = exp_t[exp_aq[exp_cls[vmax]][v] + dd] for the score v = vmax + dmin + dd (dd = 0: every score at or
// below vmax + dmin, the floor constant of the table whatever the class); one thread per entry
__global__ __launch_bounds__(64) void shiftmax_rowtable_kernel(const uint16_t *__restrict__ aq, const float *__restrict__ et,
                                                               const uint8_t *__restrict__ cls, int dmin, float *__restrict__ rowtab) {
    const int q = blockIdx.x, dd = threadIdx.x, R = 1 - dmin;
    int vi = q + dmin + dd;
    float val = 0.f;
    if (dd < R && (vi >= 0 || dd == 0)) {
        vi = vi < 0 ? 0 : vi;
        val = et[(int)aq[(int)cls[q] * 256 + vi] + dd];
    }
    rowtab[q * 64 + dd] = val;
}

extern "C" int ivit_shiftmax_rowtable(ivit_handle h, const uint16_t *exp_aq, const float *exp_t, const uint8_t *exp_cls,
                                      int nclass, int t_count, int dmin, float *rowtab) {
    CHECK_H(h);
    REQUIRE(h, exp_aq && exp_t && exp_cls && rowtab && nclass >= 1 && nclass <= 64 && t_count >= 1 && t_count <= 16384 &&
                   dmin <= 0 && dmin >= -255, "bad Shiftmax tables");
    if (1 - dmin > 64) {
        snprintf(h->err, sizeof(h->err), "%s: a table line holds 64 entries, this scale needs %d (use ivit_attention_fused_lut)", __func__, 1 - dmin);
        return IVIT_ERR_UNSUPPORTED;
    }
    REQUIRE(h, ((uintptr_t)rowtab & 15) == 0, "rowtab must be 16-byte aligned");
    shiftmax_rowtable_kernel<<<256, 64, 0, h->stream>>>(exp_aq, exp_t, exp_cls, dmin, rowtab);
    LAUNCH_CHECK(h);
    return IVIT_OK;
}

extern "C" int ivit_attention_fused_rowlut(ivit_handle h, const int8_t *q, const int8_t *k, const int8_t *vt,
                                           ivit_dyadic dy_qk, float s_softmax, const float *rowtab, int dmin, ivit_dyadic dy_pv,
                                           int8_t *ctx8, int B, int H, int T, int dh, int ldv) {
    if (!h) return IVIT_ERR_INVALID;
    REQUIRE(h, rowtab && ((uintptr_t)rowtab & 15) == 0 && dmin <= 0 && dmin >= -63, "bad row table (16-byte aligned, 1 - dmin <= 64)");
    const double cq = dy_qk.m * dy_qk.r, cp = dy_pv.m * dy_pv.r;
    if (!(cq < 512.0 && cq > -512.0 && cp < 512.0 && cp > -512.0)) {
        snprintf(h->err, sizeof(h->err), "%s: requant multipliers out of the fast range (use ivit_attention_fused_lut)", __func__);
        return IVIT_ERR_UNSUPPORTED;
    }
    return attention_fused_impl(h, q, k, vt, dy_qk, s_softmax, dy_pv, ctx8, B, H, T, dh, ldv, nullptr, nullptr, nullptr, 0, 0, dmin, rowtab);
}

// ---------------------------------------------------------------- requant
template <typename ZT>
static int requant_any(ivit_handle h, const ZT *z, const ivit_dyadic *dy, int nch, const int32_t *z_id,
                       const ivit_dyadic *dy_id, int bits, void *out, int64_t rows, int C) {
    const long long total = (long long)rows * C;
    if ((C % 8) == 0 && nch == 1) {   // per-channel tables keep the element-per-lane form (coalesced constant loads)
        const int g8 = grid_for(h, total / 8, 256 * 2);
        if (bits == 8) requant_vec8_kernel<ZT, 8><<<g8, 256, 0, h->stream>>>(z, dy, nch, z_id, dy_id, out, total / 8, C / 8);
        else if (bits == 16) requant_vec8_kernel<ZT, 16><<<g8, 256, 0, h->stream>>>(z, dy, nch, z_id, dy_id, out, total / 8, C / 8);
        else requant_vec8_kernel<ZT, 32><<<g8, 256, 0, h->stream>>>(z, dy, nch, z_id, dy_id, out, total / 8, C / 8);
        LAUNCH_CHECK(h);
        return IVIT_OK;
    }
    const int g = grid_for(h, total, 256 * 4);
    if (bits == 8) requant_kernel<ZT, 8><<<g, 256, 0, h->stream>>>(z, dy, nch, z_id, dy_id, out, total, C);
    else if (bits == 16) requant_kernel<ZT, 16><<<g, 256, 0, h->stream>>>(z, dy, nch, z_id, dy_id, out, total, C);
    else requant_kernel<ZT, 32><<<g, 256, 0, h->stream>>>(z, dy, nch, z_id, dy_id, out, total, C);
    LAUNCH_CHECK(h);
    return IVIT_OK;
}

extern "C" {

int ivit_requant_i32(ivit_handle h, const int32_t *z, const ivit_dyadic *dy, int nch, const int32_t *z_id,
                     const ivit_dyadic *dy_id, int bits, void *out, int64_t rows, int C) {
    CHECK_H(h);
    REQUIRE(h, z && dy && out && rows > 0 && C > 0, "bad arguments");
    REQUIRE(h, nch == 1 || nch == C, "nch must be 1 or C");
    REQUIRE(h, bits == 8 || bits == 16 || bits == 32, "bits must be 8, 16 or 32");
    REQUIRE(h, (z_id == nullptr) == (dy_id == nullptr), "z_id and dy_id go together");
    return requant_any<int32_t>(h, z, dy, nch, z_id, dy_id, bits, out, rows, C);
}

int ivit_requant_i16(ivit_handle h, const int16_t *z, const ivit_dyadic *dy, int nch, const int32_t *z_id,
                     const ivit_dyadic *dy_id, int bits, void *out, int64_t rows, int C) {
    CHECK_H(h);
    REQUIRE(h, z && dy && out && rows > 0 && C > 0, "bad arguments");
    REQUIRE(h, nch == 1 || nch == C, "nch must be 1 or C");
    REQUIRE(h, bits == 8 || bits == 16 || bits == 32, "bits must be 8, 16 or 32");
    REQUIRE(h, (z_id == nullptr) == (dy_id == nullptr), "z_id and dy_id go together");
    return requant_any<int16_t>(h, z, dy, nch, z_id, dy_id, bits, out, rows, C);
}

int ivit_requant_f32(ivit_handle h, const float *z, const ivit_dyadic *dy, int nch, const int32_t *z_id,
                     const ivit_dyadic *dy_id, int bits, void *out, int64_t rows, int C) {
    CHECK_H(h);
    REQUIRE(h, z && dy && out && rows > 0 && C > 0, "bad arguments");
    REQUIRE(h, nch == 1 || nch == C, "nch must be 1 or C");
    REQUIRE(h, bits == 8 || bits == 16 || bits == 32, "bits must be 8, 16 or 32");
    REQUIRE(h, (z_id == nullptr) == (dy_id == nullptr), "z_id and dy_id go together");
    return requant_any<float>(h, z, dy, nch, z_id, dy_id, bits, out, rows, C);
}

// ---------------------------------------------------------------- shift/norm kernels
static int set_dyn_lds(ivit_handle h, const void *fn, size_t bytes) {
    if (bytes > 65536) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) {
            snprintf(h->err, sizeof(h->err), "hipFuncSetAttribute: %s", hipGetErrorString(e));
            return IVIT_ERR_HIP;
        }
    }
    return IVIT_OK;
}

int ivit_shiftmax(ivit_handle h, const int8_t *x, int64_t rows, int n, int ld_in, float scale, int out_bits,
                  uint16_t *out, int ld_out) {
    CHECK_H(h);
    REQUIRE(h, x && out && rows > 0 && n > 0 && ld_in >= n && ld_out >= n && scale > 0.f, "bad arguments");
    REQUIRE(h, out_bits == 8 || out_bits == 16, "out_bits must be 8 or 16");
    const size_t lds = (size_t)4 * n * sizeof(float);
    REQUIRE(h, lds <= 160 * 1024, "row too long for LDS staging");
    int st = set_dyn_lds(h, (const void *)shiftmax_kernel, lds);
    if (st) return st;
    shiftmax_kernel<<<(unsigned)((rows + 3) / 4), 256, lds, h->stream>>>(x, rows, n, ld_in, scale, out_bits, out, ld_out);
    LAUNCH_CHECK(h);
    return IVIT_OK;
}

int ivit_shiftgelu(ivit_handle h, const int8_t *x, int64_t rows, int C, float scale, int16_t *out16) {
    CHECK_H(h);
    REQUIRE(h, x && out16 && rows > 0 && C > 0 && scale > 0.f, "bad arguments");
    REQUIRE(h, (C % 16) == 0, "C must be a multiple of 16");
    ivit_dyadic d = {0.0, 0.0};
    shiftgelu_kernel<false><<<(unsigned)((rows + 3) / 4), 256, 0, h->stream>>>(x, rows, C, scale, d, out16);
    LAUNCH_CHECK(h);
    return IVIT_OK;
}

int ivit_shiftgelu_requant(ivit_handle h, const int8_t *x, int64_t rows, int C, float scale, ivit_dyadic dy,
                           int8_t *out8) {
    CHECK_H(h);
    REQUIRE(h, x && out8 && rows > 0 && C > 0 && scale > 0.f, "bad arguments");
    REQUIRE(h, (C % 16) == 0, "C must be a multiple of 16");
    shiftgelu_kernel<true><<<(unsigned)((rows + 3) / 4), 256, 0, h->stream>>>(x, rows, C, scale, dy, out8);
    LAUNCH_CHECK(h);
    return IVIT_OK;
}

int ivit_shiftgelu_build_table(ivit_handle h, float scale, ivit_dyadic dy, int8_t *table) {
    CHECK_H(h);
    REQUIRE(h, table && scale > 0.f, "bad arguments");
    shiftgelu_table_kernel<<<256, 256, 0, h->stream>>>(scale, dy, table);
    LAUNCH_CHECK(h);
    return IVIT_OK;
}

int ivit_shiftgelu_requant_lut(ivit_handle h, const int8_t *x, int64_t rows, int C, const int8_t *table,
                               int8_t *out8) {
    CHECK_H(h);
    REQUIRE(h, x && out8 && table && rows > 0 && C > 0, "bad arguments");
    REQUIRE(h, (C % 16) == 0, "C must be a multiple of 16");
    // half a wavefront per row, row in registers: ITER = ceil(C / 512) chunks of 16 bytes per lane
    constexpr bool lut_old = false;
    const int iter = (C / 16 + 31) / 32;
    const unsigned grid2 = (unsigned)((rows + 7) / 8);
    if (lut_old || iter > 6) shiftgelu_lut_kernel<<<(unsigned)((rows + 3) / 4), 256, 0, h->stream>>>(x, rows, C, table, out8);
    else if (iter == 1) shiftgelu_lut2_kernel<1><<<grid2, 256, 0, h->stream>>>(x, rows, C, table, out8);
    else if (iter == 2) shiftgelu_lut2_kernel<2><<<grid2, 256, 0, h->stream>>>(x, rows, C, table, out8);
    else if (iter == 3) shiftgelu_lut2_kernel<3><<<grid2, 256, 0, h->stream>>>(x, rows, C, table, out8);
    else if (iter == 4) shiftgelu_lut2_kernel<4><<<grid2, 256, 0, h->stream>>>(x, rows, C, table, out8);
    else shiftgelu_lut2_kernel<6><<<grid2, 256, 0, h->stream>>>(x, rows, C, table, out8);
    LAUNCH_CHECK(h);
    return IVIT_OK;
}

int ivit_layernorm(ivit_handle h, const int16_t *x, int64_t rows, int C, float scale, const float *bias_int,
                   const float *sc, float *z) {
    CHECK_H(h);
    REQUIRE(h, x && bias_int && sc && z && rows > 0 && C > 0 && scale > 0.f, "bad arguments");
    REQUIRE(h, (C % 8) == 0, "C must be a multiple of 8");
    const size_t lds = (size_t)8 * C * sizeof(float);
    REQUIRE(h, lds <= 160 * 1024, "C too large for LDS staging");
    int st = set_dyn_lds(h, (const void *)layernorm_kernel<false>, lds);
    if (st) return st;
    layernorm_kernel<false><<<(unsigned)((rows + 7) / 8), 256, lds, h->stream>>>(x, rows, C, C, scale, bias_int, sc, nullptr, z);
    LAUNCH_CHECK(h);
    return IVIT_OK;
}

int ivit_layernorm_requant(ivit_handle h, const int16_t *x, int64_t rows, int C, int64_t row_stride, float scale,
                           const float *bias_int, const float *sc, const ivit_dyadic *dy_ch, int8_t *out8) {
    CHECK_H(h);
    REQUIRE(h, x && bias_int && sc && dy_ch && out8 && rows > 0 && C > 0 && scale > 0.f, "bad arguments");
    REQUIRE(h, (C % 8) == 0 && (row_stride % 8) == 0 && row_stride >= C, "C, row_stride multiples of 8");
    {   // production form: the row in registers, 4 S lanes per row (ivit_layernorm.h)
#define LNR_LAUNCH(CC, S)                                                                                      \
        do {                                                                                                    \
            constexpr int rpb = (LNR_THREADS(S) / 64) * (64 / (4 * S));                                         \
            layernorm_reg_kernel<CC, S><<<(unsigned)((rows + rpb - 1) / rpb), LNR_THREADS(S), 0, h->stream>>>( \
                x, rows, row_stride, scale, bias_int, sc, dy_ch, out8);                                         \
            LAUNCH_CHECK(h);                                                                                    \
            return IVIT_OK;                                                                                     \
        } while (0)
        switch (C) {
            // lanes per row = 4 S (measured on DeiT-S b256: S = 1 23.2 us, S = 2 20.1, S = 4 20.2 — shorter per-wave instruction
            // chains and more waves per SIMD beat the cheaper quad-only reduction).  S = 1 exists in probe builds only
            // (ivit_layernorm.h has its history)
            case 96: LNR_LAUNCH(96, 2);        // Swin-T/S stage 0 (token-order sums use their own kernel)
            case 128: LNR_LAUNCH(128, 2);      // Swin-B stage 0
#if IVIT_PROBE_LN192_S1 == 2
            case 192: break;                   // probe: C = 192 on layernorm16_kernel, the round-2 LayerNorm
#elif IVIT_PROBE_LN192_S1
            case 192: LNR_LAUNCH(192, 1);      // probe builds only (tools/ln_s1_probe.sh)
#else
            case 192: LNR_LAUNCH(192, 2);      // DeiT-T, Swin stage 1
#endif
            case 256: LNR_LAUNCH(256, 2);
            case 384: LNR_LAUNCH(384, 2);      // DeiT-S, Swin stage 2, PatchMerging
            case 512: LNR_LAUNCH(512, 4);
            case 768: LNR_LAUNCH(768, 4);      // DeiT-B / ViT-B, Swin stage 3
            case 1024: LNR_LAUNCH(1024, 4);    // ViT-L
            case 1536: LNR_LAUNCH(1536, 4);    // Swin PatchMerging before stage 3
            default: break;
        }
#undef LNR_LAUNCH
    }
    {   // other channel counts: 16 lanes per row, the row staged in LDS
        const size_t lds16 = (size_t)16 * (C + 16) * 4 + (size_t)C * 20;
        if (lds16 <= 150 * 1024) {
            long long riter = rows / (16LL * 6 * h->num_cu);
            riter = riter < 1 ? 1 : (riter > LN_RITER ? LN_RITER : riter);
            const long long per_block = 16 * riter;
            const unsigned grid = (unsigned)((rows + per_block - 1) / per_block);
            int st16 = set_dyn_lds(h, (const void *)layernorm16_kernel<0>, lds16);
            if (st16) return st16;
            layernorm16_kernel<0><<<grid, 256, lds16, h->stream>>>(x, rows, C, row_stride, scale, bias_int, sc, dy_ch, out8, (int)riter);
            LAUNCH_CHECK(h);
            return IVIT_OK;
        }
    }
    const size_t lds = (size_t)8 * C * sizeof(float);
    REQUIRE(h, lds <= 160 * 1024, "C too large for LDS staging");
    int st = set_dyn_lds(h, (const void *)layernorm_kernel<true>, lds);
    if (st) return st;
    layernorm_kernel<true><<<(unsigned)((rows + 7) / 8), 256, lds, h->stream>>>(x, rows, C, row_stride, scale, bias_int, sc, dy_ch, out8);
    LAUNCH_CHECK(h);
    return IVIT_OK;
}

// PatchMerging's gather folded into the LayerNorm that follows it (swin_quant.py:336-349: the 2 x 2 gather, norm over 4C, qact1):
// x int16 [B, R, R, C] -> out8 int8 [B (R/2)^2, 4C].  Same integers as ivit_patch_merge_gather + ivit_layernorm_requant.
int ivit_patch_merge_layernorm_requant(ivit_handle h, const int16_t *x, int B, int R, int C, float scale, const float *bias_int,
                                       const float *sc, const ivit_dyadic *dy_ch, int8_t *out8) {
    CHECK_H(h);
    REQUIRE(h, x && bias_int && sc && dy_ch && out8 && B > 0 && R > 0 && (R % 2) == 0 && C > 0 && scale > 0.f, "bad arguments");
    const long long rows = (long long)B * (R / 2) * (R / 2);
#define LNM_LAUNCH(CC, S)                                                                                                \
    do {                                                                                                                 \
        constexpr int rpb = (LNR_THREADS(S) / 64) * (64 / (4 * S));                                                      \
        layernorm_reg_kernel<CC, S, true><<<(unsigned)((rows + rpb - 1) / rpb), LNR_THREADS(S), 0, h->stream>>>(        \
            x, rows, 0, scale, bias_int, sc, dy_ch, out8, R);                                                            \
        LAUNCH_CHECK(h);                                                                                                 \
        return IVIT_OK;                                                                                                  \
    } while (0)
    switch (4 * C) {
        case 384: LNM_LAUNCH(384, 2);       // Swin-T/S: 96 -> 384
        case 512: LNM_LAUNCH(512, 4);       // Swin-B: 128 -> 512
        case 768: LNM_LAUNCH(768, 4);
        case 1024: LNM_LAUNCH(1024, 4);
        case 1536: LNM_LAUNCH(1536, 4);
        default: break;
    }
#undef LNM_LAUNCH
    snprintf(h->err, sizeof(h->err), "%s: built for C = 96, 128, 192, 256, 384 (use ivit_patch_merge_gather + ivit_layernorm_requant)", __func__);
    return IVIT_ERR_UNSUPPORTED;
}

// ---------------------------------------------------------------- Swin-specific operators
int ivit_shiftmax_masked(ivit_handle h, const int8_t *x, int64_t rows, int n, int ld_in, float scale, int out_bits,
                         const float *mask, int nW, int H, uint16_t *out, int ld_out) {
    CHECK_H(h);
    REQUIRE(h, x && out && rows > 0 && n > 0 && ld_in >= n && ld_out >= n && scale > 0.f, "bad arguments");
    REQUIRE(h, out_bits == 8 || out_bits == 16, "out_bits must be 8 or 16");
    REQUIRE(h, mask == nullptr || (nW > 0 && H > 0), "mask needs nW, H");
    const size_t lds = (size_t)4 * n * sizeof(float);
    REQUIRE(h, lds <= 160 * 1024, "row too long for LDS staging");
    int st = set_dyn_lds(h, (const void *)shiftmax_masked_kernel, lds);
    if (st) return st;
    shiftmax_masked_kernel<<<(unsigned)((rows + 3) / 4), 256, lds, h->stream>>>(x, rows, n, ld_in, scale, out_bits, mask,
                                                                          nW > 0 ? nW : 1, H > 0 ? H : 1, out, ld_out);
    LAUNCH_CHECK(h);
    return IVIT_OK;
}

int ivit_requant_i32_bcast(ivit_handle h, const int32_t *z, ivit_dyadic dy, const int32_t *z_id, int64_t id_period,
                           ivit_dyadic dy_id, int bits, void *out, int64_t total) {
    CHECK_H(h);
    REQUIRE(h, z && z_id && out && total > 0 && id_period > 0, "bad arguments");
    REQUIRE(h, bits == 8 || bits == 16, "bits must be 8 or 16");
    const int g = grid_for(h, total, 256 * 4);
    if (bits == 8) requant_bcast_kernel<8><<<g, 256, 0, h->stream>>>(z, dy, z_id, id_period, dy_id, out, total);
    else requant_bcast_kernel<16><<<g, 256, 0, h->stream>>>(z, dy, z_id, id_period, dy_id, out, total);
    LAUNCH_CHECK(h);
    return IVIT_OK;
}

int ivit_avgpool_requant(ivit_handle h, const int8_t *x, int B, int L, int C, ivit_dyadic dy, int8_t *out8) {
    CHECK_H(h);
    REQUIRE(h, x && out8 && B > 0 && L > 0 && C > 0, "bad arguments");
    REQUIRE(h, (L & 1) == 1, "token count must be odd (no rounding ties; see kernel comment)");
    avgpool_requant_kernel<<<(unsigned)((B * C + 255) / 256), 256, 0, h->stream>>>(x, B, L, C, dy, out8);
    LAUNCH_CHECK(h);
    return IVIT_OK;
}

int ivit_layernorm_tokenorder_requant(ivit_handle h, const int16_t *x, int64_t rows, int C, float scale,
                                      const float *bias_int, const float *sc, const ivit_dyadic *dy,
                                      int tokens_per_image, int8_t *out8) {
    CHECK_H(h);
    REQUIRE(h, x && bias_int && sc && dy && out8 && rows > 0 && C > 0 && scale > 0.f && tokens_per_image > 0, "bad arguments");
    const size_t lds = ((size_t)LNT_ROWS * (C + 1) + 3 * (size_t)C + 2 * LNT_ROWS + 2) * sizeof(float) + (size_t)C * sizeof(double);
    REQUIRE(h, lds <= 160 * 1024, "C too large for LDS staging");
    if (C == 96 || C == 128) {          // Swin-T/S and Swin-B stage 0
        const size_t lds8 = ((size_t)LNT8_ROWS * (C + 1) + C + 2 * LNT8_ROWS) * sizeof(float);
        const unsigned g8 = (unsigned)((rows + LNT8_ROWS - 1) / LNT8_ROWS);
        REQUIRE(h, (((uintptr_t)x | (uintptr_t)out8 | (uintptr_t)bias_int | (uintptr_t)sc) & 15) == 0,
                "C = 96 / 128: x, out, bias_int, sc must be 16-byte aligned (vector loads)");
        const void *fn = C == 96 ? (const void *)layernorm_tokenorder8_kernel<1, 96, int16_t> : (const void *)layernorm_tokenorder8_kernel<1, 128, int16_t>;
        int st = set_dyn_lds(h, fn, lds8);
        if (st) return st;
        if (C == 96) layernorm_tokenorder8_kernel<1, 96, int16_t><<<g8, 96 / 8 * 16, lds8, h->stream>>>(x, rows, scale, bias_int, sc, dy, tokens_per_image, out8, ivit_dyadic{0.0, 0.0});
        else layernorm_tokenorder8_kernel<1, 128, int16_t><<<g8, 128 / 8 * 16, lds8, h->stream>>>(x, rows, scale, bias_int, sc, dy, tokens_per_image, out8, ivit_dyadic{0.0, 0.0});
    } else {
        int st = set_dyn_lds(h, (const void *)layernorm_tokenorder_kernel<1, 0>, lds);
        if (st) return st;
        layernorm_tokenorder_kernel<1, 0><<<(unsigned)((rows + LNT_ROWS - 1) / LNT_ROWS), 256, lds, h->stream>>>(
            x, rows, C, scale, bias_int, sc, dy, tokens_per_image, out8);
    }
    LAUNCH_CHECK(h);
    return IVIT_OK;
}

int ivit_patch_norm_tokenorder(ivit_handle h, const int8_t *x8, int64_t rows, int C, float scale, const float *bias_int,
                               const float *sc, const ivit_dyadic *dy_ch, ivit_dyadic dy2, int tokens_per_image,
                               int16_t *out16) {
    CHECK_H(h);
    REQUIRE(h, x8 && bias_int && sc && dy_ch && out16 && rows > 0 && C > 0 && scale > 0.f && tokens_per_image > 0, "bad arguments");
    const size_t lds = ((size_t)LNT_ROWS * (C + 1) + 3 * (size_t)C + 2 * LNT_ROWS + 2) * sizeof(float) + (size_t)C * sizeof(double);
    REQUIRE(h, lds <= 160 * 1024, "C too large for LDS staging");
    const unsigned grid = (unsigned)((rows + LNT_ROWS - 1) / LNT_ROWS);
    if (C == 96 || C == 128) {
        const size_t lds8 = ((size_t)LNT8_ROWS * (C + 1) + C + 2 * LNT8_ROWS) * sizeof(float);
        const unsigned g8 = (unsigned)((rows + LNT8_ROWS - 1) / LNT8_ROWS);
        REQUIRE(h, (((uintptr_t)x8 | (uintptr_t)out16 | (uintptr_t)bias_int | (uintptr_t)sc) & 15) == 0,
                "C = 96 / 128: x, out, bias_int, sc must be 16-byte aligned (vector loads)");
        const void *fn = C == 96 ? (const void *)layernorm_tokenorder8_kernel<2, 96, int8_t> : (const void *)layernorm_tokenorder8_kernel<2, 128, int8_t>;
        int st = set_dyn_lds(h, fn, lds8);
        if (st) return st;
        if (C == 96) layernorm_tokenorder8_kernel<2, 96, int8_t><<<g8, 96 / 8 * 16, lds8, h->stream>>>(x8, rows, scale, bias_int, sc, dy_ch, tokens_per_image, out16, dy2);
        else layernorm_tokenorder8_kernel<2, 128, int8_t><<<g8, 128 / 8 * 16, lds8, h->stream>>>(x8, rows, scale, bias_int, sc, dy_ch, tokens_per_image, out16, dy2);
    } else {
        int st = set_dyn_lds(h, (const void *)layernorm_tokenorder_kernel<2, 0, int8_t>, lds);
        if (st) return st;
        layernorm_tokenorder_kernel<2, 0, int8_t><<<grid, 256, lds, h->stream>>>(x8, rows, C, scale, bias_int, sc, dy_ch,
                                                                           tokens_per_image, out16, dy2);
    }
    LAUNCH_CHECK(h);
    return IVIT_OK;
}

int ivit_layernorm_tokenorder(ivit_handle h, const int16_t *x, int64_t rows, int C, float scale,
                              const float *bias_int, const float *sc, int tokens_per_image, float *z) {
    CHECK_H(h);
    REQUIRE(h, x && bias_int && sc && z && rows > 0 && C > 0 && scale > 0.f && tokens_per_image > 0, "bad arguments");
    const size_t lds = ((size_t)LNT_ROWS * (C + 1) + 3 * (size_t)C + 2 * LNT_ROWS + 2) * sizeof(float) + (size_t)C * sizeof(double);
    REQUIRE(h, lds <= 160 * 1024, "C too large for LDS staging");
    int st = set_dyn_lds(h, (const void *)layernorm_tokenorder_kernel<0, 0>, lds);
    if (st) return st;
    layernorm_tokenorder_kernel<0, 0><<<(unsigned)((rows + LNT_ROWS - 1) / LNT_ROWS), 256, lds, h->stream>>>(
        x, rows, C, scale, bias_int, sc, nullptr, tokens_per_image, z);
    LAUNCH_CHECK(h);
    return IVIT_OK;
}

int ivit_debug_div(ivit_handle h, const float *n, const float *d, float *q_ieee, float *q_lean, int64_t count) {
    CHECK_H(h);
    REQUIRE(h, n && d && q_ieee && q_lean && count > 0, "bad arguments");
    debug_div_kernel<<<(unsigned)((count + 255) / 256), 256, 0, h->stream>>>(n, d, q_ieee, q_lean, count);
    LAUNCH_CHECK(h);
    return IVIT_OK;
}

int ivit_debug_requotient(ivit_handle h, const float *q, const float *d, float *r_ieee, float *r_markstein, int64_t count) {
    CHECK_H(h);
    REQUIRE(h, q && d && r_ieee && r_markstein && count > 0, "bad arguments");
    debug_requotient_kernel<<<(unsigned)((count + 255) / 256), 256, 0, h->stream>>>(q, d, r_ieee, r_markstein, count);
    LAUNCH_CHECK(h);
    return IVIT_OK;
}

int ivit_im2col_patch(ivit_handle h, const int8_t *img, int B, int Cin, int H, int W, int P, int8_t *rows) {
    CHECK_H(h);
    REQUIRE(h, img && rows && B > 0 && Cin > 0 && P > 0, "bad arguments");
    REQUIRE(h, (H % P) == 0 && (W % P) == 0 && (P % 4) == 0 && (W % 4) == 0, "H,W multiples of P; P,W multiples of 4");
    const size_t lds = (size_t)Cin * P * W;
    REQUIRE(h, lds <= 64 * 1024, "patch strip too large for LDS staging");
    if (P == 16 && Cin == 3 && (W % 16) == 0 && B <= 65535)
        im2col_patch16_kernel<3><<<dim3((unsigned)(H / P), (unsigned)B), 256, lds, h->stream>>>(img, H, W, rows);
    else
        im2col_patch_kernel<<<(unsigned)(B * (H / P)), 256, lds, h->stream>>>(img, B, Cin, H, W, P, rows);
    LAUNCH_CHECK(h);
    return IVIT_OK;
}

int ivit_embed_finish(ivit_handle h, const int16_t *patch16, const int32_t *z_cls, const int16_t *pos,
                      ivit_dyadic dy_x, ivit_dyadic dy_pos, int16_t *x16, int B, int T, int D) {
    CHECK_H(h);
    REQUIRE(h, patch16 && z_cls && pos && x16 && B > 0 && T > 1 && D > 0 && (D % 8) == 0, "bad arguments (D must be a multiple of 8)");
    REQUIRE(h, (long long)T * D / 8 < (1 << 22) && B <= 65535, "T * D / 8 < 2^22, B < 2^16");
    const int fast = fabs(dy_x.m * dy_x.r) < RQ_FAST_CLIM && fabs(dy_pos.m * dy_pos.r) < RQ_FAST_CLIM;
    embed_finish_kernel<<<dim3((unsigned)((T * (D / 8) + 255) / 256), (unsigned)B), 256, 0, h->stream>>>(
        patch16, z_cls, pos, dy_x, dy_pos, x16, T, D, 1.0f / (float)(D / 8), fast);
    LAUNCH_CHECK(h);
    return IVIT_OK;
}

#ifndef IVIT_OPT_PATCH_EMBED
#define IVIT_OPT_PATCH_EMBED 1          // A/B: 0 = ivit_patch_embed always answers IVIT_ERR_UNSUPPORTED (im2col + GEMM + embed_finish launches)
#endif
int ivit_patch_embed(ivit_handle h, const int8_t *images, int B, int C, int H, int W, int P, const int8_t *w, const int32_t *bias,
                     const ivit_dyadic *dy_ch, const int32_t *z_cls, const int16_t *pos, ivit_dyadic dy_x, ivit_dyadic dy_pos,
                     int16_t *x16, int D) {
    CHECK_H(h);
    REQUIRE(h, images && w && dy_ch && z_cls && pos && x16 && B > 0 && C > 0 && H > 0 && W > 0 && P > 0 && D > 0, "bad arguments");
    const int gh = H / P, gw = W / P, np = gh * gw, K = C * P * P, T = np + 1;
    const bool fast = fabs(dy_x.m * dy_x.r) < RQ_FAST_CLIM && fabs(dy_pos.m * dy_pos.r) < RQ_FAST_CLIM;
    GemmArgs a = linear_args(images, w, bias, B * np, D, K);
    if (!IVIT_OPT_PATCH_EMBED || P != 16 || H % 16 || W % 16 || (D % 8) != 0 || (K % 64) != 0 || !fast || !use_gemm2(a) || (long long)B * np >= (1ll << 30) ||
        (long long)T * D / 8 >= (1 << 22) || B > 65535) {
        snprintf(h->err, sizeof(h->err), "%s: built for 16 x 16 patches, K %% 64 == 0, both multipliers in the fast range", __func__);
        return IVIT_ERR_UNSUPPORTED;
    }
    // class-token rows, then every patch row from the GEMM's epilogue: QuantConv2d -> QuantAct(16) -> + pos -> QuantAct(16) (vit_quant.py:255-265)
    embed_finish_kernel<<<dim3((unsigned)((D / 8 + 255) / 256), (unsigned)B), 256, 0, h->stream>>>(nullptr, z_cls, pos, dy_x, dy_pos, x16, T, D,
                                                                                                  1.0f / (float)(D / 8), 1, 1);
    LAUNCH_CHECK(h);
    a.out = x16; a.dy_ch = dy_ch; a.dy_main = dy_x; a.dy_res = dy_pos; a.residual = pos;
    a.img = images; a.img_C = C; a.img_H = H; a.img_W = W; a.pe_gw = gw; a.pe_P = np;
    a.tiles_n = (a.N + G2_BN - 1) / G2_BN;
    a.dbg = 0;
    const long long t128 = (long long)((a.M + 127) / 128) * a.tiles_n;
    gemm_glds_kernel<EPI_RQ16_CH_RES, 128, true><<<dim3((unsigned)t128), 256, 0, h->stream>>>(a);
    LAUNCH_CHECK(h);
    return IVIT_OK;
}

}  // extern "C"

extern "C" {

static int window_attention_impl(ivit_handle h, const int8_t *qkv, ivit_dyadic dy_qk, ivit_dyadic dy_a,
                                 const int16_t *relb, float s_softmax, const uint16_t *aq, const float *et, const uint8_t *cls,
                                 int nc, int t_count, int dmin, ivit_dyadic dy_pv, int8_t *ctx, int B, int R,
                                 int window, int shift, int heads, int dh) {
    CHECK_H(h);
    REQUIRE(h, qkv && relb && ctx && B > 0 && R > 0 && heads > 0, "bad arguments");
    if (window != 7 || dh != 32 || (R % 7) != 0) {
        snprintf(h->err, sizeof(h->err), "%s: built for window 7, head dim 32, R %% 7 == 0", __func__);
        return IVIT_ERR_UNSUPPORTED;
    }
    REQUIRE(h, shift >= 0 && shift < 7 && s_softmax > 0.f, "bad shift / scale");
    // magic-number rounding needs |z*c| < 2^31 with |q.k| <= 2^19 and |sum P*v| <= 2^20
    REQUIRE(h, fabs(dy_qk.m * dy_qk.r) < 2048.0 && fabs(dy_pv.m * dy_pv.r) < 1024.0, "requant factor out of range");
    WinAttnArgs a;
    memset(&a, 0, sizeof(a));
    a.qkv = qkv; a.ctx = ctx; a.relb = relb; a.B = B; a.R = R; a.shift = shift; a.heads = heads;
    a.dy_qk = dy_qk; a.dy_a = dy_a; a.dy_pv = dy_pv; a.s = s_softmax;
    a.units = (long long)B * (R / 7) * (R / 7) * heads;
    a.wpw = 1;
    const long long nwin = (long long)B * (R / 7) * (R / 7);
    if (aq) {
        REQUIRE(h, et && cls && nc >= 1 && nc <= 64 && t_count >= 1 && t_count <= 16384 && dmin <= 0 && dmin >= -255,
                "bad Shiftmax tables");
        REQUIRE(h, (((uintptr_t)aq | (uintptr_t)et) & 15) == 0 && ((uintptr_t)cls & 3) == 0,
                "Shiftmax tables: exp_aq and exp_t must be 16-byte aligned, exp_cls 4-byte aligned (copied in 16-byte pieces)");
        a.aq = aq; a.et = et; a.cls = cls; a.nc = nc; a.t_count = t_count; a.dmin = dmin;
        // windows per wavefront: as many as keep >= 3 workgroups per resident slot (2 per CU), at most 4
        const long long slots = 2LL * h->num_cu;
        int wpw = 4;
        while (wpw > 1 && ((nwin + 8LL * wpw - 1) / (8LL * wpw)) * heads < 3 * slots) wpw >>= 1;
        a.wpw = wpw;
        const size_t lds = WA_FIXED(8) + (size_t)((t_count + 3) & ~3) * 4 + (size_t)nc * 512 + 256;
        int st = set_dyn_lds(h, (const void *)window_attention_kernel<true>, lds);
        if (st) return st;
        const long long groups = (nwin + 8LL * wpw - 1) / (8LL * wpw);      // grid: whole groups of 8 window groups (see the kernel)
        window_attention_kernel<true><<<dim3((unsigned)(((groups + 7) / 8) * 8 * heads)), 512, lds, h->stream>>>(a);
    } else {
        const long long groups = (nwin + 3) / 4;
        window_attention_kernel<false><<<dim3((unsigned)(((groups + 7) / 8) * 8 * heads)), 256, WA_FIXED(4), h->stream>>>(a);
    }
    LAUNCH_CHECK(h);
    return IVIT_OK;
}

int ivit_window_attention_fused(ivit_handle h, const int8_t *qkv, ivit_dyadic dy_qk, ivit_dyadic dy_a,
                                const int16_t *relb, float s_softmax, ivit_dyadic dy_pv, int8_t *ctx, int B, int R,
                                int window, int shift, int heads, int dh) {
    return window_attention_impl(h, qkv, dy_qk, dy_a, relb, s_softmax, nullptr, nullptr, nullptr, 0, 0, 0, dy_pv, ctx, B, R,
                                 window, shift, heads, dh);
}

int ivit_window_attention_fused_lut(ivit_handle h, const int8_t *qkv, ivit_dyadic dy_qk, ivit_dyadic dy_a,
                                    const int16_t *relb, float s_softmax, const uint16_t *exp_aq, const float *exp_t,
                                    const uint8_t *exp_cls, int nclass, int t_count, int dmin, ivit_dyadic dy_pv,
                                    int8_t *ctx, int B, int R, int window, int shift, int heads, int dh) {
    if (h && !(exp_aq && exp_t && exp_cls)) { snprintf(h->err, sizeof(h->err), "ivit_window_attention_fused_lut: null table"); return IVIT_ERR_INVALID; }
    return window_attention_impl(h, qkv, dy_qk, dy_a, relb, s_softmax, exp_aq, exp_t, exp_cls, nclass, t_count, dmin, dy_pv, ctx,
                                 B, R, window, shift, heads, dh);
}

int ivit_mlp_fused(ivit_handle h, const int8_t *x, const int8_t *w1, const int32_t *b1, const ivit_dyadic *dy1,
                   const int8_t *gelu_table, const int8_t *w2, const int32_t *b2, const ivit_dyadic *dy2,
                   ivit_dyadic dy_main, ivit_dyadic dy_res, const int16_t *residual, int16_t *out, int64_t M, int C,
                   int hidden) {
    CHECK_H(h);
    REQUIRE(h, x && w1 && dy1 && gelu_table && w2 && dy2 && residual && out && M > 0, "bad arguments");
    if (C != MF_C || hidden != MF_HD) {
        snprintf(h->err, sizeof(h->err), "%s: built for C = 96, hidden = 384 (weights resident in LDS)", __func__);
        return IVIT_ERR_UNSUPPORTED;
    }
    // the dynamic-LDS attribute is per device (one process-wide flag left the second GPU of a process without it)
    static std::atomic<bool> attr_dev[IVIT_MAX_DEVICES];
    const bool cached = h->device >= 0 && h->device < IVIT_MAX_DEVICES;
    if (!cached || !attr_dev[h->device].load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute((const void *)swin_mlp_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, MF_SMEM);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)swin_mlp_rs_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SR_SMEM);
        if (e != hipSuccess) { snprintf(h->err, sizeof(h->err), "mlp_fused attr: %s", hipGetErrorString(e)); return IVIT_ERR_HIP; }
        if (cached) attr_dev[h->device].store(true, std::memory_order_release);
    }
    MlpFusedArgs a;
    a.x = x; a.w1 = w1; a.b1 = b1; a.dy1 = dy1; a.tab = gelu_table; a.w2 = w2; a.b2 = b2; a.dy2 = dy2;
    a.dy_main = dy_main; a.dy_res = dy_res; a.residual = residual; a.out = out; a.M = M;
    const long long ntiles = (M + MF_BM - 1) / MF_BM;
    const unsigned grid = (unsigned)(ntiles < h->num_cu ? ntiles : h->num_cu);
    // role-split form (ivit_swin_mlp_rs.h: producers on fc1 of tile i + 1 beside consumers on ShiftGELU / fc2 of tile i) when a
    // workgroup has at least two tiles to overlap; the phase-by-phase kernel otherwise
    if (IVIT_OPT_SWIN_MLP_RS && ntiles >= 2 * (long long)grid) swin_mlp_rs_kernel<<<grid, SR_THREADS, SR_SMEM, h->stream>>>(a);
    else swin_mlp_fused_kernel<<<grid, MF_THREADS, MF_SMEM, h->stream>>>(a);
    LAUNCH_CHECK(h);
    return IVIT_OK;
}

int ivit_patch_merge_gather(ivit_handle h, const void *x, int in_bits, int B, int R, int C, int16_t *out) {
    CHECK_H(h);
    REQUIRE(h, x && out && B > 0 && R > 0 && (R % 2) == 0 && C > 0 && (C % 8) == 0, "bad arguments (C must be a multiple of 8)");
    REQUIRE(h, in_bits == 8 || in_bits == 16, "in_bits must be 8 or 16");
    const long long total = (long long)B * R * R * C / 8;
    if (in_bits == 8) patch_merge_gather_kernel<int8_t><<<grid_for(h, total, 1024), 256, 0, h->stream>>>((const int8_t *)x, B, R, C, out);
    else patch_merge_gather_kernel<int16_t><<<grid_for(h, total, 1024), 256, 0, h->stream>>>((const int16_t *)x, B, R, C, out);
    LAUNCH_CHECK(h);
    return IVIT_OK;
}

int ivit_widen_i8_i16(ivit_handle h, const int8_t *x, int16_t *out, int64_t n) {
    CHECK_H(h);
    REQUIRE(h, x && out && n >= 0, "bad arguments");
    if (n == 0) return IVIT_OK;
    widen_i8_i16_kernel<<<grid_for(h, n, 1024), 256, 0, h->stream>>>(x, out, n);
    LAUNCH_CHECK(h);
    return IVIT_OK;
}

}  // extern "C"

#include "ivit_model.h"
