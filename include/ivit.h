/*
 * ivit.h — C-ABI of the MI355X-native integer-only ViT inference path.
 *
 * The reference (zkkli/I-ViT) has no FFI layer: its boundary for this path is the
 * Python operator surface of models/quantization_utils/quant_modules.py (re-exported
 * by models/quantization_utils/__init__.py:1).  Each entry point below names the
 * reference interface it replaces (file:line, relative to the reference root).  The
 * Python face that mirrors the reference classes lives in i-vit_amd/quant_modules.py
 * and binds these symbols with ctypes.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in _host;
 *   - tensors are dense row-major, the channel dimension is last;
 *   - every call is asynchronous on the handle's HIP stream, allocates nothing,
 *     synchronises nothing and is hipGraph-capturable;
 *   - return value: IVIT_OK or an error code (no exceptions, no aborts);
 *   - activations are carried as INTEGERS plus an fp32 scale held by the caller
 *     (the reference carries fp32 "integer*scale" tensors; X = fl(Q*s) is
 *     re-derived inside the kernels where its rounding matters).
 *
 * Dyadic requantisation (reference quant_utils.py:150-175, 213-253):
 *   out = clamp( rne( (double(z) * m) * r ) [+ rne((double(z_id) * m_id) * r_id)] )
 *   with m = round_half_away(mant * 2^31), r = 2^-(31 - exponent) of
 *   double(s_pre)/double(float(s_out)); prepared on the host (ivit_amd.freeze).
 */
#ifndef IVIT_H
#define IVIT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ivit_ctx *ivit_handle;

typedef struct ivit_dyadic {
    double m; /* integer-valued multiplier, |m| <= 2^31 (negative after LN with w<0) */
    double r; /* 2^-e */
} ivit_dyadic;

enum {
    IVIT_OK = 0,
    IVIT_ERR_INVALID = 1,     /* bad argument (null pointer, size, alignment, bits) */
    IVIT_ERR_HIP = 2,         /* HIP runtime error, see ivit_last_error            */
    IVIT_ERR_UNSUPPORTED = 3, /* shape outside what the kernels are built for      */
    IVIT_ERR_NO_DEVICE = 4
};

/* 100 * major + minor.  104 (round 6): ivit_set_cu_share, ivit_linear_i8_requant8_store16 (additions only).
 * 103 (round 6): ivit_linear_plan_prepare_ws, ivit_layernorm_linear_i8_qkv_planned,
 * ivit_linear_i8_requant_residual_layernorm_planned, ivit_layernorm_linear_i8_requant_planned, ivit_patch_embed,
 * ivit_layernorm_mlp_fused_planned (additions only).
 * 102 (round 6): ivit_shiftmax_rowtable, ivit_attention_fused_rowlut (additions only).
 * 101 (round 5): ivit_mlp_plan_select; ivit_swin_block / ivit_vit_block carry the optional Shiftmax-table
 * fields exp_* at their END (added in 100 without a bump: a caller compiled against an older layout must be rebuilt).
 * Parameter structs are read field by field: ZERO-INITIALISE them (memset / = {0}) before filling — exp_aq == NULL (and
 * exp_nc == exp_tcount == exp_dmin == 0) selects the arithmetic Shiftmax, anything else is taken as device pointers.        */
#define IVIT_VERSION 104
int ivit_version(void);
const char *ivit_status_string(int status);

/* One handle per (device, stream).  `hip_stream` is a hipStream_t (NULL = default). */
int ivit_create(ivit_handle *out, int device, void *hip_stream);
int ivit_destroy(ivit_handle h);
int ivit_set_stream(ivit_handle h, void *hip_stream);
/* CUs the one-workgroup-per-CU kernels launched through `h` (the K = 384 QuantLinear layers on gemm_ws_qkv_kernel, the fused Mlp) size their
 * grids for; 0 (the default) = every CU of the device.  A caller that runs several handles side by side on slices of a batch — what
 * ivit_vit_forward / ivit_swin_forward do internally with nslices > 1 — gives each handle its share (CUs / slices): a slice's launch then
 * has the per-workgroup geometry of the unsliced one and the slices' kernels run on disjoint CUs (Swin-T b256, two slices: +2.0 %).
 * No reference counterpart (the reference has no launch geometry); results do not depend on it. */
int ivit_set_cu_share(ivit_handle h, int cus);
const char *ivit_last_error(ivit_handle h);

/* ---- a4  QuantAct.forward, input branch  (quant_modules.py:194-196 ->
 * quant_utils.py:77-96, 12-48):  q = clamp(rne(fl(fl(1/s)*x)), -128, 127)            */
int ivit_quantize_input_f32(ivit_handle h, const float *x, float scale, int8_t *q, int64_t n);
/* ToTensor -> Normalize(mean, std) -> the same input QuantAct, from uint8 pixels on the device
 * (utils/data_utils.py:89-91: transforms.ToTensor, transforms.Normalize; then vit_quant.py:257):
 * hwc uint8 [B, H, W, 3] (a centre-cropped image as PIL hands it over) -> nchw int8 [B, 3, H, W].
 * mean / std are HOST arrays of 3 floats.  Resize / crop: ivit_resize_center_crop_u8 below.       */
int ivit_normalize_quantize_u8(ivit_handle h, const uint8_t *hwc, int B, int H, int W, const float mean_host[3],
                               const float std_host[3], float scale, int8_t *nchw);

/* Resize(size, interpolation=bicubic) + CenterCrop(crop) of the same eval transform (utils/data_utils.py:82-88) on the
 * device: hwc uint8 [B, H0, W0, 3] -> out_hwc uint8 [B, crop, crop, 3] (feed it to ivit_normalize_quantize_u8).  The
 * shorter side is resized to `size` (the longer to int(size * long / short)), antialiased separable bicubic (a = -0.5)
 * in fp32, rne, clamp; then the centre crop.  workspace: B * H0 * crop * 3 floats (caller-owned).  The reference resizes
 * with PIL, which this image does not have: the pin is torch's antialiased bicubic (tests/golden/resize.npz).       */
int ivit_resize_center_crop_u8(ivit_handle h, const uint8_t *hwc, int B, int H0, int W0, int size, int crop,
                               float *workspace, uint8_t *out_hwc);

/* ---- a1  QuantLinear.forward  (quant_modules.py:67-97) — integer accumulators.
 * acc[i,j] = sum_k x[i,k]*w[j,k] + bias[j];  x int8 [M,K], w int8 [N,K], K % 16 == 0. */
int ivit_linear_i8(ivit_handle h, const int8_t *x, const int8_t *w, const int32_t *bias,
                   int32_t *acc, int M, int N, int K);

/* a1 + a3: the QuantLinear -> QuantAct pairs of the reference, fused.
 * out = clamp_bits(rq(acc[i,j], dy_ch[j]));  bits = 8 (int8 out) or 16 (int16 out).    */
int ivit_linear_i8_requant(ivit_handle h, const int8_t *x, const int8_t *w, const int32_t *bias,
                           const ivit_dyadic *dy_ch, int bits, void *out, int M, int N, int K);

/* a1 + a3 + a3(identity): proj/fc2 -> QuantAct(16) -> Block.qact2/qact4 residual add
 * (vit_quant.py:84-85,135,141; layers_quant.py:150-151):
 * t = clamp16(rq(acc, dy_ch[j])); out = clamp16(rq(t, dy_main) + rq(residual, dy_res)). */
int ivit_linear_i8_requant_residual(ivit_handle h, const int8_t *x, const int8_t *w,
                                    const int32_t *bias, const ivit_dyadic *dy_ch,
                                    ivit_dyadic dy_main, ivit_dyadic dy_res,
                                    const int16_t *residual, int16_t *out, int M, int N, int K);

/* a1 + a3 + head split (vit_quant.py:61-69): qkv Linear -> QuantAct(8) -> q,k as
 * [B,H,T,dh] and v TRANSPOSED as [B,H,dh,ldv] (token dim contiguous, ldv % 16 == 0,
 * ldv >= T) — the layout the MFMA attn·v operand wants.  x is [B*T, D], w [3D, D].
 * ldv == 0 (round 6): v ROW-major [B,H,T,dh] like q and k, for ivit_attention_fused_rowlut(ldv = 0). */
int ivit_linear_i8_qkv(ivit_handle h, const int8_t *x, const int8_t *w, const int32_t *bias,
                       const ivit_dyadic *dy_ch, int8_t *q, int8_t *k, int8_t *vt, int B, int T,
                       int H, int dh, int ldv);

/* ---- constants: one packed byte blob per frozen model (int8 weights, int32 biases, dyadic tables, position
 * embedding; layout = the host packer's table) — how a C host distributes them without
 * Python or torch.distributed (SURVEY.md §8b minimum surface, §8e).  Replaces the reference's per-GPU
 * `model.cuda()` of fp32 parameters + per-forward re-quantisation (quant_modules.py:68-91).
 * upload: host -> device on the handle's stream (asynchronous; the host buffer must stay valid until the stream
 * reaches it).  broadcast: in-place ncclBroadcast of the device blob from rank `root` over a caller-supplied RCCL
 * communicator (ncclComm_t), enqueued on the handle's stream — ONE large message, ring over xGMI; the only
 * collective of the data-parallel path.  librccl.so is resolved at first use.                                   */
int ivit_constants_upload(ivit_handle h, const void *host_blob, size_t bytes, void *device_blob);
int ivit_constants_broadcast(ivit_handle h, void *device_blob, size_t bytes, int root, void *rccl_comm);

/* ---- linear plans: a frozen QuantLinear prepared once (quant_modules.py:67-97 with QuantAct.fix(), :153-157).
 * ivit_linear_plan_create precomputes the per-channel multipliers c[n] = m*2^-e and checks, from
 * sum_k |w[n,k]| and bias[n], the two bounds under which the persistent pipelined GEMM (csrc/ivit_gemm3.h)
 * is bit-identical to the reference arithmetic.  It allocates a small device buffer and SYNCHRONISES the
 * handle's stream: a build-time call (freeze), never on the per-batch path.  w / bias / dy_ch must outlive
 * the plan.  The *_planned entry points are drop-ins for the unplanned ones above (same results bit for bit);
 * shapes or constants outside the pipelined kernel's contract run on the launch-per-tile kernels.          */
typedef struct ivit_linear_plan_s *ivit_linear_plan;
int ivit_linear_plan_create(ivit_handle h, const int8_t *w, const int32_t *bias, const ivit_dyadic *dy_ch, int N, int K,
                            ivit_linear_plan *out);
int ivit_linear_plan_destroy(ivit_linear_plan p);
/* pipelined_ok: the persistent kernel may be used; single_fma_ok: its one-FMA requant form is exact.        */
int ivit_linear_plan_query(ivit_linear_plan p, int *pipelined_ok, int *single_fma_ok);
/* debug: copy the 8 KB behind the plan's store scratch to the host (kernel timeline traces of -DG3_TRACE builds;
 * zeros in production builds).  Synchronises the device.                                                     */
int ivit_debug_plan_scratch(ivit_linear_plan p, void *host_dst, int nbytes);
int ivit_linear_i8_requant_planned(ivit_handle h, ivit_linear_plan p, const int8_t *x, int bits, void *out, int M);
int ivit_linear_i8_requant_residual_planned(ivit_handle h, ivit_linear_plan p, const int8_t *x, ivit_dyadic dy_main,
                                            ivit_dyadic dy_res, const int16_t *residual, int16_t *out, int M);
int ivit_linear_i8_qkv_planned(ivit_handle h, ivit_linear_plan p, const int8_t *x, int8_t *q, int8_t *k, int8_t *vt,
                               int B, int T, int H, int dh, int ldv);
/* Round 6: K = 384 layers on the kernel that keeps a CU's tokens in LDS and a 64-channel slab of weights in registers
 * (csrc/ivit_gemm_ws.h).  ivit_linear_plan_prepare_ws adds the weights in that kernel's fragment order to a plan (K = 384, N a
 * multiple of 64 up to 1536; IVIT_ERR_UNSUPPORTED otherwise; idempotent; build-time call, synchronises the handle's stream).
 * On a prepared plan ivit_linear_i8_qkv_planned(ldv = 0, dh = 64) and ivit_linear_i8_requant_residual_planned (multipliers in the
 * fast range) run on it — same bytes as on an unprepared plan.
 * ivit_layernorm_linear_i8_qkv_planned = norm1 + qact1 + attn.qkv of a block in ONE launch (vit_quant.py:136-137, 65-74;
 * quant_modules.py:353-386 + quant_utils.py:213-253 + quant_modules.py:21-80): x16 [B*T, 384] is the block's 16-bit input with
 * per-tensor `scale`, bias_int / sc / ln_dy are ivit_layernorm_requant's arguments for norm1, q / k / v are [B*H, T, 64] each
 * (v ROW-major, the ldv = 0 layout).  The bytes are those of ivit_layernorm_requant followed by ivit_linear_i8_qkv_planned;
 * norm1's 8-bit output never exists in HBM.  IVIT_ERR_UNSUPPORTED (nothing launched) unless the plan is prepared, dh = 64 and
 * B*H*T*64 < 2^31.                                                                                             */
int ivit_linear_plan_prepare_ws(ivit_handle h, ivit_linear_plan p);
/* IntLayerNorm + QuantAct(8) + QuantLinear + QuantAct(8) in one launch, plain [M, N] output (swin_quant.py:256-258 + 121-130: norm1 -> qact1 -> attn.qkv of a
 * C = 384 block, activations in natural token order): out8 == ivit_layernorm_requant followed by ivit_linear_i8_requant_planned(bits = 8).
 * IVIT_ERR_UNSUPPORTED (nothing launched) unless the plan is a prepared K = 384 one.  On a prepared plan
 * ivit_linear_i8_requant_planned(bits = 8) runs on the same kernel.                                                            */
int ivit_layernorm_linear_i8_requant_planned(ivit_handle h, ivit_linear_plan p, const int16_t *x16, float scale,
                                             const float *bias_int, const float *sc, const ivit_dyadic *ln_dy, int8_t *out8, int M);
/* attn.proj + qact2 with the identity branch + norm2 + qact3 of a D = 384 block in ONE launch (vit_quant.py:137-140):
 * out [M, 384] = ivit_linear_i8_requant_residual_planned's result, ln_out8 [M, 384] = ivit_layernorm_requant(out, ln_scale,
 * ln_bias_int, ln_sc, ln_dy) — the workgroup that produced a row normalises it.  IVIT_ERR_UNSUPPORTED (nothing launched) unless
 * the plan is a prepared 384 x 384 one and both residual multipliers are in the fast range (|m * 2^-e| < 2^9).  Measured SLOWER
 * than the two launches in the DeiT-S forward (profiles/README.md, round 6): the native runner does not use it.            */
int ivit_linear_i8_requant_residual_layernorm_planned(ivit_handle h, ivit_linear_plan p, const int8_t *x, ivit_dyadic dy_main,
                                                      ivit_dyadic dy_res, const int16_t *residual, int16_t *out, int M,
                                                      float ln_scale, const float *ln_bias_int, const float *ln_sc,
                                                      const ivit_dyadic *ln_dy, int8_t *ln_out8);
int ivit_layernorm_linear_i8_qkv_planned(ivit_handle h, ivit_linear_plan p, const int16_t *x16, float scale,
                                         const float *bias_int, const float *sc, const ivit_dyadic *ln_dy, int8_t *q,
                                         int8_t *k, int8_t *v, int B, int T, int H, int dh);

/* ---- a2  QuantMatMul.forward  (quant_modules.py:223-228), batched, "NT" form:
 * C[b] = A[b] (M x K) * B[b]^T (B[b] is N x K), int32.  q·kᵀ: A=q, B=k.
 * lda/ldb/ldc in elements, strides per batch in elements; lda,ldb % 16 == 0.            */
int ivit_bmm_nt_i8(ivit_handle h, const int8_t *A, const int8_t *B, int32_t *C, int nb, int M,
                   int N, int K, int lda, int ldb, int ldc, int64_t strideA, int64_t strideB,
                   int64_t strideC);
/* attn·v with the 16-bit Shiftmax output as A (values 0..32768): exact, two int8 MFMA
 * passes (a-16384 = 256*hi + lo) + 16384*colsum(B).  B = vᵀ [N=dh, K=T].                */
int ivit_bmm_nt_u16i8(ivit_handle h, const uint16_t *A, const int8_t *B, int32_t *C, int nb,
                      int M, int N, int K, int lda, int ldb, int ldc, int64_t strideA,
                      int64_t strideB, int64_t strideC);

/* a2 + a3, attention-shaped (vit_quant.py:70-74, 79-83):
 * scores8[bh,i,j] = clamp8(rq(q[bh,i,:]·k[bh,j,:], dy));  scores8 is [B*H, T, lds].       */
int ivit_attn_qk_requant(ivit_handle h, const int8_t *q, const int8_t *k, ivit_dyadic dy,
                         int8_t *scores8, int BH, int T, int dh, int lds);
/* ctx8[b,i,h*dh+d] = clamp8(rq(sum_j p[bh,i,j]*v[bh,j,d], dy));  p is [B*H,T,ldp] uint16,
 * vt is [B*H,dh,ldv];  ctx8 is [B,T,H*dh] (heads merged, vit_quant.py:81).               */
int ivit_attn_pv_requant(ivit_handle h, const uint16_t *p, const int8_t *vt, ivit_dyadic dy,
                         int8_t *ctx8, int B, int H, int T, int dh, int ldp, int ldv);

/* a10 core, fused (vit_quant.py:70-83): matmul_1 -> *scale -> qact_attn1 -> IntSoftmax(16)
 * -> matmul_2 -> qact2 in ONE kernel per (image, head); scores and probabilities stay in
 * registers.  Bit-identical to ivit_attn_qk_requant + ivit_shiftmax + ivit_attn_pv_requant.
 * q,k [B*H,T,dh], vt [B*H,dh,ldv] (as written by ivit_linear_i8_qkv), ctx8 [B,T,H*dh].
 * Built for dh == 64 and T <= 640 (DeiT/ViT at 224 and 384); else IVIT_ERR_UNSUPPORTED.   */
int ivit_attention_fused(ivit_handle h, const int8_t *q, const int8_t *k, const int8_t *vt,
                         ivit_dyadic dy_qk, float s_softmax, ivit_dyadic dy_pv, int8_t *ctx8,
                         int B, int H, int T, int dh, int ldv);

/* The same with Shiftmax's exp_int taken from tables built on the host for the layer's frozen scale
 * (ivit_amd.freeze.shiftmax_tables; exhaustively checked against the arithmetic form when built):
 * exp_int = exp_t[exp_aq[exp_cls[vmax]][v] + max(v - vmax, dmin) - dmin].  exp_aq uint16 [nclass][256],
 * exp_t float [t_count], exp_cls uint8 [256] (device; exp_aq and exp_t 16-byte aligned, exp_cls 4-byte aligned — they are
 * copied into LDS in 16-byte pieces; IVIT_ERR_INVALID otherwise).  Same integers as ivit_attention_fused.        */
int ivit_attention_fused_lut(ivit_handle h, const int8_t *q, const int8_t *k, const int8_t *vt,
                             ivit_dyadic dy_qk, float s_softmax, const uint16_t *exp_aq, const float *exp_t,
                             const uint8_t *exp_cls, int nclass, int t_count, int dmin, ivit_dyadic dy_pv,
                             int8_t *ctx8, int B, int H, int T, int dh, int ldv);

/* Row form of the same tables (round 6).  In a score row with maximum vmax only the scores v in (vmax + dmin, vmax] have an
 * exp_int above the floor constant (IntSoftmax.int_exp_shift, quant_modules.py:469-481), so exp_int as a function of
 * dd = max(v - vmax, dmin) - dmin is ONE line of R = 1 - dmin entries per value of vmax:
 *     rowtab[vmax + 128][dd] = exp_t[exp_aq[exp_cls[vmax]][vmax + dmin + dd] + dd],   float [256][64] (device, 16-byte aligned).
 * ivit_shiftmax_rowtable builds it on the device from the tables of ivit_attention_fused_lut (IVIT_ERR_UNSUPPORTED when a
 * line does not fit: 1 - dmin > 64; the two-level form then stays); ivit_attention_fused_rowlut is ivit_attention_fused with
 * ONE table gather per score: a wavefront fetches the 16 lines of its query tile once the row maxima are known.  Needs
 * |m 2^-e| < 2^9 for both requant multipliers (IVIT_ERR_UNSUPPORTED otherwise).  Same integers as ivit_attention_fused
 * (vit_quant.py:70-83).
 * ldv == 0 (this entry point only): `vt` is v ROW-major [B*H, T, dh] — what ivit_linear_i8_qkv / _planned write when THEIR ldv is 0
 * (one 16-byte store per token and head instead of sixteen byte stores); the kernel transposes on its way into the LDS.      */
int ivit_shiftmax_rowtable(ivit_handle h, const uint16_t *exp_aq, const float *exp_t, const uint8_t *exp_cls,
                           int nclass, int t_count, int dmin, float *rowtab);
int ivit_attention_fused_rowlut(ivit_handle h, const int8_t *q, const int8_t *k, const int8_t *vt,
                                ivit_dyadic dy_qk, float s_softmax, const float *rowtab, int dmin, ivit_dyadic dy_pv,
                                int8_t *ctx8, int B, int H, int T, int dh, int ldv);

/* ---- a3  QuantAct.forward with a previous scale -> fixedpoint_mul.forward
 * (quant_modules.py:197-206, quant_utils.py:192-253).  z int32 or float (integer-valued;
 * the I-LayerNorm output exceeds int32), [rows, C];  dy has nch = 1 or C entries;
 * optional identity z_id (int32, same shape) with scalar dy_id;  bits 8 -> int8 out,
 * 16 -> int16 out, 32 -> int32 out.                                                      */
int ivit_requant_i32(ivit_handle h, const int32_t *z, const ivit_dyadic *dy, int nch,
                     const int32_t *z_id, const ivit_dyadic *dy_id, int bits, void *out,
                     int64_t rows, int C);
int ivit_requant_i16(ivit_handle h, const int16_t *z, const ivit_dyadic *dy, int nch,
                     const int32_t *z_id, const ivit_dyadic *dy_id, int bits, void *out,
                     int64_t rows, int C);
int ivit_requant_f32(ivit_handle h, const float *z, const ivit_dyadic *dy, int nch,
                     const int32_t *z_id, const ivit_dyadic *dy_id, int bits, void *out,
                     int64_t rows, int C);

/* ---- a5  IntSoftmax.forward (Shiftmax)  (quant_modules.py:469-497).
 * x int8 [rows, n] (row stride ld_in), per-tensor scale; out_bits 16 (ViT/DeiT) or 8 (Swin);
 * integer probabilities with scale 2^-(out_bits-1), stored uint16 (16-bit results reach
 * 32768).  fp32-faithful interior incl. torch's CPU summation order.                     */
int ivit_shiftmax(ivit_handle h, const int8_t *x, int64_t rows, int n, int ld_in, float scale,
                  int out_bits, uint16_t *out, int ld_out);

/* ---- a6  IntGELU.forward (ShiftGELU)  (quant_modules.py:410-445).
 * x int8 [rows, C], per-tensor scale -> out16[i] = Q*sigmoid_int (scale s*2^-7).          */
int ivit_shiftgelu(ivit_handle h, const int8_t *x, int64_t rows, int C, float scale,
                   int16_t *out16);
/* a6 + a3 (layers_quant.py:146-147): ... -> clamp8(rq(Q*sigmoid_int, dy))               */
int ivit_shiftgelu_requant(ivit_handle h, const int8_t *x, int64_t rows, int C, float scale,
                           ivit_dyadic dy, int8_t *out8);

/* Same function as ivit_shiftgelu_requant, table form.  For a frozen layer the int8 result
 * depends only on (Q, row max): ivit_shiftgelu_build_table fills table[(qmax+128)*256 +
 * (Q+128)] (65536 bytes) once with the same device arithmetic; the per-token call is then
 * a row max plus byte gathers (HBM-bound instead of VALU-bound).                          */
int ivit_shiftgelu_build_table(ivit_handle h, float scale, ivit_dyadic dy, int8_t *table);
int ivit_shiftgelu_requant_lut(ivit_handle h, const int8_t *x, int64_t rows, int C,
                               const int8_t *table, int8_t *out8);

/* ---- a7  IntLayerNorm.forward  (quant_modules.py:353-386).
 * x int16 [rows, C] with per-tensor scale; bias_int[c] = floor(fl(fl(b/w)/sf)) and
 * sc[c] = fl(sf*w[c]) from the host.  z[i,c] = rne(fl(fl(out*sc)/sc)) as float — the
 * integer the following QuantAct derives (quant_utils.py:220).                           */
int ivit_layernorm(ivit_handle h, const int16_t *x, int64_t rows, int C, float scale,
                   const float *bias_int, const float *sc, float *z);
/* a7 + a3 (vit_quant.py:131-132): ... -> clamp8(rq(z, dy_ch[c])).  row_stride in
 * elements lets the final norm read only the class-token rows (vit_quant.py:271-272).    */
int ivit_layernorm_requant(ivit_handle h, const int16_t *x, int64_t rows, int C,
                           int64_t row_stride, float scale, const float *bias_int,
                           const float *sc, const ivit_dyadic *dy_ch, int8_t *out8);

/* ---- a8  PatchEmbed.forward: QuantConv2d(kernel=stride=P) -> QuantAct(16)
 * (layers_quant.py:184-196, quant_modules.py:297-330), then class token + position
 * embedding (vit_quant.py:259-265).
 * ivit_im2col_patch: NCHW int8 image -> [B*gh*gw, Cin*P*P] rows in conv-weight order.    */
int ivit_im2col_patch(ivit_handle h, const int8_t *img, int B, int Cin, int H, int W, int P,
                      int8_t *rows);
/* x16[b,0,:]  = clamp16(rq(z_cls[:], dy_x) + rq(pos[0,:], dy_pos))
 * x16[b,1+i,:] = clamp16(rq(patch16[b,i,:], dy_x) + rq(pos[1+i,:], dy_pos))             */
int ivit_embed_finish(ivit_handle h, const int16_t *patch16, const int32_t *z_cls,
                      const int16_t *pos, ivit_dyadic dy_x, ivit_dyadic dy_pos, int16_t *x16,
                      int B, int T, int D);
/* Round 6: the three steps above in one GEMM launch (plus the B class-token rows): the A rows are gathered from the NCHW images
 * (every 16-byte chunk of an im2col row is one pixel row of a 16 x 16 patch), the epilogue requantises to 16 bits, adds the
 * patch's position embedding and writes row 1 + i of its image.  w [D, Cin*P*P] in conv-weight order, bias int32 [D] or NULL,
 * dy_ch [D] as for ivit_linear_i8_requant(bits = 16).  x16 == ivit_im2col_patch -> ivit_linear_i8_requant(16) -> ivit_embed_finish.
 * IVIT_ERR_UNSUPPORTED (nothing launched) unless P == 16, H and W multiples of 16, Cin*P*P a multiple of 64 and |dy_x|, |dy_pos| < 2^9. */
int ivit_patch_embed(ivit_handle h, const int8_t *images, int B, int Cin, int H, int W, int P, const int8_t *w,
                     const int32_t *bias, const ivit_dyadic *dy_ch, const int32_t *z_cls, const int16_t *pos,
                     ivit_dyadic dy_x, ivit_dyadic dy_pos, int16_t *x16, int D);

/* ---- a9-a11  whole-model runner: VisionTransformer.forward (vit_quant.py:254-282) with
 * Block.forward (:130-143), Attention.forward (:59-88) and Mlp.forward (layers_quant.py:144-153)
 * chained natively.  Replaces the Python module tree for inference on frozen constants: one C call
 * per batch, nothing allocated after ivit_vit_create, the whole forward capturable in a hipGraph.
 *
 * All `const T *` members point into DEVICE memory owned by the caller (typically one packed blob
 * that was broadcast once over RCCL); scalars and by-value dyadics are host values.            */
typedef struct ivit_vit_config {
    int img_size, patch_size, in_chans, embed_dim, depth, num_heads, hidden_dim, num_classes;
} ivit_vit_config;

typedef struct ivit_vit_block {
    /* norm1 -> qact1 (vit_quant.py:131-132) */
    float s_ln1; const float *n1_bias_int; const float *n1_sc; const ivit_dyadic *n1_dy;
    /* attn.qkv -> qact1 (:63-65); q.k^T*scale -> qact_attn1 (:71-73); Shiftmax (:74);
       attn.v -> qact2 (:77-79); proj -> qact3 (:80-82) */
    const int8_t *qkv_w; const int32_t *qkv_b; const ivit_dyadic *qkv_dy;
    ivit_dyadic dy_qk; float s_softmax; ivit_dyadic dy_pv;
    const uint16_t *exp_aq; const float *exp_t; const uint8_t *exp_cls;   /* optional Shiftmax tables (NULL: arithmetic) */
    int exp_nc, exp_tcount, exp_dmin;
    const int8_t *proj_w; const int32_t *proj_b; const ivit_dyadic *proj_dy;
    ivit_dyadic res1_main, res1_res;                       /* qact2 with identity (:134) */
    /* norm2 -> qact3 (:135-136); mlp (layers_quant.py:144-153); qact4 with identity (:138) */
    float s_ln2; const float *n2_bias_int; const float *n2_sc; const ivit_dyadic *n2_dy;
    const int8_t *fc1_w; const int32_t *fc1_b; const ivit_dyadic *fc1_dy;
    float s_gelu; ivit_dyadic dy_gelu;
    const int8_t *fc2_w; const int32_t *fc2_b; const ivit_dyadic *fc2_dy;
    ivit_dyadic res2_main, res2_res;
} ivit_vit_block;

typedef struct ivit_vit_params {
    const int8_t *pe_w; const int32_t *pe_b; const ivit_dyadic *pe_dy;   /* patch_embed.proj -> qact */
    const int32_t *z_cls; const int16_t *pos; ivit_dyadic dy_x, dy_pos;  /* cls token, pos_embed, qact1 */
    const ivit_vit_block *blocks_host;                                   /* HOST array [depth]        */
    float s_ln; const float *n_bias_int; const float *n_sc; const ivit_dyadic *n_dy;   /* norm -> qact2 */
    const int8_t *head_w; const int32_t *head_b;                         /* head (int32 accumulators) */
} ivit_vit_params;

typedef struct ivit_vit_s *ivit_vit;

/* Builds the per-layer ShiftGELU tables (device allocation happens here, never later) and
 * `max_slices` internal streams/events for the sliced mode.                                     */
int ivit_vit_create(ivit_handle h, const ivit_vit_config *cfg, const ivit_vit_params *params,
                    int max_slices, ivit_vit *out);
int ivit_vit_destroy(ivit_vit m);
/* Caller-provided workspace: bytes for `batch` images cut into `nslices` slices.               */
int ivit_vit_workspace_bytes(ivit_vit m, int batch, int nslices, size_t *bytes);
/* Must run once per (workspace, batch, nslices) before the first forward (zeroes the padded
 * key columns of the transposed V buffers); asynchronous on the handle's stream.               */
int ivit_vit_workspace_init(ivit_vit m, void *workspace, size_t bytes, int batch, int nslices);
/* images int8 [batch, in_chans, img, img] -> logits int32 [batch, num_classes].
 * nslices == 1: everything on the handle's stream.  nslices > 1: the batch is cut into slices that
 * run on internal streams forked from / joined to the handle's stream with events, so VALU-bound
 * kernels of one slice overlap MFMA-bound GEMMs of another.  Same integers either way.          */
int ivit_vit_forward(ivit_vit m, const int8_t *images, int batch, int nslices, void *workspace,
                     size_t bytes, int32_t *logits);
/* hipGraph of one ivit_vit_forward call with fixed buffers; launch replays it on the handle's
 * stream.                                                                                      */
typedef struct ivit_graph_s *ivit_graph;
int ivit_vit_graph_create(ivit_vit m, const int8_t *images, int batch, int nslices, void *workspace,
                          size_t bytes, int32_t *logits, ivit_graph *out);
int ivit_graph_launch(ivit_graph g);
int ivit_graph_destroy(ivit_graph g);

/* ---- a12  whole-model runner for Swin: SwinTransformer.forward (swin_quant.py:539-564) with
 * SwinTransformerBlock.forward (:251-301), WindowAttention.forward (:121-169), PatchMerging.forward
 * (:328-349) and PatchEmbed.forward (layers_quant.py:184-196) chained natively; activations stay in natural
 * token order, roll / window partition / reverse are index arithmetic inside ivit_window_attention_fused.
 * Built for window 7 and head dim 32 (every reference factory).  Same conventions as ivit_vit_*.      */
typedef struct ivit_ln_params { const float *bias_int; const float *sc; const ivit_dyadic *dy; } ivit_ln_params;
typedef struct ivit_lin_params { const int8_t *w; const int32_t *b; const ivit_dyadic *dy; } ivit_lin_params;

typedef struct ivit_swin_config {
    int img_size, patch_size, in_chans, embed_dim, num_layers, window_size, mlp_ratio, num_classes;
    int depths[4];
    int num_heads[4];
} ivit_swin_config;

typedef struct ivit_swin_block {
    float s_in; ivit_ln_params n1;                         /* norm1 -> qact1 (:253-255)                         */
    ivit_lin_params qkv;                                   /* attn.qkv -> attn.qact1 (:123-124)                 */
    ivit_dyadic dy_qk, dy_a; const int16_t *relb;          /* qact_attn1; qact2 with the bias identity (:133-149) */
    float s_softmax; ivit_dyadic dy_pv;                    /* Shiftmax 8 bit (:151-157); attn.v -> qact3 (:159-162) */
    ivit_lin_params proj;                                  /* attn.proj -> attn.qact4 (:163-164)                */
    ivit_dyadic res1_main, res1_res;                       /* qact2 with identity (:289)                        */
    float s_mid; ivit_ln_params n2;                        /* norm2 -> qact3 (:291-292)                         */
    ivit_lin_params fc1; float s_gelu; ivit_dyadic dy_gelu; ivit_lin_params fc2;   /* mlp (layers_quant.py:144-153) */
    ivit_dyadic res2_main, res2_res;                       /* qact4 with identity (:296)                        */
    const uint16_t *exp_aq; const float *exp_t; const uint8_t *exp_cls;   /* optional Shiftmax tables for s_softmax (NULL: arithmetic) */
    int exp_nc, exp_tcount, exp_dmin;                      /* zero-initialise the struct: garbage here is read as pointers */
} ivit_swin_block;

typedef struct ivit_swin_merge {                           /* PatchMerging: norm -> qact1 -> reduction -> qact2 */
    float s_in; ivit_ln_params n; ivit_lin_params red;
} ivit_swin_merge;

typedef struct ivit_swin_params {
    ivit_lin_params pe; float s_bn;                        /* patch_embed.proj -> qact_before_norm (8 bit)      */
    ivit_ln_params pn; const ivit_dyadic *dy_qact1;        /* patch_embed.norm -> qact (16); qact1 (16), 1 entry */
    const ivit_swin_block *blocks_host;                    /* HOST array [sum(depths)], stage-major             */
    const ivit_swin_merge *merges_host;                    /* HOST array [num_layers - 1]                       */
    float s_norm_in; ivit_ln_params n;                     /* norm -> qact2                                     */
    ivit_dyadic dy_pool;                                   /* avgpool -> qact3                                  */
    const int8_t *head_w; const int32_t *head_b;
} ivit_swin_params;

typedef struct ivit_swin_s *ivit_swin;
int ivit_swin_create(ivit_handle h, const ivit_swin_config *cfg, const ivit_swin_params *params,
                     int max_slices, ivit_swin *out);
int ivit_swin_destroy(ivit_swin m);
int ivit_swin_workspace_bytes(ivit_swin m, int batch, int nslices, size_t *bytes);
int ivit_swin_forward(ivit_swin m, const int8_t *images, int batch, int nslices, void *workspace,
                      size_t bytes, int32_t *logits);
int ivit_swin_graph_create(ivit_swin m, const int8_t *images, int batch, int nslices, void *workspace,
                           size_t bytes, int32_t *logits, ivit_graph *out);

/* ---- a12  Swin-specific operators (models/swin_quant.py)
 * IntSoftmax on `attn + mask` (:151-156): float mask [nW, n, n] (0 / -100.0) added to fl(Q*s)
 * before the division by s; row r of the flattened [B_, H, n] rows uses window (r/(H*n)) % nW.
 * mask == NULL: identical to ivit_shiftmax.                                                */
int ivit_shiftmax_masked(ivit_handle h, const int8_t *x, int64_t rows, int n, int ld_in, float scale,
                         int out_bits, const float *mask, int nW, int H, uint16_t *out, int ld_out);
/* QuantAct with an identity that repeats every id_period elements (relative position bias
 * [H,N,N] broadcast over windows, :149): out = clamp(rq(z[i],dy) + rq(z_id[i % period],dy_id)) */
int ivit_requant_i32_bcast(ivit_handle h, const int32_t *z, ivit_dyadic dy, const int32_t *z_id,
                           int64_t id_period, ivit_dyadic dy_id, int bits, void *out, int64_t total);
/* AdaptiveAvgPool1d(1) over L (odd) tokens + QuantAct(8) (:553-555): x int8 [B,L,C] -> [B,C]   */
int ivit_avgpool_requant(ivit_handle h, const int8_t *x, int B, int L, int C, ivit_dyadic dy,
                         int8_t *out8);
/* IntLayerNorm whose row sums follow torch's order for a TOKEN-contiguous input — what the
 * reference computes in Swin stage 0, where activations keep the layout of
 * flatten(2).transpose(1,2) (layers_quant.py:188).  Same outputs as
 * ivit_layernorm otherwise.                                                                */
int ivit_layernorm_tokenorder(ivit_handle h, const int16_t *x, int64_t rows, int C, float scale,
                              const float *bias_int, const float *sc, int tokens_per_image, float *z);

/* the same with the per-channel QuantAct(8) that follows fused in (out8 int8 [rows, C]).  For C = 96 / 128 (the
 * vectorised kernel) x, out8, bias_int and sc must be 16-byte aligned: IVIT_ERR_INVALID otherwise; the same holds for
 * ivit_patch_norm_tokenorder below.                                                              */
int ivit_layernorm_tokenorder_requant(ivit_handle h, const int16_t *x, int64_t rows, int C, float scale,
                                      const float *bias_int, const float *sc, const ivit_dyadic *dy,
                                      int tokens_per_image, int8_t *out8);
/* PatchEmbed's tail in one pass (layers_quant.py:193-195, swin_quant.py:543): int8 conv output ->
 * norm (token-order sums) -> qact (16 bit, per-channel dy_ch) -> qact1 (16 bit, per-tensor dy2).          */
int ivit_patch_norm_tokenorder(ivit_handle h, const int8_t *x8, int64_t rows, int C, float scale,
                               const float *bias_int, const float *sc, const ivit_dyadic *dy_ch,
                               ivit_dyadic dy2, int tokens_per_image, int16_t *out16);
/* Fused windowed attention: everything of WindowAttention.forward (swin_quant.py:121-169) between
 * the qkv QuantAct and proj — q.k^T*scale -> qact_attn1 -> (+ relative position bias) qact2 ->
 * Shiftmax 8 bit on attn (+ shift mask, :151-156) -> attn.v -> qact3 — including torch.roll,
 * window_partition and their inverses (swin_quant.py:18-50, 268-287) as index arithmetic.
 * qkv int8 [B, R, R, 3, heads, dh] in natural token order; ctx int8 [B, R*R, heads*dh] natural order.
 * relb int16 [heads, 49, 49] = rq(quantised bias table gathered by relative_position_index,
 * dy(qact_table -> qact2)); dy_a = dy(qact_attn1 -> qact2).  Built for window 7, dh 32.       */
int ivit_window_attention_fused(ivit_handle h, const int8_t *qkv, ivit_dyadic dy_qk, ivit_dyadic dy_a,
                                const int16_t *relb, float s_softmax, ivit_dyadic dy_pv, int8_t *ctx,
                                int B, int R, int window, int shift, int heads, int dh);
/* The same with Shiftmax's exp_int (IntSoftmax.int_exp_shift, quant_modules.py:469-481) taken from the tables of
 * ivit_attention_fused_lut (ivit_amd.freeze.shiftmax_tables for the layer's frozen qact2 scale; same layout, alignment and
 * errors) in the windows that carry no shift mask; windows under the mask (swin_quant.py:151-156) keep the arithmetic
 * form, because the float -100 enters between the requotient's multiply and divide.  Same integers as the entry above. */
int ivit_window_attention_fused_lut(ivit_handle h, const int8_t *qkv, ivit_dyadic dy_qk, ivit_dyadic dy_a,
                                    const int16_t *relb, float s_softmax, const uint16_t *exp_aq, const float *exp_t,
                                    const uint8_t *exp_cls, int nclass, int t_count, int dmin, ivit_dyadic dy_pv,
                                    int8_t *ctx, int B, int R, int window, int shift, int heads, int dh);
/* Fused Mlp.forward + closing QuantAct(identity) for a narrow stage (layers_quant.py:144-153,
 * swin_quant.py:293-296): fc1 -> qact_gelu(8) -> ShiftGELU -> qact1(8) -> fc2 -> qact2(16) -> qact4(16, +identity)
 * with both weight matrices resident in LDS and the hidden tensor never written to HBM.
 * x int8 [M, C] (norm2 -> qact3), gelu_table from ivit_shiftgelu_build_table, residual / out int16 [M, C].
 * Built for C = 96, hidden = 384 (Swin-T/S stage 0); other shapes: IVIT_ERR_UNSUPPORTED.               */
int ivit_mlp_fused(ivit_handle h, const int8_t *x, const int8_t *w1, const int32_t *b1, const ivit_dyadic *dy1,
                   const int8_t *gelu_table, const int8_t *w2, const int32_t *b2, const ivit_dyadic *dy2,
                   ivit_dyadic dy_main, ivit_dyadic dy_res, const int16_t *residual, int16_t *out, int64_t M,
                   int C, int hidden);
/* The same chain for C = 384, hidden = 1536 (DeiT-S, Swin stage 2) on frozen linear plans: both weight matrices are
 * re-laid-out once in MFMA-fragment order and stream L2 -> registers, the hidden tile of 64 tokens lives in LDS between
 * fc1, the ShiftGELU table pass and fc2 (csrc/ivit_mlp.h).  The linear plans are borrowed and must outlive the Mlp plan.
 * IVIT_ERR_UNSUPPORTED for other shapes, for plans whose requant bound is not provable, and (at call time) for
 * residual multipliers >= 2^9: callers then run the unfused chain.  Replaces layers_quant.py:144-153 +
 * vit_quant.py:141-142 (swin_quant.py:296-300) like ivit_mlp_fused.                                        */
typedef struct ivit_mlp_plan_s *ivit_mlp_plan;
int ivit_mlp_plan_create(ivit_handle h, ivit_linear_plan fc1, ivit_linear_plan fc2, ivit_mlp_plan *out);
int ivit_mlp_plan_destroy(ivit_mlp_plan p);
/* Tuning / test switch: which of the plan's two kernels ivit_mlp_fused_planned launches.  0 = by shape (the default:
 * the role-split kernel of csrc/ivit_mlp_rs.h from two 80-token units per CU on, the lock-step kernel of csrc/ivit_mlp.h
 * below that), 1 = lock-step, 2 = role-split.  Both compute the same integers (layers_quant.py:144-153).  NOT safe to call
 * while another thread or stream is inside ivit_mlp_fused_planned on the same plan: it rewrites a field the launch reads.  */
int ivit_mlp_plan_select(ivit_mlp_plan p, int kernel);
int ivit_mlp_fused_planned(ivit_handle h, ivit_mlp_plan p, const int8_t *x, const int8_t *gelu_table,
                           ivit_dyadic dy_main, ivit_dyadic dy_res, const int16_t *residual, int16_t *out,
                           int64_t M);
/* Round 6: norm2 + qact3 + the fused Mlp + the block's residual QuantAct in ONE launch (vit_quant.py:139-142): every workgroup first
 * normalises the rows it is going to multiply (x16 [M, 384], the block's 16-bit stream, which is also the identity branch; scale /
 * bias_int / sc / ln_dy as for ivit_layernorm_requant) into scratch8 [M, 384] and reads its activation tiles from there.  out ==
 * ivit_layernorm_requant followed by ivit_mlp_fused_planned.  IVIT_ERR_UNSUPPORTED (nothing launched) where ivit_mlp_fused_planned is,
 * and where the launch would run on the lock-step kernel (fewer than two units per CU).                                        */
int ivit_layernorm_mlp_fused_planned(ivit_handle h, ivit_mlp_plan p, const int16_t *x16, float scale, const float *bias_int,
                                     const float *sc, const ivit_dyadic *ln_dy, int8_t *scratch8, const int8_t *gelu_table,
                                     ivit_dyadic dy_main, ivit_dyadic dy_res, int16_t *out, int64_t M);
/* PatchMerging's 2x2 gather (swin_quant.py:336-342): x [B,R,R,C] (in_bits 8 or 16) ->
 * int16 [B, (R/2)^2, 4C], channel blocks in the reference's torch.cat order.                  */
int ivit_patch_merge_gather(ivit_handle h, const void *x, int in_bits, int B, int R, int C, int16_t *out);
/* The same gather folded into the I-LayerNorm + QuantAct(8) that follows it in PatchMerging.forward (swin_quant.py:336-349:
 * gather, self.norm over 4C, qact1) — x int16 [B,R,R,C] -> out8 int8 [B (R/2)^2, 4C]; the gathered tensor never exists.  Same
 * integers as ivit_patch_merge_gather + ivit_layernorm_requant.  C in {96, 128, 192, 256, 384}; else IVIT_ERR_UNSUPPORTED.      */
int ivit_patch_merge_layernorm_requant(ivit_handle h, const int16_t *x, int B, int R, int C, float scale,
                                       const float *bias_int, const float *sc, const ivit_dyadic *dy_ch, int8_t *out8);
int ivit_widen_i8_i16(ivit_handle h, const int8_t *x, int16_t *out, int64_t n);
/* PatchMerging's reduction -> qact2 (swin_quant.py:343-349: QuantLinear(4C, 2C, bias = False) then an 8-bit QuantAct) whose consumer is the
 * next stage's 16-bit stream: out16 = clamp8(rq(acc + bias, dy_ch[n])) stored as int16 [M][N] — ivit_linear_i8_requant(bits = 8) followed by
 * ivit_widen_i8_i16 in one launch, same integers.  gemm_glds_kernel's shapes only (K % 32 == 0, K >= 64, not the short-K streaming kernel's):
 * IVIT_ERR_UNSUPPORTED otherwise, nothing launched.                                                                                  */
int ivit_linear_i8_requant8_store16(ivit_handle h, const int8_t *x, const int8_t *w, const int32_t *bias, const ivit_dyadic *dy_ch,
                                    int16_t *out16, int M, int N, int K);

/* ---- diagnostics (used by the parity tests only) ------------------------------------
 * q_ieee = n / d (compiler's correctly-rounded division) and q_lean = the hoisted-reciprocal
 * FMA sequence the kernels use for constant divisors; must agree bit for bit.            */
int ivit_debug_div(ivit_handle h, const float *n, const float *d, float *q_ieee, float *q_lean,
                   int64_t count);
/* r_ieee = fl(fl(q*d)/d) with the compiler's division and r_markstein = the three-operation form the
 * I-LayerNorm kernel uses (product, exact residual, one correction with RN(1/d)); must agree bit
 * for bit (quant_modules.py:359 x / scaling_factor after :204-206 x * scaling_factor).       */
int ivit_debug_requotient(ivit_handle h, const float *q, const float *d, float *r_ieee, float *r_markstein,
                          int64_t count);

#ifdef __cplusplus
}
#endif
#endif /* IVIT_H */
