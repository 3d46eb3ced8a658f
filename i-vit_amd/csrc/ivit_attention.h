// ivit_attention.h — fused integer attention for gfx950 (reference
// models/vit_quant.py:70-83: matmul_1 -> *scale -> qact_attn1 -> IntSoftmax(16) ->
// matmul_2 -> qact2), one workgroup per (image, head).
//
// Layout trick: S^T = K · Q^T is computed with v_mfma_i32_16x16x64_i8 (A = 16 keys x dh,
// B = 16 queries), so in the accumulator layout (col = lane&15 = query, row = 4*(lane>>4)
// + reg = key) the whole score row of one query sits in the registers of 4 lanes:
//   * row max and the torch-ordered row sum of Shiftmax are (almost) lane-local,
//   * the 16-bit probabilities, split into two int8 planes, ARE the A operand of the
//     P·V MFMA (K = 64 keys = 4 consecutive 16-key tiles), no cross-lane movement.
// Scores and probabilities never touch LDS or HBM.  K and V^T of the head are staged in
// LDS once (XOR-swizzled / stride-skewed so every ds_read_b128 is conflict-free); V^T is
// stored with the key order permuted to match the P fragment's key order.
//
// 16-bit P (0..32768): P - 16384 = 256*hi + lo, hi in [-64,64], lo in [-128,127];
// out = lo·V + 256*(hi·V) + 16384*colsum(V), all exact in int32.
#pragma once
#include "ivit_device.h"

struct AttnArgs {
    const int8_t *q, *k, *vt;  // q,k: [B*H, T, dh=64]; vt: [B*H, 64, ldv]
    int8_t *ctx;               // [B, T, H*64]
    int T, H, ldv;
    float s_softmax;           // scale of the int8 scores (qact_attn1)
    ivit_dyadic dy_qk, dy_pv;
    // optional Shiftmax tables (ivit_amd.freeze.shiftmax_tables): exp_int = et[aq[class(vmax)][v] + max(v - vmax, dmin) - dmin]
    const uint16_t *aq;        // [nc][256]
    const float *et;           // [t_count]
    const uint8_t *cls;        // [256] class of v
    int nc, t_count, dmin;
    // optional ROW tables (ivit_shiftmax_rowtable): rowtab[vmax + 128][dd] = exp_int of a score v = vmax + dmin + dd in a row
    // whose maximum is vmax, dd = max(v - vmax, dmin) - dmin in [0, 64): both table levels above folded per row maximum
    const float *rowtab;       // [256][64]
};
#define ATT_HAS_ROWTAB 1
// LDS of the row-line form: per wavefront 16 lines (one per query of the tile) of 64 entries at a pitch of 66 dwords — 8-byte
// aligned for the ds_write_b64 that fills them, and entry dd of line i sits on bank (2 i + dd) mod 32, so the 16 rows' entry 0
// (every score further than dmin below its row maximum reads entry 0) are 16 different banks
#define ATT_LINE_PITCH 264
#define ATT_ROWLINE_BYTES (ATT_WAVES * 16 * ATT_LINE_PITCH)

// wavefronts per workgroup.  8 = two workgroups per CU fill the 16 wave slots that 121 registers allow (round 4, after the prologue
// changes: 52-54 us against 60-61 with 7, which had been the better choice in round 3; 6: slower)
#ifndef ATT_WAVES
#define ATT_WAVES 8
#endif
#ifndef ATT_PROBE          // timing probes only (results invalid): 1 first gather lane-linear, 2 second gather lane-linear with the
#define ATT_PROBE 0        // first one dead, 4 second gather lane-linear with the first one kept alive, 8 no second gather
#endif
#ifndef ATT_PERMLANE       // probe: 1 = row maximum by v_permlane16/32_swap, 2 = the partner accumulators of the row sum likewise
#define ATT_PERMLANE 2     // (instead of ds_bpermute).  Same box, interleaved twice: 1 costs +3 us (it sits on the tile's critical path), 2 gains 0.3
#endif
#define ATT_DH 64

__device__ __forceinline__ int att_kswz(int row, int g) {   // 16-byte chunk position in a K row
    return g ^ ((0x78 >> (2 * ((row >> 2) & 3))) & 3);
}

template <int NB>  // key blocks of 64 (T <= 64*NB)
struct AttCfg {
    static constexpr int TK = NB * 64;
    static constexpr int NT = NB * 4;  // 16-key tiles
    // sV row stride in 16-byte slots: smallest >= TK/16 that is == 2 (mod 16)
    static constexpr int VS_SLOTS = ((TK / 16 + 13) / 16) * 16 + 2;
    static constexpr int VS = VS_SLOTS * 16;
    static constexpr int SK_BYTES = TK * 64;
    static constexpr int SV_BYTES = 64 * VS;
    static constexpr int SO_BYTES = ATT_WAVES * 1024;
    static constexpr int SMEM = SK_BYTES + SV_BYTES + SO_BYTES + 256 + 1024;
};

// FAST: |c_qk|, |c_pv| < 2^9 (host-checked) -> rq_fast is exact.  TT: the token count when it is known at
// compile time (197 / 577: the 224- and 384-pixel ViTs), 0 = run-time p.T.  With TT fixed every tile-validity
// test folds away; the generic form keeps ~70 loop-invariant lane masks alive and spills SGPRs in the hot loop.
// LUT = 1: shift-exp by table lookup (two LDS gathers per score instead of ~20 fp32 operations); the tables follow
// the fixed LDS regions and are copied in once per workgroup.
// LUT = 2 (round 6): ONE gather per score.  In a row with maximum vmax only the scores v in (vmax + dmin, vmax] have an exp_int
// above the floor constant, so exp_int as a function of dd = max(v - vmax, dmin) - dmin is a table line of R = 1 - dmin <= 64
// entries that depends on vmax alone: the wave fetches its 16 queries' lines (256 B each, from the 64 KB rowtab in L2) into LDS
// once the row maxima are known and every score then costs a saturating subtract, an address and one ds_read_b32 — the
// class-of-v gather, its address arithmetic and the per-workgroup copy of the two-level tables are gone.
#ifndef ATT_MINW           // waves per SIMD the kernel is compiled for (probe; 1 = whatever the workgroup size implies)
#define ATT_MINW 1
#endif
// VROW (round 6, with LUT = 2): v arrives ROW-major [B*H, T, 64] like q and k (p.ldv == 0) and is transposed on its way into the
// LDS — four keys x 16 channels per thread, byte-transposed in registers with v_perm — instead of v^T [B*H, 64, ldv] written by the
// qkv GEMM with sixteen byte stores per token (which cost that GEMM ~20 % of its time).
template <int NB, bool FAST, int TT = 0, int LUT = 0, bool VROW = false>
__global__ __launch_bounds__(ATT_WAVES * 64, ATT_MINW) void attn_fused_kernel(AttnArgs p) {
    using C = AttCfg<NB>;
    extern __shared__ __attribute__((aligned(16))) char dsmem[];
    char *sK = dsmem;
    char *sV = dsmem + C::SK_BYTES;
    char *sO = sV + C::SV_BYTES;
    int *sCol = reinterpret_cast<int *>(sO + C::SO_BYTES);
    float *sXq = reinterpret_cast<float *>(sO + C::SO_BYTES + 256);   // fl(fl(Q*s)/s) for Q = -128..127
    float *sT = reinterpret_cast<float *>(dsmem + C::SMEM);           // LUT = 1 only: exp table, then aq, then cls
    unsigned short *sAQ = reinterpret_cast<unsigned short *>(sT + (LUT == 1 ? (p.t_count + 3) & ~3 : 0));   // 16-byte aligned
    unsigned char *sCls = reinterpret_cast<unsigned char *>(sAQ + (LUT == 1 ? p.nc * 256 : 0));

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bh = blockIdx.x, b = bh / p.H, h = bh - b * p.H;
    const int T = TT ? TT : p.T;
    // The exp-table gathers below address the table as the LITERAL C::SMEM + offset (an integer-built LDS address keeps the
    // constant in the instruction's offset field; through the pointer every gather pays a v_add of the link-time base).  That
    // is only right while this kernel's dynamic LDS block starts at LDS address 0 — true as long as it has no static
    // __shared__ — so a layout change fails loudly here instead of reading the wrong table.
    if (LUT && (unsigned)(size_t)((__attribute__((address_space(3))) char *)dsmem) != 0u) __builtin_trap();
    const int8_t *qg = p.q + (long long)bh * T * 64;
    const int8_t *kg = p.k + (long long)bh * T * 64;
    const int8_t *vg = p.vt + (VROW ? (long long)bh * T * 64 : (long long)bh * 64 * p.ldv);

    // ---- stage K (rows >= T zero) and V^T (keys >= T zero, permuted) into LDS, and the tables.
    // Every global load of the prologue is issued before the first one is waited for (round 4: written as load-store
    // loops it was ~20 serial memory latencies per workgroup, 9 us of the launch when timed alone)
    constexpr int NTH = ATT_WAVES * 64;
    constexpr int KI = (C::TK * 4 + NTH - 1) / NTH, VI = VROW ? 4 * ((C::NT * 16 + NTH - 1) / NTH) : (64 * C::NT + NTH - 1) / NTH;
    v4i kreg[KI], vreg[VI];

#pragma unroll
    for (int i = 0; i < KI; ++i) {
        const int c = tid + i * NTH, row = c >> 2, g = c & 3;
        kreg[i] = v4i{0, 0, 0, 0};
        if (c < C::TK * 4 && row < T) kreg[i] = *reinterpret_cast<const v4i *>(kg + row * 64 + g * 16);
    }
    if constexpr (VROW) {
        // item c = (key quad c >> 2, channel group c & 3): rows 4 tq .. 4 tq + 3, bytes [16 dg, 16 dg + 16) — four lanes cover a row
#pragma unroll
        for (int i = 0; i < VI / 4; ++i) {
            const int c = tid + i * NTH, tq = c >> 2, dg = c & 3;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                vreg[4 * i + r] = v4i{0, 0, 0, 0};
                if (c < C::NT * 16 && 4 * tq + r < T) vreg[4 * i + r] = *reinterpret_cast<const v4i *>(vg + (4 * tq + r) * 64 + dg * 16);
            }
        }
    } else {
#pragma unroll
    for (int i = 0; i < VI; ++i) {
        const int c = tid + i * NTH, d = c / C::NT, t0 = (c - d * C::NT) * 16;
        vreg[i] = v4i{0, 0, 0, 0};
        if (c < 64 * C::NT && t0 < T) vreg[i] = *reinterpret_cast<const v4i *>(vg + (long long)d * p.ldv + t0);
    }
    }
    if (LUT == 1 && !(ATT_PROBE & 32)) {
        // table offsets are staged as BYTE offsets into sT (x4: t_count <= 16384 keeps them in 16 bits): a score's
        // table address is then one v_lshl_add_u32 on top of the saturating distance
        const int n4 = p.t_count >> 2, a4 = p.nc * 32;          // whole 16-byte chunks (the launcher checks the alignment)
        for (int i0 = tid; i0 < n4; i0 += 4 * NTH) {
            v4i t[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i0 + u * NTH < n4) t[u] = reinterpret_cast<const v4i *>(p.et)[i0 + u * NTH];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i0 + u * NTH < n4) reinterpret_cast<v4i *>(sT)[i0 + u * NTH] = t[u];
        }
        if (tid < (p.t_count & 3)) sT[(n4 << 2) + tid] = p.et[(n4 << 2) + tid];
        for (int i = tid; i < a4; i += NTH) {
            v4i t = reinterpret_cast<const v4i *>(p.aq)[i];
#pragma unroll
            for (int u = 0; u < 4; ++u) t[u] = (int)(((unsigned)t[u] & 0x3fff3fffu) << 2);
            *reinterpret_cast<v4i *>(reinterpret_cast<char *>(sAQ) + i * 16) = t;
        }
        if (tid < 64) reinterpret_cast<unsigned *>(sCls)[tid] = reinterpret_cast<const unsigned *>(p.cls)[tid];
    }
#pragma unroll
    for (int i = 0; i < KI; ++i) {
        const int c = tid + i * NTH, row = c >> 2, g = c & 3;
        if (c < C::TK * 4) *reinterpret_cast<v4i *>(sK + row * 64 + att_kswz(row, g) * 16) = kreg[i];
    }
    if constexpr (VROW) {
        // 4 x 4 byte transposes: dword w of the four rows -> for each channel 16 dg + 4 w + b one dword of four consecutive keys,
        // stored where the v^T path puts keys 4 tq .. 4 tq + 3 of that channel (block kb, position 16 g + 4 jj inside it)
#pragma unroll
        for (int i = 0; i < VI / 4; ++i) {
            const int c = tid + i * NTH, tq = c >> 2, dg = c & 3;
            if (c < C::NT * 16) {
                const int k = (4 * tq) & 63, pos = ((4 * tq) >> 6) * 64 + ((k >> 2) & 3) * 16 + (k >> 4) * 4;
                char *dst = sV + (dg * 16) * C::VS + pos;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const unsigned r0 = (unsigned)vreg[4 * i][w], r1 = (unsigned)vreg[4 * i + 1][w], r2 = (unsigned)vreg[4 * i + 2][w], r3 = (unsigned)vreg[4 * i + 3][w];
                    const unsigned a01l = __builtin_amdgcn_perm(r1, r0, 0x05010400u), a01h = __builtin_amdgcn_perm(r1, r0, 0x07030602u);
                    const unsigned a23l = __builtin_amdgcn_perm(r3, r2, 0x05010400u), a23h = __builtin_amdgcn_perm(r3, r2, 0x07030602u);
                    *reinterpret_cast<unsigned *>(dst + (4 * w + 0) * C::VS) = __builtin_amdgcn_perm(a23l, a01l, 0x05040100u);
                    *reinterpret_cast<unsigned *>(dst + (4 * w + 1) * C::VS) = __builtin_amdgcn_perm(a23l, a01l, 0x07060302u);
                    *reinterpret_cast<unsigned *>(dst + (4 * w + 2) * C::VS) = __builtin_amdgcn_perm(a23h, a01h, 0x05040100u);
                    *reinterpret_cast<unsigned *>(dst + (4 * w + 3) * C::VS) = __builtin_amdgcn_perm(a23h, a01h, 0x07060302u);
                }
            }
        }
    } else {
#pragma unroll
    for (int i = 0; i < VI; ++i) {
        const int c = tid + i * NTH, d = c / C::NT, ci = c - d * C::NT, t0 = ci * 16;   // ci: 16-key chunk index
        if (c < 64 * C::NT) {
            v4i v = vreg[i];
            if (t0 < T && t0 + 16 > T) {
                const int valid = T - t0;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const int nb = valid - w * 4;
                    const unsigned m = nb >= 4 ? 0xffffffffu : (nb <= 0 ? 0u : ((1u << (nb * 8)) - 1u));
                    v[w] &= (int)m;
                }
            }
            const int kb = ci >> 2, jj = ci & 3;
            char *dst = sV + d * C::VS + kb * 64 + jj * 4;
#pragma unroll
            for (int g = 0; g < 4; ++g) *reinterpret_cast<int *>(dst + g * 16) = v[g];
        }
    }
    }
    if (tid < 256) sXq[tid] = requotient_c((float)(tid - 128), rcp_prepare(p.s_softmax));
    if (tid < 64) sCol[tid] = 0;
    __syncthreads();
    {   // column sums of V (per d) over all keys: every wave takes 1 / ATT_WAVES of the keys of all 64 columns and adds its
        // partial sum with one LDS atomic (round 6: one wave walking all TK / 4 words was a ~2 us serial section per workgroup)
        constexpr int WPP = (C::TK / 4 + ATT_WAVES - 1) / ATT_WAVES;
        int s = 0;
        const int *row = reinterpret_cast<const int *>(sV + lane * C::VS);
#pragma unroll
        for (int w = 0; w < WPP; ++w)
            if ((C::TK / 4) % ATT_WAVES == 0 || wave * WPP + w < C::TK / 4) s = __builtin_amdgcn_sdot4(row[wave * WPP + w], 0x01010101, s, false);
        atomicAdd(&sCol[lane], s);
    }
    __syncthreads();

    const int qi = lane & 15, g = lane >> 4;
    const float s = p.s_softmax;
    const float x0 = floorf(-1.0f / s);
    const float nx0 = 15.0f * x0;
    const RcpC sr = rcp_prepare(s), x0r = rcp_prepare(x0);
    const double c_qk = p.dy_qk.m * p.dy_qk.r, c_pv = p.dy_pv.m * p.dy_pv.r;
    // |q.k| <= 64*2^14 = 2^20 and |sum P*v| <= 2^15*2^7 = 2^22 (sum P <= 2^15): rq_fast is exact if |c| < 2^9
    const int ntile = (T + 15) >> 4;       // live 16-key tiles
    const int nqt = (T + 15) >> 4;         // query tiles
    const int nvec = T >> 3, size = nvec >> 2;

    if (ATT_PROBE & 16) return;           // prologue only
    // (round 6: requesting this fragment together with the K / V^T rows of the prologue measured 0.5-1 us SLOWER on one box)
    v4i qnext = {0, 0, 0, 0};
    if (wave < nqt && wave * 16 + qi < T) qnext = *reinterpret_cast<const v4i *>(qg + (wave * 16 + qi) * 64 + g * 16);
    for (int qt = wave; qt < nqt; qt += ATT_WAVES) {
        const int q0 = qt * 16;
        // ---- Q fragment (B operand): query qi, dh bytes [16g, 16g+16)
        const v4i qf = qnext;
        if (qt + ATT_WAVES < nqt) {            // the next tile's fragment travels while this one is worked on
            qnext = v4i{0, 0, 0, 0};
            if (q0 + ATT_WAVES * 16 + qi < T) qnext = *reinterpret_cast<const v4i *>(qg + (q0 + ATT_WAVES * 16 + qi) * 64 + g * 16);
        }

        // ---- S^T tiles -> requant -> x~ = fl(fl(Q*s)/s) by table; running integer max
        // LUT form: scores are carried with a bias of VB = 384 (v' = v + 384 in [256, 511]; the bias rides in the requant's
        // magic constant for free), so that max(v - vmax - dmin, 0) is ONE unsigned saturating subtract
        constexpr int VB = LUT ? 384 : 0;
        float f[C::NT][4];
        int qmax = -128 + VB;
#pragma unroll
        for (int j = 0; j < C::NT; ++j) {
            if (j < ntile) {
                int row = j * 16 + qi;
                v4i kf = *reinterpret_cast<const v4i *>(sK + row * 64 + att_kswz(row, g) * 16);
                v4i acc = {0, 0, 0, 0};
                acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(kf, qf, acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int v = FAST ? min(max(__double2loint(__builtin_fma((double)acc[r], c_qk, 6755399441055744.0 + VB)), VB - 128), VB + 127)
                                 : rq_c((double)acc[r], c_qk, -128, 127) + VB;
                    f[j][r] = LUT ? __int_as_float(v) : sXq[v + 128];      // LUT: the (biased) integer itself waits for vmax
                    if (j * 16 + 15 < T || j * 16 + g * 4 + r < T) qmax = max(qmax, v);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) f[j][r] = 0.f;
            }
        }
        // the query's four lanes (qi, qi + 16, qi + 32, qi + 48): two register swaps instead of two trips through the LDS crossbar —
        // v_permlane16_swap(a, a) leaves (rows 0 0 2 2 | rows 1 1 3 3), v_permlane32_swap(a, a) (rows 0 1 0 1 | rows 2 3 2 3)
        if (ATT_PERMLANE & 1) {
            auto s16 = __builtin_amdgcn_permlane16_swap((unsigned)qmax, (unsigned)qmax, false, false);
            qmax = max((int)s16[0], (int)s16[1]);
            auto s32 = __builtin_amdgcn_permlane32_swap((unsigned)qmax, (unsigned)qmax, false, false);
            qmax = max((int)s32[0], (int)s32[1]);
        } else {
            qmax = max(qmax, __shfl_xor(qmax, 16));
            qmax = max(qmax, __shfl_xor(qmax, 32));
        }
        const float mx = sXq[qmax + 128 - VB];
        // byte offset of aq[class(vmax)][v' = 0]: ((class * 256 + 128) - VB) * 2
        const int rowbase2 = LUT == 1 ? ((int)sCls[qmax + 128 - VB] * 256 + 128 - VB) * 2 : 0;
        // LDS address of this query row's table line, once per row: the first gather's address is then ONE v_lshl_add_u32 per
        // score (round 4: left to the compiler it was a shift plus a three-input add on the run-time table base, 1.4 + 1.3 per score)
        typedef __attribute__((address_space(3))) const char att_lds_c;
        const unsigned aqrow = (unsigned)(size_t)((att_lds_c *)sAQ) + (unsigned)rowbase2;

        // ---- shift-exp; keys >= T contribute exactly 0
        if (LUT == 2) {
            // this wave's 16 table lines: lane (qi, g) brings entries [16 g, 16 g + 16) of its query's line
            const unsigned lines = (unsigned)C::SMEM + (unsigned)wave * (16 * ATT_LINE_PITCH);
            {
                const v4f *src = reinterpret_cast<const v4f *>(p.rowtab + (qmax + 128 - VB) * 64 + 16 * g);
                v4f l[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) l[u] = src[u];
                typedef float v2f __attribute__((ext_vector_type(2)));
                typedef __attribute__((address_space(3))) v2f lds_v2f;
                const unsigned dst = lines + (unsigned)qi * ATT_LINE_PITCH + (unsigned)g * 64;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    *(lds_v2f *)(size_t)(dst + u * 16) = v2f{l[u][0], l[u][1]};
                    *(lds_v2f *)(size_t)(dst + u * 16 + 8) = v2f{l[u][2], l[u][3]};
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // the lines are read by the other lanes of THIS wave only
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const unsigned qd = (unsigned)(qmax + p.dmin);      // max(v - vmax, dmin) - dmin == max(v' - qd', 0); qd' >= 1
            const unsigned line = lines + (unsigned)qi * ATT_LINE_PITCH;
#pragma unroll
            for (int j = 0; j < C::NT; ++j)
                if (j < ntile) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        unsigned a = (__builtin_elementwise_sub_sat((unsigned)__float_as_int(f[j][r]), qd) << 2) + line;
                        asm("" : "+v"(a));                       // one v_lshl_add_u32; not re-associated with the LDS base
                        const float e = *reinterpret_cast<__attribute__((address_space(3))) const float *>((size_t)a);
                        f[j][r] = (j * 16 + 15 < T || j * 16 + g * 4 + r < T) ? e : 0.f;
                    }
                }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // the next tile's lines overwrite these
            __builtin_amdgcn_wave_barrier();
        } else if (LUT) {
            // two dependent LDS gathers per score, issued as two whole sweeps so the reads of a sweep are all in
            // flight together (one wait per sweep instead of one per score)
            const unsigned qd = (unsigned)(qmax + p.dmin);      // max(v - vmax, dmin) - dmin == max(v' - qd', 0); qd' >= 1
#pragma unroll
            for (int j0 = 0; j0 < C::NT; j0 += 4) {             // 16 scores per sweep: bounded extra registers
                int e1[4][4];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
                    if (j0 + jj < C::NT && j0 + jj < ntile) {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            e1[jj][r] = (int)*reinterpret_cast<__attribute__((address_space(3))) const unsigned short *>(
                                (size_t)(ATT_PROBE & 1 ? aqrow + ((unsigned)(threadIdx.x & 63) << 1) : (((unsigned)__float_as_int(f[j0 + jj][r]) << 1) + aqrow)));
                    }
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
                    if (j0 + jj < C::NT && j0 + jj < ntile) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            e1[jj][r] += (int)(__builtin_elementwise_sub_sat((unsigned)__float_as_int(f[j0 + jj][r]), qd) << 2);
                            // opaque: otherwise the sum is re-associated with the LDS base of the table into a shift plus a
                            // three-input add instead of one v_lshl_add_u32
                            asm("" : "+v"(e1[jj][r]));
                        }
                    }
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
                    if (j0 + jj < C::NT && j0 + jj < ntile) {
                        const int j = j0 + jj;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if (ATT_PROBE & 4) asm volatile("" :: "v"(e1[jj][r]));
                            const float e = ATT_PROBE & 8 ? __int_as_float(e1[jj][r]) : *reinterpret_cast<__attribute__((address_space(3))) const float *>(
                                (att_lds_c *)(size_t)(unsigned)C::SMEM + (ATT_PROBE & 6 ? ((unsigned)(threadIdx.x & 63) << 2) + (ATT_PROBE & 2 ? (unsigned)e1[jj][r] & 0u : 0u) : (unsigned)e1[jj][r]));
                            f[j][r] = (j * 16 + 15 < T || j * 16 + g * 4 + r < T) ? e : 0.f;
                        }
                    }
            }
        } else {
#pragma unroll
            for (int j = 0; j < C::NT; ++j) {
                if (j < ntile) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float e = shift_exp_nonpos(f[j][r] - mx, x0r, nx0, 15);
                        f[j][r] = (j * 16 + 15 < T || j * 16 + g * 4 + r < T) ? e : 0.f;
                    }
                }
            }
        }

        // ---- row sum in torch's CPU order (ivit_device.h torch_order_sum32):
        // element t -> accumulator a = t & 31 = 16*(j&1) + 4*g + r, step = j >> 1
        float A0[2][4], A1[2][4], A2[2][4], A3[2][4];
#pragma unroll
        for (int pp = 0; pp < 2; ++pp)
#pragma unroll
            for (int r = 0; r < 4; ++r) { A0[pp][r] = 0.f; A1[pp][r] = 0.f; A2[pp][r] = 0.f; A3[pp][r] = 0.f; }
#pragma unroll
        for (int i = 0; i < C::NT / 2; ++i) {
            if (i < size) {
                const bool in_blocks = (i < (size & ~15));   // inside a full 16-step level block
#pragma unroll
                for (int pp = 0; pp < 2; ++pp)
#pragma unroll
                    for (int r = 0; r < 4; ++r) A0[pp][r] += f[2 * i + pp][r];
                if (in_blocks && ((i + 1) & 15) == 0) {
                    const int ii = i + 1;
#pragma unroll
                    for (int pp = 0; pp < 2; ++pp)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            A1[pp][r] += A0[pp][r]; A0[pp][r] = 0.f;
                            if ((ii & 0xF0) == 0) {
                                A2[pp][r] += A1[pp][r]; A1[pp][r] = 0.f;
                                if ((ii & 0xF00) == 0) { A3[pp][r] += A2[pp][r]; A2[pp][r] = 0.f; }
                            }
                        }
                }
            }
        }
#pragma unroll
        for (int pp = 0; pp < 2; ++pp)
#pragma unroll
            for (int r = 0; r < 4; ++r) { A0[pp][r] += A1[pp][r]; A0[pp][r] += A2[pp][r]; A0[pp][r] += A3[pp][r]; }
        // leftover whole 8-vectors (nvec % 4) go to accumulators a = 0..7 (lanes g = 0,1)
        for (int v = size * 4; v < nvec; ++v) {
            const int jv = v >> 1, odd = v & 1;
            float val[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < C::NT; ++j)
                if (j == jv) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) val[r] = f[j][r];
                }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float o = __shfl(val[r], (lane + 32 * odd) & 63);
                if (g < 2) A0[0][r] += o;
            }
        }
        // p[l] = ((acc[l] + acc[8+l]) + acc[16+l]) + acc[24+l];  a = 16*pp + 4*g + r
        float pl[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            // lane ^ 32's accumulators, needed on the lower 32 lanes only: after v_permlane32_swap(a, a) the second result holds
            // (upper half | upper half)
            float o0, o1;
            if (ATT_PERMLANE & 2) {
                o0 = __uint_as_float(__builtin_amdgcn_permlane32_swap(__float_as_uint(A0[0][r]), __float_as_uint(A0[0][r]), false, false)[1]);
                o1 = __uint_as_float(__builtin_amdgcn_permlane32_swap(__float_as_uint(A0[1][r]), __float_as_uint(A0[1][r]), false, false)[1]);
            } else {
                o0 = __shfl_xor(A0[0][r], 32);
                o1 = __shfl_xor(A0[1][r], 32);
            }
            pl[r] = ((A0[0][r] + o0) + A0[1][r]) + o1;   // valid on lanes g = 0 (l=r) and g = 1 (l=4+r)
        }
        float fin = 0.f;
        for (int t = nvec * 8; t < T; ++t) {   // scalar tail, sequential
            const int jt = t >> 4, gt = (t >> 2) & 3, rt = t & 3;
            float val = 0.f;
#pragma unroll
            for (int j = 0; j < C::NT; ++j)
                if (j == jt) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (r == rt) val = f[j][r];
                }
            fin += __shfl(val, (gt << 4) | qi);
        }
        if (nvec > 0) {
#pragma unroll
            for (int l = 0; l < 8; ++l) fin += __shfl(pl[l & 3], ((l >> 2) << 4) | qi);
        }
        const float F = recip_factor(fin);

        // ---- probabilities -> (hi, lo) int8 planes, packed as P·V A fragments.
        // P - 16384 = 256*hi + lo (lo signed)  <=>  lo = byte0(P), hi = byte1(P - 16256)
        const float F16 = F * 1.52587890625e-05f;     // F / 2**16: exact scaling, so
        v4i plo[NB], phi[NB];                         // fl(e*F16) == fl(e*F) / 2**16
#pragma unroll
        for (int kb = 0; kb < NB; ++kb)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int j = kb * 4 + jj;
                unsigned wl = 0, wh = 0xC0C0C0C0u;   // P = 0 -> hi = -64, lo = 0 (V is 0 there)
                if (j < ntile) {
                    // e*F16 >= 0: the float -> int conversion truncates, which IS the reference's floor
                    unsigned P0 = (unsigned)(int)(f[j][0] * F16), P1 = (unsigned)(int)(f[j][1] * F16);
                    unsigned P2 = (unsigned)(int)(f[j][2] * F16), P3 = (unsigned)(int)(f[j][3] * F16);
                    // P <= 2^15: pack pairs as 16-bit halves, subtract 16256 from both halves at once (the low 16 bits of
                    // P - 16256 are all the hi plane needs), then pick bytes 0 / 1 of each half
                    typedef unsigned short v2us __attribute__((ext_vector_type(2)));
                    const unsigned p01 = __builtin_amdgcn_perm(P1, P0, 0x05040100u), p23 = __builtin_amdgcn_perm(P3, P2, 0x05040100u);
                    wl = __builtin_amdgcn_perm(p23, p01, 0x06040200u);
                    const v2us off = {16256, 16256};
                    const unsigned s01 = __builtin_bit_cast(unsigned, (v2us)(__builtin_bit_cast(v2us, p01) - off));
                    const unsigned s23 = __builtin_bit_cast(unsigned, (v2us)(__builtin_bit_cast(v2us, p23) - off));
                    wh = __builtin_amdgcn_perm(s23, s01, 0x07050301u);
                }
                plo[kb][jj] = (int)wl;
                phi[kb][jj] = (int)wh;
            }

        // ---- P·V : out[query 4g+r][d = 16*dt + qi]
        v4i oL[4], oH[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) { oL[dt] = v4i{0, 0, 0, 0}; oH[dt] = v4i{0, 0, 0, 0}; }
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) {
            if (kb * 64 < T) {
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    v4i vf = *reinterpret_cast<const v4i *>(sV + (dt * 16 + qi) * C::VS + kb * 64 + g * 16);
                    oL[dt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(plo[kb], vf, oL[dt], 0, 0, 0);
                    oH[dt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(phi[kb], vf, oH[dt], 0, 0, 0);
                }
            }
        }
        // ---- epilogue: exact recombination, requant, stage [16 q][64 d] and store rows
        char *so = sO + wave * 1024;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const int cs = sCol[dt * 16 + qi];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int v = (int)((unsigned)oL[dt][r] + ((unsigned)oH[dt][r] << 8) + ((unsigned)cs << 14));
                int o = FAST ? min(max(rq_fast(v, c_pv), -128), 127) : rq_c((double)v, c_pv, -128, 127);
                so[(g * 4 + r) * 64 + dt * 16 + qi] = (char)o;
            }
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        {
            const int row = lane >> 2, ch = lane & 3;
            if (q0 + row < T) {
                v4i v = *reinterpret_cast<const v4i *>(so + row * 64 + ch * 16);
                *reinterpret_cast<v4i *>(p.ctx + ((long long)b * T + q0 + row) * (p.H * 64) + h * 64 + ch * 16) = v;
            }
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
}
