"""CPU twin of the C-ABI (oracle/ivit_twin.c): ivit_cpu_X takes the parameter list of ivit_X with host
pointers.  CPU part: the generated header is current, every twin symbol is exported, the twin agrees with the golden
fixtures / the oracle's Python-level chaining.  GPU part: ONE argument list per entry point, executed through the HIP
library (device pointers) and through the twin (host pointers) with the same ctypes signature — results must be equal
bit for bit."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, load_golden

import ivit_amd as iv
from conftest import golden_scales
from ivit_amd import _lib

sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_twin_header  # noqa: E402

_P = ctypes.c_void_p


@pytest.fixture(scope="module")
def twin():
    from oracle import oracle as orc
    lib = ctypes.CDLL(orc.build())
    for name in gen_twin_header.TWIN:
        fn = getattr(lib, "ivit_cpu_" + name)           # AttributeError = a declared twin is not exported
        fn.argtypes = _sig(name)                        # the C-ABI's own ctypes signature, unchanged
        fn.restype = ctypes.c_int
    return lib


# the two plan calls without a handle argument are declared outside _lib.SIGNATURES (which prepends nothing, but whose
# Handle.call wrapper does): same argtypes as _lib.load() sets
_EXTRA = {"linear_plan_destroy": [_P], "mlp_plan_destroy": [_P], "linear_plan_query": [_P, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]}


def _sig(name):
    return _EXTRA.get(name) or _lib.SIGNATURES["ivit_" + name]


def hp(a):
    return a.ctypes.data_as(_P)


def dyv(d):
    return _lib.Dyadic(float(d[0, 0]), float(d[0, 1]))


def test_twin_header_is_generated_from_the_c_abi():
    """oracle/ivit_twin.h == tools/gen_twin_header.py(include/ivit.h): same parameter lists by construction; ivit_twin.c
    includes it, so a drifting definition does not compile"""
    assert open(os.path.join(ROOT, "oracle", "ivit_twin.h")).read() == gen_twin_header.render()
    assert set("ivit_" + n for n in gen_twin_header.TWIN) <= set(_lib.SIGNATURES) | set("ivit_" + n for n in _EXTRA)


def test_twin_exports_every_declared_symbol(twin):
    out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(ROOT, "oracle", "libivit_oracle.so")],
                         capture_output=True, text=True, check=True).stdout
    for name in gen_twin_header.TWIN:
        assert f" T ivit_cpu_{name}\n" in out, name


def test_twin_operators_match_golden(twin):
    """the twin's single-operator entry points on the reference's own captured tensors (tests/golden/ops.npz)"""
    g = load_golden("ops.npz")
    for i in range(int(g["ln/n"])):
        x = np.ascontiguousarray(g[f"ln/{i}/x"])
        rows, C = x.shape
        bias_int, sc = iv.freeze.layernorm_constants(g[f"ln/{i}/w"], g[f"ln/{i}/b"])
        z = np.empty((rows, C), np.float32)
        assert twin.ivit_cpu_layernorm(None, hp(x), rows, C, float(g[f"ln/{i}/s"]), hp(bias_int), hp(sc), hp(z)) == 0
        assert np.array_equal(z, g[f"ln/{i}/z"]), i
        d = iv.freeze.dyadic(sc, g[f"ln/{i}/s_out"])
        o8 = np.empty((rows, C), np.int8)
        assert twin.ivit_cpu_layernorm_requant(None, hp(x), rows, C, C, float(g[f"ln/{i}/s"]), hp(bias_int), hp(sc),
                                               hp(d), hp(o8)) == 0
        assert np.array_equal(o8, g[f"ln/{i}/out8"]), i
    for i in range(int(g["requant/n"])):
        z = np.ascontiguousarray(g[f"requant/{i}/z"])
        rows, C = z.shape
        bits = int(g[f"requant/{i}/bits"])
        d = iv.freeze.dyadic(g[f"requant/{i}/s_pre"], g[f"requant/{i}/s_out"])
        out = np.empty((rows, C), {8: np.int8, 16: np.int16}[bits])
        if f"requant/{i}/z_id" in g.files:
            di = iv.freeze.dyadic(g[f"requant/{i}/s_id"], g[f"requant/{i}/s_out"])
            zid = np.ascontiguousarray(g[f"requant/{i}/z_id"])
            assert twin.ivit_cpu_requant_f32(None, hp(z), hp(d), d.shape[0], hp(zid), hp(di), bits, hp(out), rows, C) == 0
        else:
            assert twin.ivit_cpu_requant_f32(None, hp(z), hp(d), d.shape[0], None, None, bits, hp(out), rows, C) == 0
        assert np.array_equal(out.astype(np.int32), g[f"requant/{i}/out"]), i


def test_twin_rejects_bad_arguments(twin):
    """error behaviour of the C-ABI: status codes, never a crash"""
    x = np.zeros((4, 16), np.int8)
    assert twin.ivit_cpu_shiftmax(None, None, 4, 16, 16, 0.1, 16, hp(np.zeros((4, 16), np.uint16)), 16) == 1
    assert twin.ivit_cpu_shiftmax(None, hp(x), 4, 16, 8, 0.1, 16, hp(np.zeros((4, 16), np.uint16)), 16) == 1   # ld_in < n
    assert twin.ivit_cpu_linear_i8_requant(None, hp(x), hp(x), None, None, 8, hp(x), 4, 4, 16) == 1          # no dyadics


# ---------------------------------------------------------------- the same calls through both libraries
def _cases(rng, variant=0):
    """-> list of (entry point, args); an argument is a scalar / Dyadic, ("in", array), ("out", array) or None.
    variant 1 is a second argument list per entry point: ragged row counts, other channel / token counts, other scales."""
    I, O = (lambda a: ("in", np.ascontiguousarray(a))), (lambda a: ("out", a))
    cs = []
    V = variant
    M, N, K = ((300, 96, 64), (173, 160, 128))[V]
    x = rng.integers(-128, 128, (M, K), dtype=np.int8)
    w = np.rint(rng.normal(0, 40, (N, K)).clip(-127, 127)).astype(np.int8)
    b = rng.integers(-3000, 3000, N).astype(np.int32)
    s_pre = (10 ** rng.uniform(*((-5.5, -4.5), (-5.1, -4.2))[V], N)).astype(np.float32)
    d8, d16 = iv.freeze.dyadic(s_pre, np.float32((2e-2, 3.1e-2)[V])), iv.freeze.dyadic(s_pre, np.float32((1e-4, 1.7e-4)[V]))
    dm, dr = iv.freeze.dyadic(np.float32(1e-4), np.float32(2e-4)), iv.freeze.dyadic(np.float32(3e-4), np.float32(2e-4))
    res = rng.integers(-20000, 20000, (M, N)).astype(np.int16)
    cs.append(("quantize_input_f32", [I(rng.normal(0, 1, 5000).astype(np.float32)), 0.02, O(np.zeros(5000, np.int8)), 5000]))
    cs.append(("linear_i8", [I(x), I(w), I(b), O(np.zeros((M, N), np.int32)), M, N, K]))
    cs.append(("linear_i8_requant", [I(x), I(w), I(b), I(d8), 8, O(np.zeros((M, N), np.int8)), M, N, K]))
    cs.append(("linear_i8_requant", [I(x), I(w), I(b), I(d16), 16, O(np.zeros((M, N), np.int16)), M, N, K]))
    cs.append(("linear_i8_requant_residual", [I(x), I(w), I(b), I(d16), dyv(dm), dyv(dr), I(res), O(np.zeros((M, N), np.int16)), M, N, K]))
    # attention-shaped entry points: B = 2, H = 2, T = 50, dh = 64
    B, Hh, T, dh, ld = ((2, 2, 50, 64, 64), (3, 1, 37, 64, 48))[V]
    D = Hh * dh
    xq = rng.integers(-128, 128, (B * T, D), dtype=np.int8)
    wq = np.rint(rng.normal(0, 30, (3 * D, D)).clip(-127, 127)).astype(np.int8)
    bq = rng.integers(-3000, 3000, 3 * D).astype(np.int32)
    dq = iv.freeze.dyadic((10 ** rng.uniform(-5.5, -5.0, 3 * D)).astype(np.float32), np.float32(4e-2))
    q = rng.integers(-128, 128, (B * Hh, T, dh), dtype=np.int8)
    k = rng.integers(-128, 128, (B * Hh, T, dh), dtype=np.int8)
    vt = np.zeros((B * Hh, dh, ld), np.int8)
    vt[:, :, :T] = rng.integers(-128, 128, (B * Hh, dh, T), dtype=np.int8)
    dqk, dpv = iv.freeze.dyadic(np.float32(2e-4), np.float32(6e-2)), iv.freeze.dyadic(np.float32(3e-6), np.float32(9e-3))
    cs.append(("linear_i8_qkv", [I(xq), I(wq), I(bq), I(dq), O(np.zeros((B * Hh, T, dh), np.int8)), O(np.zeros((B * Hh, T, dh), np.int8)),
                                 O(np.zeros((B * Hh, dh, ld), np.int8)), B, T, Hh, dh, ld]))
    cs.append(("bmm_nt_i8", [I(q), I(k), O(np.zeros((B * Hh, T, T), np.int32)), B * Hh, T, T, dh, dh, dh, T, T * dh, T * dh, T * T]))
    p16 = np.zeros((B * Hh, T, ld), np.uint16)
    p16[:, :, :T] = rng.integers(0, 32769, (B * Hh, T, T)).astype(np.uint16)
    cs.append(("bmm_nt_u16i8", [I(p16), I(vt), O(np.zeros((B * Hh, T, dh), np.int32)), B * Hh, T, dh, T, ld, ld, dh, T * ld, dh * ld, T * dh]))
    cs.append(("attn_qk_requant", [I(q), I(k), dyv(dqk), O(np.zeros((B * Hh, T, ld), np.int8)), B * Hh, T, dh, ld]))
    cs.append(("attn_pv_requant", [I(p16), I(vt), dyv(dpv), O(np.zeros((B, T, D), np.int8)), B, Hh, T, dh, ld, ld]))
    cs.append(("attention_fused", [I(q), I(k), I(vt), dyv(dqk), 0.06, dyv(dpv), O(np.zeros((B, T, D), np.int8)), B, Hh, T, dh, ld]))
    # requant flavours
    z32 = rng.integers(-2 ** 20, 2 ** 20, (M, N)).astype(np.int32)
    zid = rng.integers(-30000, 30000, (M, N)).astype(np.int32)
    cs.append(("requant_i32", [I(z32), I(d8), N, None, None, 8, O(np.zeros((M, N), np.int8)), M, N]))
    cs.append(("requant_i32", [I(z32), I(dm), 1, I(zid), I(dr), 16, O(np.zeros((M, N), np.int16)), M, N]))
    cs.append(("requant_i16", [I(res), I(dm), 1, None, None, 16, O(np.zeros((M, N), np.int16)), M, N]))
    cs.append(("requant_f32", [I(z32.astype(np.float32) * 4096.0), I(iv.freeze.dyadic(s_pre * np.float32(1e-3), np.float32(2e-2))), N,
                               None, None, 8, O(np.zeros((M, N), np.int8)), M, N]))
    # elementwise operators
    R8 = (64, 61)[V]
    s8 = rng.integers(-128, 128, (R8, 197), dtype=np.int8)
    s8[3] = -128; s8[4] = 127; s8[5, :] = -100; s8[5, 17] = 90                       # flat, saturated and peaky rows
    cs.append(("shiftmax", [I(s8), R8, 197, 197, (0.07, 0.093)[V], 16, O(np.zeros((R8, 197), np.uint16)), 197]))
    cs.append(("shiftmax", [I(s8), R8, 197, 197, (0.11, 0.157)[V], 8, O(np.zeros((R8, 197), np.uint16)), 197]))
    Rg, Cg, sg = ((40, 384, 0.03), (37, 768, 0.045))[V]
    g8 = rng.integers(-128, 128, (Rg, Cg), dtype=np.int8)
    g8[2] = rng.integers(-128, -60, Cg, dtype=np.int8)                                # an all-negative row
    dg = iv.freeze.dyadic(np.float32(sg * 2.0 ** -7), np.float32((0.025, 0.033)[V]))
    tab = np.zeros(65536, np.int8)
    cs.append(("shiftgelu", [I(g8), Rg, Cg, sg, O(np.zeros((Rg, Cg), np.int16))]))
    cs.append(("shiftgelu_requant", [I(g8), Rg, Cg, sg, dyv(dg), O(np.zeros((Rg, Cg), np.int8))]))
    cs.append(("shiftgelu_build_table", [sg, dyv(dg), O(tab)]))
    C = (192, 384)[V]
    Rl = (48, 50)[V]
    xl = rng.integers(-9000, 9000, (Rl, C)).astype(np.int16)
    xl[1] = 1234                                                                       # a zero-variance row
    wl, bl = rng.uniform(0.4, 1.6, C).astype(np.float32), rng.normal(0, 0.3, C).astype(np.float32)
    wl[5] = -0.7
    bias_int, sc = iv.freeze.layernorm_constants(wl, bl)
    dl = iv.freeze.dyadic(sc, np.float32(0.03))
    sl = (2.5e-4, 3.3e-4)[V]
    cs.append(("layernorm", [I(xl), Rl, C, sl, I(bias_int), I(sc), O(np.zeros((Rl, C), np.float32))]))
    cs.append(("layernorm_requant", [I(xl), Rl, C, C, sl, I(bias_int), I(sc), I(dl), O(np.zeros((Rl, C), np.int8))]))
    # one channel with a multiplier far above the |z * c| < 2^31 bound: the block takes the v_rndne_f64 / saturating form of the
    # 8-bit requant (ivit_layernorm.h) instead of the magic-number one
    dl_wide = dl.copy()
    dl_wide[3:4] = iv.freeze.dyadic(sc[3:4], np.float32(1e-13))
    cs.append(("layernorm_requant", [I(xl), Rl, C, C, sl, I(bias_int), I(sc), I(dl_wide), O(np.zeros((Rl, C), np.int8))]))
    img = rng.integers(-128, 128, (2, 3, 32, 32), dtype=np.int8)
    cs.append(("im2col_patch", [I(img), 2, 3, 32, 32, 8, O(np.zeros((2 * 16, 3 * 64), np.int8))]))
    Te, De = ((17, 64), (10, 128))[V]
    cs.append(("embed_finish", [I(rng.integers(-20000, 20000, (2, Te - 1, De)).astype(np.int16)), I(rng.integers(-10 ** 6, 10 ** 6, De).astype(np.int32)),
                                I(rng.integers(-20000, 20000, (Te, De)).astype(np.int16)), dyv(dm), dyv(dr), O(np.zeros((2, Te, De), np.int16)), 2, Te, De]))
    # ---- round 3: Swin-specific operators
    a49 = rng.integers(-128, 128, (8 * 3 * 49, 49), dtype=np.int8)                     # [B_ = 8, H = 3, 49] rows
    mk = np.where(rng.random((4, 49, 49)) < 0.3, np.float32(-100.0), np.float32(0.0)).astype(np.float32)
    cs.append(("shiftmax_masked", [I(a49), 8 * 3 * 49, 49, 49, 0.05, 8, I(mk), 4, 3, O(np.zeros((8 * 3 * 49, 49), np.uint16)), 49]))
    cs.append(("shiftmax_masked", [I(a49), 8 * 3 * 49, 49, 49, 0.05, 8, None, 0, 0, O(np.zeros((8 * 3 * 49, 49), np.uint16)), 49]))
    zb = rng.integers(-128, 128, 6 * 3 * 2401).astype(np.int32)
    zi = rng.integers(-128, 128, 3 * 2401).astype(np.int32)
    da, db = iv.freeze.dyadic(np.float32(0.04), np.float32(0.05)), iv.freeze.dyadic(np.float32(0.01), np.float32(0.05))
    cs.append(("requant_i32_bcast", [I(zb), dyv(da), I(zi), 3 * 2401, dyv(db), 8, O(np.zeros(6 * 3 * 2401, np.int8)), 6 * 3 * 2401]))
    cs.append(("avgpool_requant", [I(rng.integers(-128, 128, (3, 49, 96), dtype=np.int8)), 3, 49, 96, dyv(iv.freeze.dyadic(np.float32(0.03), np.float32(0.02))),
                                   O(np.zeros((3, 96), np.int8))]))
    Ct, Lt = ((96, 64), (128, 49))[V]                                                   # token-order sums: 2 images of Lt tokens
    xt = rng.integers(-9000, 9000, (2 * Lt, Ct)).astype(np.int16)
    wt, bt = rng.uniform(0.4, 1.6, Ct).astype(np.float32), rng.normal(0, 0.3, Ct).astype(np.float32)
    bit, sct = iv.freeze.layernorm_constants(wt, bt)
    dt8, dt16 = iv.freeze.dyadic(sct, np.float32(0.03)), iv.freeze.dyadic(sct, np.float32(2e-4))
    cs.append(("layernorm_tokenorder", [I(xt), 2 * Lt, Ct, 2.5e-4, I(bit), I(sct), Lt, O(np.zeros((2 * Lt, Ct), np.float32))]))
    cs.append(("layernorm_tokenorder_requant", [I(xt), 2 * Lt, Ct, 2.5e-4, I(bit), I(sct), I(dt8), Lt, O(np.zeros((2 * Lt, Ct), np.int8))]))
    cs.append(("patch_norm_tokenorder", [I(rng.integers(-128, 128, (2 * Lt, Ct), dtype=np.int8)), 2 * Lt, Ct, 0.02, I(bit), I(sct), I(dt16),
                                         dyv(iv.freeze.dyadic(np.float32(2e-4), np.float32(2.5e-4))), Lt, O(np.zeros((2 * Lt, Ct), np.int16))]))
    pm = rng.integers(-20000, 20000, (2, 14, 14, 96)).astype(np.int16)
    cs.append(("patch_merge_gather", [I(pm), 16, 2, 14, 96, O(np.zeros((2, 49, 384), np.int16))]))
    cs.append(("patch_merge_gather", [I(rng.integers(-128, 128, (2, 14, 14, 96), dtype=np.int8)), 8, 2, 14, 96, O(np.zeros((2, 49, 384), np.int16))]))
    cs.append(("widen_i8_i16", [I(rng.integers(-128, 128, 5000, dtype=np.int8)), O(np.zeros(5000, np.int16)), 5000]))
    # windowed attention: 2 images of 14 x 14 tokens (2 x 2 windows), 3 heads, with and without the cyclic shift
    Bw, Rw, Hw = ((2, 14, 3), (1, 21, 2))[V]
    qkvw = rng.integers(-128, 128, (Bw, Rw, Rw, 3 * Hw * 32), dtype=np.int8)
    relb = rng.integers(-60, 60, (Hw, 49, 49)).astype(np.int16)
    dwq, dwa, dwp = (iv.freeze.dyadic(np.float32(a), np.float32(b)) for a, b in (((3e-4, 0.05), (0.05, 0.06), (4e-4, 0.03)), ((2.2e-4, 0.043), (0.043, 0.071), (5e-4, 0.026)))[V])
    for sh in (0, 3):
        cs.append(("window_attention_fused", [I(qkvw), dyv(dwq), dyv(dwa), I(relb), (0.06, 0.071)[V], dyv(dwp), O(np.zeros((Bw, Rw * Rw, Hw * 32), np.int8)),
                                              Bw, Rw, 7, sh, Hw, 32]))
    wtab = iv.freeze.shiftmax_tables(np.float32((0.06, 0.071)[V]))
    for sh in (0, 3):
        cs.append(("window_attention_fused_lut", [I(qkvw), dyv(dwq), dyv(dwa), I(relb), (0.06, 0.071)[V], I(wtab["aq"]), I(wtab["t"]), I(wtab["cls"]),
                                                  int(wtab["NC"]), int(wtab["t"].size), int(wtab["dmin"]), dyv(dwp),
                                                  O(np.zeros((Bw, Rw * Rw, Hw * 32), np.int8)), Bw, Rw, 7, sh, Hw, 32]))
    # ---- the uint8 front end (N3): ToTensor -> Normalize -> input QuantAct; antialiased bicubic resize + centre crop
    u8 = rng.integers(0, 256, (2, 37, 53, 3), dtype=np.uint8)
    u8.reshape(-1)[:256] = np.arange(256, dtype=np.uint8)
    mean, std = np.array([0.485, 0.456, 0.406], np.float32), np.array([0.229, 0.224, 0.225], np.float32)
    cs.append(("normalize_quantize_u8", [I(u8), 2, 37, 53, ("host", mean), ("host", std), 0.0207, O(np.zeros((2, 3, 37, 53), np.int8))]))
    big = rng.integers(0, 256, (2, 60, 83, 3), dtype=np.uint8)
    cs.append(("resize_center_crop_u8", [I(big), 2, 60, 83, 40, 32, O(np.zeros((2, 60, 32, 3), np.float32)), O(np.zeros((2, 32, 32, 3), np.uint8))]))
    tall = rng.integers(0, 256, (1, 75, 50, 3), dtype=np.uint8)                        # portrait, upscaling
    cs.append(("resize_center_crop_u8", [I(tall), 1, 75, 50, 64, 56, O(np.zeros((1, 75, 56, 3), np.float32)), O(np.zeros((1, 56, 56, 3), np.uint8))]))
    return cs


def _run(fn, handle, args, to_ptr):
    outs, call = [], [handle]
    for a in args:
        if isinstance(a, tuple) and a[0] == "host":          # a HOST array on both sides (mean / std of the normalisation)
            call.append(a[1].ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
        elif isinstance(a, tuple):
            buf = to_ptr(a[1], a[0] == "out")
            call.append(buf[0])
            if a[0] == "out":
                outs.append(buf[1])
        else:
            call.append(a)
    st = fn(*call)
    assert st == 0, st
    return outs


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [0, 1])
def test_every_twinned_entry_point_agrees_with_the_hip_library(twin, variant):
    import torch
    H = _lib.Handle(0, torch.cuda.current_stream().cuda_stream)
    keep = []

    def to_dev(a, is_out):
        t = torch.from_numpy(a.copy()).cuda()
        keep.append(t)
        return _P(t.data_ptr()), t

    def to_host(a, is_out):
        h = a.copy()
        keep.append(h)
        return hp(h), h

    cases = _cases(np.random.default_rng(77 + variant), variant)
    seen = set()
    for name, args in cases:
        seen.add(name)
        gfn = getattr(H.lib, "ivit_" + name)
        gfn.argtypes = _sig(name)
        gout = _run(gfn, H.h, args, to_dev)
        cout = _run(getattr(twin, "ivit_cpu_" + name), None, args, to_host)
        torch.cuda.synchronize()
        for i, (g, c) in enumerate(zip(gout, cout)):
            g = g.cpu().numpy()
            if name == "shiftgelu_build_table":       # entries with Q > row max are never indexed (twin leaves them 0)
                idx = np.arange(65536)
                valid = (idx & 255) <= (idx >> 8)
                g, c = g[valid], c[valid]
            Tv = (50, 37)[variant]
            if name == "linear_i8_qkv" and i == 2:    # pad columns of v^T (t >= T) are not written by either side
                g, c = g[:, :, :Tv], c[:, :, :Tv]
            if name == "attn_qk_requant":
                g, c = g[:, :, :Tv], c[:, :, :Tv]
            assert np.array_equal(g, c), (name, i, int((g != c).sum()))
    # planned entry points: same plan-handle protocol on both sides
    rng = np.random.default_rng(5 + variant)
    M, N, K = ((700, 384, 384), (1291, 768, 384))[variant]
    x = rng.integers(-128, 128, (M, K), dtype=np.int8)
    w = np.rint(rng.normal(0, 40, (N, K)).clip(-127, 127)).astype(np.int8)
    b = rng.integers(-3000, 3000, N).astype(np.int32)
    d = iv.freeze.dyadic((10 ** rng.uniform(-5.5, -5.0, N)).astype(np.float32), np.float32(3e-2))
    xd, wd, bd, dd = (torch.from_numpy(a).cuda() for a in (x, w, b, d))
    pg, pc = _P(), _P()
    f = H.lib.ivit_linear_plan_create
    f.argtypes = _lib.SIGNATURES["ivit_linear_plan_create"]
    assert f(H.h, _P(wd.data_ptr()), _P(bd.data_ptr()), _P(dd.data_ptr()), N, K, ctypes.byref(pg)) == 0
    assert twin.ivit_cpu_linear_plan_create(None, hp(w), hp(b), hp(d), N, K, ctypes.byref(pc)) == 0
    og, oc = torch.zeros(M, N, dtype=torch.int8, device="cuda"), np.zeros((M, N), np.int8)
    f = H.lib.ivit_linear_i8_requant_planned
    f.argtypes = _lib.SIGNATURES["ivit_linear_i8_requant_planned"]
    assert f(H.h, pg, _P(xd.data_ptr()), 8, _P(og.data_ptr()), M) == 0
    assert twin.ivit_cpu_linear_i8_requant_planned(None, pc, hp(x), 8, hp(oc), M) == 0
    assert np.array_equal(og.cpu().numpy(), oc)
    H.lib.ivit_linear_plan_destroy.argtypes = [_P]
    assert H.lib.ivit_linear_plan_destroy(pg) == 0 and twin.ivit_cpu_linear_plan_destroy(pc) == 0
    # planned fused Mlp (D = 384): linear plans -> Mlp plan -> one call, on both sides
    Mm, Cm, Hm = (333, 2051)[variant], 384, 1536       # 2051 = 128 full tiles + 3 rows: several units per workgroup shape
    xm = rng.integers(-128, 128, (Mm, Cm), dtype=np.int8)
    w1 = np.rint(rng.normal(0, 40, (Hm, Cm)).clip(-127, 127)).astype(np.int8); b1 = rng.integers(-2000, 2000, Hm).astype(np.int32)
    w2 = np.rint(rng.normal(0, 40, (Cm, Hm)).clip(-127, 127)).astype(np.int8); b2 = rng.integers(-2000, 2000, Cm).astype(np.int32)
    d1 = iv.freeze.dyadic((10 ** rng.uniform(-5.6, -5.3, Hm)).astype(np.float32), np.float32(0.04))
    d2 = iv.freeze.dyadic((10 ** rng.uniform(-5.9, -5.6, Cm)).astype(np.float32), np.float32(2e-4))
    dmm, drr = iv.freeze.dyadic(np.float32(2e-4), np.float32(2.5e-4)), iv.freeze.dyadic(np.float32(3e-4), np.float32(2.5e-4))
    resm = rng.integers(-20000, 20000, (Mm, Cm)).astype(np.int16)
    tabm = np.zeros(65536, np.int8)
    dgm = iv.freeze.dyadic(np.float32(0.04 * 2.0 ** -7), np.float32(0.03))
    assert twin.ivit_cpu_shiftgelu_build_table(None, 0.04, dyv(dgm), hp(tabm)) == 0
    tabd = torch.empty(65536, dtype=torch.int8, device="cuda")
    H.call("ivit_shiftgelu_build_table", 0.04, dyv(dgm), _P(tabd.data_ptr()))
    dev = {k: torch.from_numpy(v).cuda() for k, v in dict(x=xm, w1=w1, b1=b1, w2=w2, b2=b2, d1=d1, d2=d2, res=resm).items()}
    g1, g2, gm, c1, c2, cm = (_P() for _ in range(6))
    f = H.lib.ivit_linear_plan_create
    assert f(H.h, _P(dev["w1"].data_ptr()), _P(dev["b1"].data_ptr()), _P(dev["d1"].data_ptr()), Hm, Cm, ctypes.byref(g1)) == 0
    assert f(H.h, _P(dev["w2"].data_ptr()), _P(dev["b2"].data_ptr()), _P(dev["d2"].data_ptr()), Cm, Hm, ctypes.byref(g2)) == 0
    assert twin.ivit_cpu_linear_plan_create(None, hp(w1), hp(b1), hp(d1), Hm, Cm, ctypes.byref(c1)) == 0
    assert twin.ivit_cpu_linear_plan_create(None, hp(w2), hp(b2), hp(d2), Cm, Hm, ctypes.byref(c2)) == 0
    H.call("ivit_mlp_plan_create", g1, g2, ctypes.byref(gm))
    assert twin.ivit_cpu_mlp_plan_create(None, c1, c2, ctypes.byref(cm)) == 0
    assert twin.ivit_cpu_mlp_plan_create(None, c2, c1, ctypes.byref(_P())) == 3          # other shapes: unsupported on both sides
    og, oc = torch.zeros(Mm, Cm, dtype=torch.int16, device="cuda"), np.zeros((Mm, Cm), np.int16)
    H.call("ivit_mlp_fused_planned", gm, _P(dev["x"].data_ptr()), _P(tabd.data_ptr()), dyv(dmm), dyv(drr), _P(dev["res"].data_ptr()),
           _P(og.data_ptr()), Mm)
    assert twin.ivit_cpu_mlp_fused_planned(None, cm, hp(xm), hp(tabm), dyv(dmm), dyv(drr), hp(resm), hp(oc), Mm) == 0
    assert np.array_equal(og.cpu().numpy(), oc)
    assert H.lib.ivit_mlp_plan_destroy(gm) == 0 and twin.ivit_cpu_mlp_plan_destroy(cm) == 0
    for pl in (g1, g2):
        assert H.lib.ivit_linear_plan_destroy(pl) == 0
    for pl in (c1, c2):
        assert twin.ivit_cpu_linear_plan_destroy(pl) == 0
    # every twinned single-call entry point was exercised above (plans and the LUT forms have their own protocol)
    rest = set(gen_twin_header.TWIN) - seen - {"linear_plan_create", "linear_plan_destroy", "linear_plan_query",
                                                "linear_i8_requant_planned", "linear_i8_requant_residual_planned",
                                                "linear_i8_qkv_planned", "attention_fused_lut", "shiftgelu_requant_lut", "mlp_fused",
                                                "mlp_plan_create", "mlp_plan_destroy", "mlp_fused_planned",
                                                # the whole-model runners: test_twin_runners_agree_with_the_hip_runners
                                                "vit_create", "vit_destroy", "vit_workspace_bytes", "vit_workspace_init", "vit_forward",
                                                "swin_create", "swin_destroy", "swin_workspace_bytes", "swin_forward"}
    assert not rest, rest


def test_twin_composites_match_the_oracle_chain(twin):
    """fused entry points of the twin against the oracle's own Python-level chaining (tests/test_oracle_golden.py pins that
    chaining to the reference)"""
    from oracle import oracle as orc
    rng = np.random.default_rng(3)
    M, C, Hd = 70, 96, 384
    x = rng.integers(-128, 128, (M, C), dtype=np.int8)
    w1 = np.rint(rng.normal(0, 40, (Hd, C)).clip(-127, 127)).astype(np.int8)
    w2 = np.rint(rng.normal(0, 40, (C, Hd)).clip(-127, 127)).astype(np.int8)
    b1, b2 = rng.integers(-2000, 2000, Hd).astype(np.int32), rng.integers(-2000, 2000, C).astype(np.int32)
    s1 = (10 ** rng.uniform(-5.3, -5.0, Hd)).astype(np.float32)
    s2 = (10 ** rng.uniform(-5.3, -5.0, C)).astype(np.float32)
    s_h, s_g, s_t, s_res, s_fin = (np.float32(v) for v in (0.04, 0.03, 2e-4, 3e-4, 2.5e-4))
    res = rng.integers(-20000, 20000, (M, C)).astype(np.int16)
    d1, d2 = iv.freeze.dyadic(s1, s_h), iv.freeze.dyadic(s2, s_t)
    dg = iv.freeze.dyadic(np.float32(s_h * np.float32(2.0 ** -7)), s_g)
    dm, dr = iv.freeze.dyadic(s_t, s_fin), iv.freeze.dyadic(s_res, s_fin)
    tab = np.zeros(65536, np.int8)
    assert twin.ivit_cpu_shiftgelu_build_table(None, float(s_h), dyv(dg), hp(tab)) == 0
    out = np.zeros((M, C), np.int16)
    assert twin.ivit_cpu_mlp_fused(None, hp(x), hp(w1), hp(b1), hp(d1), hp(tab), hp(w2), hp(b2), hp(d2), dyv(dm), dyv(dr),
                                   hp(res), hp(out), M, C, Hd) == 0
    h8 = orc.requant(orc.linear_i8(x, w1, b1), orc.dyadic(s1, s_h), 8).astype(np.int8)
    g8 = orc.requant(orc.shiftgelu(h8, s_h).astype(np.int32), orc.dyadic(np.float32(s_h * np.float32(2.0 ** -7)), s_g), 8).astype(np.int8)
    t = orc.requant(orc.linear_i8(g8, w2, b2), orc.dyadic(s2, s_t), 16)
    ref = orc.requant(t, orc.dyadic(s_t, s_fin), 16, res.astype(np.int32), orc.dyadic(s_res, s_fin))
    assert np.array_equal(out.astype(np.int32), ref)


def test_twin_front_end_matches_oracle_and_torch(twin):
    """N3 twins on the CPU: ivit_cpu_resize_center_crop_u8 == oracle.resize_center_crop_u8 (the restatement that
    tests/golden/resize.npz pins to torch's antialiased bicubic) on every fixture case; ivit_cpu_normalize_quantize_u8 == the
    ToTensor -> Normalize -> QuantAct chain in torch fp32 for every pixel value."""
    import torch
    from oracle import oracle as orc
    from conftest import load_golden
    g = load_golden("resize.npz")
    for ci in range(int(g["n"])):
        size, crop = [int(v) for v in g[f"cfg/{ci}"]]
        img = np.ascontiguousarray(g[f"in/{ci}"])
        B, H0, W0, _ = img.shape
        ws, out = np.zeros((B, H0, crop, 3), np.float32), np.zeros((B, crop, crop, 3), np.uint8)
        assert twin.ivit_cpu_resize_center_crop_u8(None, hp(img), B, H0, W0, size, crop, hp(ws), hp(out)) == 0
        assert np.array_equal(out, orc.resize_center_crop_u8(img, size, crop)), ci
    rng = np.random.default_rng(3)
    u = rng.integers(0, 256, (2, 19, 23, 3), dtype=np.uint8)
    u.reshape(-1)[:256] = np.arange(256, dtype=np.uint8)
    mean, std, scale = np.array([0.485, 0.456, 0.406], np.float32), np.array([0.229, 0.224, 0.225], np.float32), 0.0207
    q = np.zeros((2, 3, 19, 23), np.int8)
    fp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
    assert twin.ivit_cpu_normalize_quantize_u8(None, hp(u), 2, 19, 23, fp(mean), fp(std), scale, hp(q)) == 0
    t = torch.from_numpy(u).permute(0, 3, 1, 2).float().div(255)
    x = t.sub(torch.from_numpy(mean).view(1, 3, 1, 1)).div(torch.from_numpy(std).view(1, 3, 1, 1))
    inv = np.float32(1.0) / np.float32(scale)
    ref = torch.clamp(torch.round(x * float(inv)), -128, 127).to(torch.int8).numpy()
    assert np.array_equal(q, ref)


def _twin_vit(twin, g, images):
    """ivit_cpu_vit_create / _forward on host constants packed exactly as ViTEngine packs them for the device"""
    from ivit_amd import engine as E
    cfg = iv.CONFIGS[str(g["cfg_name"])]
    consts, f32 = E.freeze_vit(cfg, iv.make_vit_weights(cfg, int(g["seed"])), golden_scales(g))
    blob, table = E.pack_constants(consts)
    f32 = {k: float(np.float32(v)) for k, v in f32.items()}
    c, prm, keep = E.vit_native_params(cfg, table, f32, E.host_scalars(blob, table), blob.ctypes.data)
    m = _P()
    assert twin.ivit_cpu_vit_create(None, ctypes.byref(c), ctypes.byref(prm), 1, ctypes.byref(m)) == 0
    n = ctypes.c_size_t()
    B = images.shape[0]
    assert twin.ivit_cpu_vit_workspace_bytes(m, B, 1, ctypes.byref(n)) == 0
    ws = np.zeros(n.value, np.uint8)
    assert twin.ivit_cpu_vit_workspace_init(m, hp(ws), n.value, B, 1) == 0
    logits = np.zeros((B, cfg.num_classes), np.int32)
    assert twin.ivit_cpu_vit_forward(m, hp(np.ascontiguousarray(images)), B, 1, hp(ws), n.value, hp(logits)) == 0
    assert twin.ivit_cpu_vit_destroy(m) == 0
    return logits


def _twin_swin(twin, g, images, exp_tables=False):
    from ivit_amd import swin_engine as S
    cfg = iv.SWIN_CONFIGS[str(g["cfg_name"])]
    blob, table, host = S.pack_swin_constants(S.freeze_swin(cfg, iv.make_swin_weights(cfg, int(g["seed"])), golden_scales(g), exp_tables))
    f, dy = S.swin_host_scalars(host)
    c, prm, keep = S.swin_native_params(cfg, table, f, dy, blob.ctypes.data)
    m = _P()
    assert twin.ivit_cpu_swin_create(None, ctypes.byref(c), ctypes.byref(prm), 1, ctypes.byref(m)) == 0
    n = ctypes.c_size_t()
    B = images.shape[0]
    assert twin.ivit_cpu_swin_workspace_bytes(m, B, 1, ctypes.byref(n)) == 0
    ws = np.zeros(n.value, np.uint8)
    logits = np.zeros((B, cfg.num_classes), np.int32)
    assert twin.ivit_cpu_swin_forward(m, hp(np.ascontiguousarray(images)), B, 1, hp(ws), n.value, hp(logits)) == 0
    assert twin.ivit_cpu_swin_destroy(m) == 0
    return logits


@pytest.mark.parametrize("fname", ["micro_vit_b2.npz", "micro_vit2h_b3.npz", "deit_tiny_b1.npz", "deit_small_b4.npz"])
def test_twin_vit_runner_matches_reference_golden(twin, fname):
    """the whole-model runner's twin (ivit_cpu_vit_*: same structs as ivit_vit_create, host pointers) == the reference's
    int32 logits, on the CPU"""
    from conftest import load_golden
    g = load_golden(fname)
    cfg = iv.CONFIGS[str(g["cfg_name"])]
    imgs = iv.make_images_int8(cfg, int(g["batch"]), int(g["images_seed"]))
    assert np.array_equal(_twin_vit(twin, g, imgs), g["logits_int"])


@pytest.mark.parametrize("fname", ["micro_swin_b2.npz", "swin_tiny_b1.npz"])
def test_twin_swin_runner_matches_reference_golden(twin, fname):
    from conftest import load_golden
    g = load_golden(fname)
    cfg = iv.SWIN_CONFIGS[str(g["cfg_name"])]
    imgs = iv.make_images_int8(cfg, int(g["batch"]), int(g["images_seed"]))
    assert np.array_equal(_twin_swin(twin, g, imgs), g["logits_int"])
    if fname.startswith("micro"):       # and with the per-layer Shiftmax tables in the block structs (ivit_cpu_window_attention_fused_lut)
        assert np.array_equal(_twin_swin(twin, g, imgs, exp_tables=True), g["logits_int"])


@pytest.mark.gpu
def test_twin_runners_agree_with_the_hip_runners(twin):
    """ivit_vit_forward / ivit_swin_forward (device) == ivit_cpu_vit_forward / ivit_cpu_swin_forward (host) on images the
    goldens do not contain; the device side also sliced"""
    import torch
    from conftest import load_golden
    from ivit_amd.engine import ViTEngine
    from ivit_amd.swin_engine import SwinEngine
    g = load_golden("micro_vit2h_b3.npz")
    cfg = iv.CONFIGS[str(g["cfg_name"])]
    imgs = iv.make_images_int8(cfg, 7, seed=4242)
    eng = ViTEngine.from_float(cfg, iv.make_vit_weights(cfg, int(g["seed"])), golden_scales(g))
    d = torch.from_numpy(imgs).cuda()
    ref = _twin_vit(twin, g, imgs)
    assert np.array_equal(eng.forward(d).cpu().numpy(), ref)
    assert np.array_equal(eng.forward(d, nslices=3).cpu().numpy(), ref)
    g = load_golden("micro_swin_b2.npz")
    cfg = iv.SWIN_CONFIGS[str(g["cfg_name"])]
    imgs = iv.make_images_int8(cfg, 5, seed=4243)
    eng = SwinEngine(cfg, iv.make_swin_weights(cfg, int(g["seed"])), golden_scales(g))
    d = torch.from_numpy(imgs).cuda()
    ref = _twin_swin(twin, g, imgs)
    assert np.array_equal(eng.forward(d).cpu().numpy(), ref)
    assert np.array_equal(eng.forward(d, nslices=2).cpu().numpy(), ref)
