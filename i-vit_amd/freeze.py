"""Host-side constant preparation for the frozen integer model ("inference mode").

The reference recomputes these every forward (weight min/max/round,
quant_modules.py:68-91; a host Decimal loop per QuantAct, quant_utils.py:150-175).
Here they are computed once when the model is frozen (reference
models/model_utils.py:5-21 + QuantAct.fix, quant_modules.py:153-157) and uploaded.
All arithmetic is numpy float32/float64 with the same operation order as the
reference so the integers agree bit for bit.
"""
import math

import numpy as np

F32_EPS = np.float32(np.finfo(np.float32).eps)


def symmetric_scale(min_val, max_val, num_bits):
    """quant_utils.py:51-69: s = max(max(-min, max) / n, eps), fp32."""
    n = np.float32(2 ** (num_bits - 1) - 1)
    m = np.maximum(-np.asarray(min_val, np.float32), np.asarray(max_val, np.float32))
    return np.maximum((m / n).astype(np.float32), F32_EPS)


def quantize(x, scale, num_bits, per_out_channel):
    """quant_utils.py:12-48 + 77-96: clamp(rne(fl(fl(1/s) * x))) as float64 integers."""
    x = np.asarray(x, np.float32)
    inv = (np.float32(1.0) / np.asarray(scale, np.float32)).astype(np.float32)
    if per_out_channel:
        inv = inv.reshape((-1,) + (1,) * (x.ndim - 1))
    n = 2 ** (num_bits - 1) - 1
    q = np.rint((inv * x).astype(np.float32))
    return np.clip(q, np.float32(-n - 1), np.float32(n)).astype(np.float64)


def quantize_weight(w):
    """Per-output-channel 8-bit weights (quant_modules.py:68-83 / 303-320)."""
    w = np.asarray(w, np.float32)
    flat = w.reshape(w.shape[0], -1)
    s_w = symmetric_scale(flat.min(axis=1), flat.max(axis=1), 8)
    return quantize(w, s_w, 8, True).astype(np.int8).reshape(w.shape), s_w


def quantize_bias(b, s_w, s_x):
    """32-bit bias on scale fl(s_w * s_x) (quant_modules.py:85-89)."""
    s_b = (np.asarray(s_w, np.float32) * np.float32(s_x)).astype(np.float32)
    q = quantize(b, s_b, 32, True)
    if np.any(np.abs(q) >= 2.0 ** 31):
        raise OverflowError("bias does not fit int32")
    return q.astype(np.int32), s_b


def dyadic(s_pre, s_out):
    """(m, 2^-e) pairs, shape [n, 2] float64 (quant_utils.py:150-175, 221-227).
    m = round-half-away-from-zero(mantissa * 2^31), e = 31 - exponent of
    double(s_pre) / double(float32(s_out))."""
    pre = np.atleast_1d(np.asarray(s_pre, np.float32)).astype(np.float64)
    ratio = pre / np.float64(np.float32(s_out))
    out = np.empty((ratio.size, 2), np.float64)
    for i, v in enumerate(ratio):
        mant, ex = math.frexp(float(v))
        num, den = (mant * 2147483648.0).as_integer_ratio()  # exact
        m = (2 * abs(num) + den) // (2 * den)
        out[i, 0] = -float(m) if num < 0 else float(m)
        out[i, 1] = 2.0 ** (ex - 31)
    return out


def layernorm_constants(weight, bias):
    """bias_int[c] = floor(fl(fl(b/w)/sf)), sc[c] = fl(sf*w[c]), sf = fl(sqrt(C))/2^30
    (quant_modules.py:354-357, 374-383)."""
    w = np.asarray(weight, np.float32)
    b = np.asarray(bias, np.float32)
    sf = np.float32(np.sqrt(np.float32(w.size)) / np.float32(2.0 ** 30))
    bias_int = np.floor(((b / w).astype(np.float32) / sf).astype(np.float32)).astype(np.float32)
    return bias_int, (sf * w).astype(np.float32)


def freeze_vit(cfg, weights, scales):
    """All integer constants of a frozen DeiT/ViT (reference models/vit_quant.py).
    Returns a flat dict name -> numpy array (int8/int32/int16/float32/float64[.,2])."""
    s = {k: np.float32(v) for k, v in scales.items()}
    D = cfg.embed_dim
    c = {}

    def linear(prefix, s_in, s_out_site, store=None):
        wq, s_w = quantize_weight(weights[prefix + ".weight"])
        bq, s_b = quantize_bias(weights[prefix + ".bias"], s_w, s_in)
        store = store or prefix
        c[store + ".w"] = wq.reshape(wq.shape[0], -1)
        c[store + ".b"] = bq
        if s_out_site is not None:
            c[store + ".dy"] = dyadic(s_b, s[s_out_site])
        return s_b

    def norm(prefix, s_out_site):
        bi, sc = layernorm_constants(weights[prefix + ".weight"], weights[prefix + ".bias"])
        c[prefix + ".bias_int"] = bi
        c[prefix + ".sc"] = sc
        c[prefix + ".dy"] = dyadic(sc, s[s_out_site])

    linear("patch_embed.proj", s["qact_input"], "patch_embed.qact")
    cls = weights["cls_token"].reshape(-1).astype(np.float32)
    c["z_cls"] = np.rint((cls / s["patch_embed.qact"]).astype(np.float32)).astype(np.int32)
    c["pos"] = quantize(weights["pos_embed"][0], s["qact_pos"], 16, False).astype(np.int16)
    c["embed.dy_x"] = dyadic(s["patch_embed.qact"], s["qact1"])
    c["embed.dy_pos"] = dyadic(s["qact_pos"], s["qact1"])
    f32 = {"s_in": s["qact_input"]}
    s_x = s["qact1"]
    head_scale = np.float32(cfg.head_dim ** -0.5)
    for i in range(cfg.depth):
        p = f"blocks.{i}."
        f32[p + "ln1.s"] = s_x
        norm(p + "norm1", p + "qact1")
        linear(p + "attn.qkv", s[p + "qact1"], p + "attn.qact1")
        s1 = s[p + "attn.qact1"]
        c[p + "attn.dy_qk"] = dyadic(np.float32(np.float32(s1 * s1) * head_scale), s[p + "attn.qact_attn1"])
        f32[p + "attn.s_softmax"] = s[p + "attn.qact_attn1"]
        tabs = shiftmax_tables(s[p + "attn.qact_attn1"])
        if tabs is not None:                       # else: the kernel's arithmetic path for this layer
            c[p + "attn.exp_aq"], c[p + "attn.exp_t"], c[p + "attn.exp_cls"] = tabs["aq"], tabs["t"], tabs["cls"]
            c[p + "attn.exp_meta"] = np.array([tabs["NC"], tabs["t"].size, tabs["dmin"]], np.int32)
        c[p + "attn.dy_pv"] = dyadic(np.float32(np.float32(2.0 ** -15) * s1), s[p + "attn.qact2"])
        linear(p + "attn.proj", s[p + "attn.qact2"], p + "attn.qact3")
        c[p + "res1.dy_main"] = dyadic(s[p + "attn.qact3"], s[p + "qact2"])
        c[p + "res1.dy_res"] = dyadic(s_x, s[p + "qact2"])
        s_x = s[p + "qact2"]
        f32[p + "ln2.s"] = s_x
        norm(p + "norm2", p + "qact3")
        linear(p + "mlp.fc1", s[p + "qact3"], p + "mlp.qact_gelu")
        f32[p + "mlp.s_gelu"] = s[p + "mlp.qact_gelu"]
        c[p + "mlp.dy_gelu"] = dyadic(np.float32(s[p + "mlp.qact_gelu"] * np.float32(2.0 ** -7)), s[p + "mlp.qact1"])
        linear(p + "mlp.fc2", s[p + "mlp.qact1"], p + "mlp.qact2")
        c[p + "res2.dy_main"] = dyadic(s[p + "mlp.qact2"], s[p + "qact4"])
        c[p + "res2.dy_res"] = dyadic(s_x, s[p + "qact4"])
        s_x = s[p + "qact4"]
    f32["ln.s"] = s_x
    norm("norm", "qact2")
    c["head.scale"] = linear("head", s["qact2"], None)
    return c, f32


# ---------------------------------------------------------------------------------------------------
# Shiftmax as a table (IntSoftmax.int_exp_shift, quant_modules.py:469-481, for a FROZEN input scale s).
# A score enters as an 8-bit integer v; the kernel sees x~ = fl(fl(v*s)/s) = v + delta(v), where delta(v) is
# 0 for ~90 % of v and one of a handful of tiny values otherwise.  The exponent argument is
# x = fl(x~(v) - x~(vmax)) = fl((v - vmax) + (delta(v) - delta(vmax))): a function of the integer distance
# d = v - vmax and of eps = delta(v) - delta(vmax), which takes at most a few dozen values per scale.  So
# exp_int is a table E[eps class][d], a few KB per layer, and the row of classes for a given vmax class is a
# second small table.  Everything is enumerated with the reference's fp32 operation sequence and then CHECKED
# against the direct formula for every (vmax, v <= vmax) pair; a layer whose table would not fit is run with
# the arithmetic path instead.
def _shift_exp_f32(x, s, n=15):
    f32 = np.float32
    x = np.asarray(x, f32)
    x0 = np.floor(f32(-1.0) / f32(s)).astype(f32)
    nx0 = f32(n) * x0
    t = (x + np.floor(x * f32(0.5))).astype(f32)
    t = (t - np.floor(x * f32(0.0625))).astype(f32)
    t = np.maximum(t, nx0)
    q = np.floor((t / x0).astype(f32))
    r = (t - (x0 * q).astype(f32)).astype(f32)
    e = ((r * f32(0.5)).astype(f32) - x0).astype(f32)
    e = np.floor(np.ldexp(e, (n - q.astype(np.int64)).astype(np.int32))).astype(f32)
    return np.maximum(e, f32(0))


def shiftmax_tables(s, max_bytes=24 * 1024):
    """-> dict(cls uint8[256], aq uint16[NC,256], t float32[NE*R], R, dmin) or None if the tables exceed max_bytes."""
    f32 = np.float32
    s = f32(s)
    v = np.arange(-128, 128).astype(f32)
    f = ((v * s).astype(f32) / s).astype(f32)                 # x~(v)
    delta = f.astype(np.float64) - v.astype(np.float64)       # exact
    dvals = np.unique(delta)
    cls = np.searchsorted(dvals, delta).astype(np.uint8)
    NC = len(dvals)
    eps_vals = np.unique(dvals[:, None] - dvals[None, :])
    NE = len(eps_vals)
    d_all = np.arange(-255, 1)
    # E over every (eps, d); the argument is the single fp32 rounding of the exact real d + eps
    X = (d_all[None, :].astype(np.float64) + eps_vals[:, None]).astype(f32)
    E = _shift_exp_f32(X, s)                                   # [NE, 256]
    # positive arguments (d = 0, eps > 0) cannot occur: vmax maximises x~ — they are never indexed
    const = E[:, 0]
    if not np.all(const == const[0]):
        return None
    same = np.all(E == const[0], axis=0)                       # per d: every eps class at the floor value
    k = 0
    while k + 1 < 256 and same[k + 1]:
        k += 1
    dmin = int(d_all[k])
    R = -dmin + 1
    if NE * R * 4 + NC * 512 + 256 > max_bytes or NE * R >= 65536:
        return None
    T = np.ascontiguousarray(E[:, k:]).astype(f32)              # [NE, R], column dd = d - dmin
    eid = np.searchsorted(eps_vals, dvals[:, None] - dvals[None, :])      # [class of v][class of vmax]
    aq = (eid[cls][:, :].T.astype(np.int64) * R).astype(np.uint16)        # [NC (vmax class), 256 (v)]
    # exhaustive check against the direct formula
    vi = np.arange(256)
    for qi in range(256):
        x = (f[: qi + 1] - f[qi]).astype(f32)
        direct = _shift_exp_f32(x, s)
        dd = np.maximum(vi[: qi + 1] - qi, dmin) - dmin
        tab = T.reshape(-1)[aq[cls[qi], : qi + 1].astype(np.int64) + dd]
        if not np.array_equal(direct, tab):
            raise AssertionError(f"shiftmax table mismatch at vmax index {qi} for scale {float(s)!r}")
    return dict(cls=cls, aq=np.ascontiguousarray(aq), t=T.reshape(-1), R=R, dmin=dmin, NC=NC, NE=NE)


def shiftmax_rowtable(tabs):
    """Row form of the Shiftmax tables (csrc/ivit_attention.h, LUT = 2): rowtab[vmax + 128][dd] = exp_int of the score
    v = vmax + dmin + dd in a row whose maximum is vmax, dd = max(v - vmax, dmin) - dmin; float32 [256, 64].  None when a line
    does not fit 64 entries (R = 1 - dmin > 64).  The device builds the same table itself (ivit_shiftmax_rowtable); this copy
    serves the tests."""
    if tabs is None or tabs["R"] > 64:
        return None
    R, dmin = int(tabs["R"]), int(tabs["dmin"])
    aq, t, cls = tabs["aq"].astype(np.int64), tabs["t"], tabs["cls"].astype(np.int64)
    out = np.zeros((256, 64), np.float32)
    for q in range(256):
        for dd in range(R):
            vi = q + dmin + dd
            if vi < 0:
                if dd:
                    continue                      # a score below -128: never indexed
                vi = 0                            # entry 0 is the floor constant whatever the class
            out[q, dd] = t[aq[cls[q], vi] + dd]
    return out
