"""Generate the golden fixtures under tests/golden/ from the imported reference.

Runs ONLY in the build container (needs /root/reference + torch CPU).  The
fixtures are data: integer inputs, scales and the integer outputs the reference
produced.  Weights are not stored — they are re-drawn from ivit_amd.synth with the
recorded seed (a checksum guards against generator drift).

    python tools/make_golden.py            # writes tests/golden/*.npz
"""
import hashlib
import os
import sys

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, _HERE)
import ivit_amd as iv  # noqa: E402
import ref_harness as rh  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def weights_digest(w):
    h = hashlib.sha256()
    for k in sorted(w):
        h.update(k.encode())
        h.update(np.ascontiguousarray(w[k]).tobytes())
    return h.hexdigest()


def csum(a):
    """order-sensitive 64-bit checksum of an integer array (as int64)."""
    a = np.asarray(a).astype(np.int64).reshape(-1)
    idx = np.arange(1, a.size + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        return np.uint64(((a.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)) ^ idx).sum())


def site_value(r):
    if r["type"] in ("QuantLinear", "QuantConv2d", "QuantMatMul"):
        return r["acc"]
    if r["type"] == "IntLayerNorm":
        return r["z"]
    return r["out"]


def model_fixture(models, cfg_name, batch, calib_batch, seed, full_sites, fname):
    cfg = iv.CONFIGS[cfg_name]
    w = iv.make_vit_weights(cfg, seed)
    m = rh.build_ref_vit(models, cfg, w)
    rh.calibrate_and_freeze(models, m, iv.make_calibration_batch(cfg, calib_batch))
    with torch.no_grad():
        m(torch.zeros(1, 3, cfg.img_size, cfg.img_size))
    sc = rh.act_scales(models, m)
    q = iv.make_images_int8(cfg, batch)
    y, recs = rh.capture(models, m, q.astype(np.float32) * sc["qact_input"])
    d = {"cfg_name": cfg_name, "seed": seed, "batch": batch, "images_seed": 1,
         "weights_sha256": weights_digest(w)}
    for k, v in sc.items():
        d["scale/" + k] = np.float32(v)
    names = []
    for r in recs:
        n = r["name"]
        if n in ("qact_input", "qact_pos"):
            continue
        v = site_value(r)
        if n == "patch_embed.proj":
            v = v.reshape(v.shape[0], v.shape[1], -1).transpose(0, 2, 1)
        if n == "norm":
            v = v[:, 0]
        names.append(n)
        if r["type"] == "IntLayerNorm":
            v = np.asarray(v, np.float64)
        d["csum/" + n] = csum(v)
        if full_sites:
            v = np.asarray(v)
            if r["type"] == "IntLayerNorm":
                d["site/" + n] = v.astype(np.float32)
            else:
                mx = np.abs(v).max() if v.size else 0
                dt = np.int8 if mx < 128 else (np.int16 if mx < 32768 else np.int32)
                if r["type"] == "IntSoftmax":
                    dt = np.uint16
                d["site/" + n] = v.astype(dt)
    d["sites"] = np.array(names)
    d["logits_int"] = recs[-1]["acc"].astype(np.int32)
    d["logits_scale"] = recs[-1]["s_out"]
    d["logits_fp32"] = y.astype(np.float32)
    np.savez_compressed(os.path.join(OUT, fname), **d)
    print(fname, "sites", len(names), "bytes", os.path.getsize(os.path.join(OUT, fname)))


def swin_fixture(models, cfg_name, batch, calib_batch, seed, full_sites, fname):
    cfg = iv.SWIN_CONFIGS[cfg_name]
    w = iv.make_swin_weights(cfg, seed)
    m = rh.build_ref_swin(models, cfg, w)
    rh.calibrate_and_freeze(models, m, iv.make_calibration_batch(cfg, calib_batch))
    with torch.no_grad():
        m(torch.zeros(1, 3, cfg.img_size, cfg.img_size))
    sc = rh.act_scales(models, m)
    q = iv.make_images_int8(cfg, batch)
    y, recs = rh.capture(models, m, q.astype(np.float32) * sc["qact_input"])
    d = {"cfg_name": cfg_name, "seed": seed, "batch": batch, "images_seed": 1, "weights_sha256": weights_digest(w)}
    for k, v in sc.items():
        d["scale/" + k] = np.float32(v)
    names = []
    for r in recs:
        n = r["name"]
        if n == "qact_input":
            continue
        v = site_value(r)
        if n == "patch_embed.proj":
            v = v.reshape(v.shape[0], v.shape[1], -1).transpose(0, 2, 1)
        names.append(n)
        if r["type"] == "IntLayerNorm":
            v = np.asarray(v, np.float64)
        d["csum/" + n] = csum(v)
        if full_sites:
            v = np.asarray(v)
            if r["type"] == "IntLayerNorm":
                d["site/" + n] = v.astype(np.float32)
            else:
                mx = np.abs(v).max() if v.size else 0
                dt = np.int8 if mx < 128 else (np.int16 if mx < 32768 else np.int32)
                if r["type"] == "IntSoftmax":
                    dt = np.uint16
                d["site/" + n] = v.astype(dt)
    d["sites"] = np.array(names)
    d["logits_int"] = recs[-1]["acc"].astype(np.int32)
    d["logits_scale"] = recs[-1]["s_out"]
    np.savez_compressed(os.path.join(OUT, fname), **d)
    print(fname, "sites", len(names), "bytes", os.path.getsize(os.path.join(OUT, fname)))


def frozen_act(models, bits, scale):
    a = models.QuantAct(bits)
    a.fix()
    n = 2 ** (bits - 1) - 1
    a.min_val = torch.tensor(-float(scale) * n)
    a.max_val = torch.tensor(float(scale) * n)
    return a


def op_fixtures(models):
    rng = np.random.Generator(np.random.PCG64(1234))
    d = {}
    # ---- Shiftmax (IntSoftmax)  16-bit (ViT) and 8-bit (Swin)
    cases = []
    for ci, (rows, n, s, bits, peaky) in enumerate([
            (48, 197, 0.0330, 16, 0), (16, 197, 0.0071, 16, 1), (8, 577, 0.052, 16, 1),
            (32, 49, 0.11, 8, 0), (8, 17, 0.29, 16, 1), (24, 197, 0.0123, 16, 2),
            (16, 64, 0.5, 16, 0), (16, 200, 1.7, 16, 1), (8, 197, 0.0009, 16, 1)]):
        x = rng.integers(-128, 128, size=(rows, n)).astype(np.int8)
        if peaky == 1:
            x = (x.astype(np.int32) // 3).astype(np.int8)
            x[np.arange(rows), rng.integers(0, n, rows)] = 127
        if peaky == 2:
            x[:] = rng.integers(-4, 5, size=(rows, n))
        s = np.float32(s)
        mod = models.IntSoftmax(bits)
        xt = torch.from_numpy(x.astype(np.float32)) * torch.tensor(s)
        with torch.no_grad():
            y, so = mod(xt, torch.tensor(s))
        out = torch.round(y / so).numpy().astype(np.int64)
        assert out.min() >= 0 and out.max() <= 65535
        d[f"shiftmax/{ci}/x"] = x
        d[f"shiftmax/{ci}/s"] = s
        d[f"shiftmax/{ci}/bits"] = bits
        d[f"shiftmax/{ci}/out"] = out.astype(np.uint16)
        cases.append(ci)
    d["shiftmax/n"] = len(cases)
    # ---- ShiftGELU
    ng = 0
    for ci, (rows, C, s, spread) in enumerate([
            (32, 1536, 0.0157, 128), (16, 768, 0.0049, 128), (16, 256, 0.083, 128),
            (8, 3072, 0.031, 40), (8, 1536, 0.0021, 128), (16, 384, 0.3, 128),
            (8, 512, 0.0157, -1)]):
        if spread > 0:
            x = rng.integers(-spread, spread, size=(rows, C)).astype(np.int8)
        else:  # all-negative rows: max < 0 -> exp(-max) branch with positive argument
            x = rng.integers(-128, -3, size=(rows, C)).astype(np.int8)
        s = np.float32(s)
        mod = models.IntGELU()
        with torch.no_grad():
            y, so = mod(torch.from_numpy(x.astype(np.float32)) * torch.tensor(s), torch.tensor(s))
        d[f"gelu/{ci}/x"] = x
        d[f"gelu/{ci}/s"] = s
        d[f"gelu/{ci}/out"] = torch.round(y / so).numpy().astype(np.int16)
        ng += 1
    d["gelu/n"] = ng
    # ---- I-LayerNorm (+ the QuantAct that follows, per-channel, negative weights)
    nl = 0
    for ci, (rows, C, s, amp) in enumerate([
            (64, 384, 8.4e-5, 9000), (32, 192, 6.7e-5, 20000), (16, 768, 1.1e-4, 3000),
            (16, 64, 9.5e-5, 32000), (8, 1024, 5e-5, 12000), (8, 1536, 7e-5, 6000),
            (16, 96, 2.3e-4, 700), (4, 3072, 9e-5, 15000), (8, 384, 8.4e-5, 3)]):
        x = np.clip(np.rint(rng.standard_normal((1, rows, C)) * amp), -32768, 32767).astype(np.int16)
        if ci == 3:
            x[0, 0, :] = 7  # zero-variance row
        s = np.float32(s)
        ln = models.IntLayerNorm(C)
        lw = (1.0 + rng.standard_normal(C) * 0.4).astype(np.float32)
        lw[0] = 0.003
        lw[1] = -0.5
        lb = (rng.standard_normal(C) * 0.5).astype(np.float32)
        ln.weight.data = torch.from_numpy(lw)
        ln.bias.data = torch.from_numpy(lb)
        s_out = np.float32(0.04)
        act = frozen_act(models, 8, s_out)
        with torch.no_grad():
            y, sc = ln(torch.from_numpy(x.astype(np.float32)) * torch.tensor(s), torch.tensor(s))
            z = torch.round(y / sc.reshape(1, 1, -1)).numpy().astype(np.float32)
            y8, so = act(y, sc)
        d[f"ln/{ci}/x"] = x[0]
        d[f"ln/{ci}/s"] = s
        d[f"ln/{ci}/w"] = lw
        d[f"ln/{ci}/b"] = lb
        d[f"ln/{ci}/z"] = z[0]
        d[f"ln/{ci}/s_out"] = np.float32(so.item())
        d[f"ln/{ci}/out8"] = torch.round(y8 / so).numpy().astype(np.int8)[0]
        nl += 1
    d["ln/n"] = nl
    # ---- dyadic requant (QuantAct / fixedpoint_mul): per-channel, scalar, with identity
    nr = 0
    for ci, (rows, C, bits, perch, ident, zmax) in enumerate([
            (64, 384, 8, True, False, 2 ** 18), (64, 384, 16, True, False, 2 ** 20),
            (32, 192, 16, False, True, 2 ** 15), (16, 64, 8, False, False, 2 ** 30),
            (16, 128, 8, True, False, 2 ** 31 - 1), (8, 256, 16, False, True, 2 ** 15)]):
        z = rng.integers(-zmax, zmax, size=(1, rows, C), dtype=np.int64).astype(np.int32)
        if perch:
            s_pre = (10 ** rng.uniform(-9, -5, size=C)).astype(np.float32)
            s_pre[::7] *= -1
        else:
            s_pre = np.array([10 ** rng.uniform(-7, -5)], np.float32)
        s_out = np.float32(np.abs(z).max() * np.abs(s_pre).mean() / (2 ** (bits - 1)) * 1.5)
        act = frozen_act(models, bits, s_out)
        sp = torch.from_numpy(s_pre)
        xt = torch.from_numpy(z.astype(np.float64)).float() * sp.reshape(1, 1, -1)
        # the integer the reference actually sees (fp32 can't hold every int32)
        z_seen = torch.round(xt / sp.reshape(1, 1, -1)).numpy().astype(np.float32)
        kw = {}
        if ident:
            zi = rng.integers(-32768, 32768, size=(1, rows, C)).astype(np.int32)
            s_id = np.array([10 ** rng.uniform(-5, -4)], np.float32)
            kw = dict(identity=torch.from_numpy(zi.astype(np.float32)) * torch.tensor(s_id[0]),
                      identity_scaling_factor=torch.from_numpy(s_id))
            d[f"requant/{ci}/z_id"] = zi[0]
            d[f"requant/{ci}/s_id"] = s_id
        with torch.no_grad():
            y, so = act(xt, sp, **kw)
        d[f"requant/{ci}/z"] = z_seen[0]
        d[f"requant/{ci}/s_pre"] = s_pre
        d[f"requant/{ci}/s_out"] = np.float32(so.item())
        d[f"requant/{ci}/bits"] = bits
        d[f"requant/{ci}/out"] = torch.round(y / so).numpy().astype(np.int32)[0]
        nr += 1
    d["requant/n"] = nr
    # ---- torch CPU sum order probes (A.7)
    ns = 0
    for C in [17, 49, 64, 96, 192, 197, 200, 384, 577, 768, 1024, 1536, 3072, 8191]:
        x = (rng.standard_normal((4, C)) * 1e7).astype(np.float32)
        d[f"sum/{ns}/x"] = x
        d[f"sum/{ns}/out"] = torch.from_numpy(x).sum(dim=-1).numpy()
        ns += 1
    d["sum/n"] = ns
    # ---- input quantisation (a4)
    xf = (rng.standard_normal((3, 3, 32, 32)) * 1.3).astype(np.float32)
    act = frozen_act(models, 8, np.float32(0.0303))
    with torch.no_grad():
        y, so = act(torch.from_numpy(xf))
    d["quant_in/x"] = xf
    d["quant_in/s"] = np.float32(so.item())
    d["quant_in/out"] = torch.round(y / so).numpy().astype(np.int8)
    np.savez_compressed(os.path.join(OUT, "ops.npz"), **d)
    print("ops.npz bytes", os.path.getsize(os.path.join(OUT, "ops.npz")))


def main():
    os.makedirs(OUT, exist_ok=True)
    models = rh.load_reference()
    torch.manual_seed(0)
    if len(sys.argv) > 1 and sys.argv[1] == "--large":
        # the larger members of the model zoo: ViT-L and Swin-S, one image each
        model_fixture(models, "vit_large", 1, 1, 0, False, "vit_large_b1.npz")
        swin_fixture(models, "swin_small", 1, 1, 0, False, "swin_small_b1.npz")
        return
    if len(sys.argv) > 1 and sys.argv[1] == "--swin-base":
        # swin_base_patch4_window7_224 (swin_quant.py:609-627): embed 128, heads 4/8/16/32 — channel counts no other fixture has
        swin_fixture(models, "swin_base", 1, 1, 0, False, "swin_base_b1.npz")
        return
    op_fixtures(models)
    model_fixture(models, "micro_vit", 2, 4, 0, True, "micro_vit_b2.npz")
    model_fixture(models, "micro_vit2h", 3, 4, 0, True, "micro_vit2h_b3.npz")
    model_fixture(models, "deit_tiny", 1, 2, 0, False, "deit_tiny_b1.npz")
    model_fixture(models, "deit_small", 4, 4, 0, False, "deit_small_b4.npz")
    model_fixture(models, "deit_base", 2, 2, 0, False, "deit_base_b2.npz")
    model_fixture(models, "vit_base_384", 1, 1, 0, False, "vit_base_384_b1.npz")
    swin_fixture(models, "micro_swin", 2, 4, 0, True, "micro_swin_b2.npz")
    swin_fixture(models, "swin_tiny", 1, 1, 0, False, "swin_tiny_b1.npz")


if __name__ == "__main__":
    main()
