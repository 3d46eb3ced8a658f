"""CPU: the C-ABI library builds/loads and exports every symbol include/ivit.h declares;
host-side freeze logic agrees with the oracle's independent restatement."""
import os
import re

import numpy as np

from conftest import ROOT, load_golden, golden_scales
import ivit_amd as iv
from ivit_amd import _lib


def test_library_exports_header_symbols():
    iv.build()
    lib = _lib.load()
    hdr = open(os.path.join(ROOT, "include", "ivit.h")).read()
    names = set(re.findall(r"\b(ivit_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), n
    for n in _lib.SIGNATURES:
        assert n in names, f"{n} bound in python but not declared in ivit.h"
    assert lib.ivit_version() >= 100
    assert lib.ivit_status_string(1) == b"invalid argument"


def _device_code_object(so_path):
    """The gfx950 ELF inside the library's clang offload bundle (.hip_fatbin)."""
    import struct
    b = open(so_path, "rb").read()
    i = b.find(b"__CLANG_OFFLOAD_BUNDLE__")
    assert i >= 0, "no offload bundle in the library"
    n = struct.unpack_from("<Q", b, i + 24)[0]
    off = i + 32
    for _ in range(n):
        o, sz, t = struct.unpack_from("<QQQ", b, off)
        off += 24
        name = b[off:off + t].decode()
        off += t
        if "gfx950" in name:
            return b[i + o:i + o + sz]
    raise AssertionError("no gfx950 code object")


def test_no_packed_fp32_in_library(tmp_path):
    """Rounds 4-5: v_pk_{add,mul}_f32 with op_sel:[0,1] reads src1's high dword as 0 on lanes 48..63 beside MFMA-issuing waves
    (profiles/r05_hazard/README.md) — the one-LSB LayerNorm differences.  The library is built with -packed-fp32-ops off and its
    ISA must not contain v_pk_{add,mul,fma}_f32 at all."""
    import shutil
    import subprocess
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        objdump = shutil.which("llvm-objdump")
    assert objdump, "llvm-objdump not found"
    so = iv.build()
    co = tmp_path / "dev.co"
    co.write_bytes(_device_code_object(so))
    dis = subprocess.run([objdump, "-d", str(co)], capture_output=True, text=True, check=True).stdout
    assert dis.count("v_mfma_i32") > 1000, "disassembly looks wrong"
    packed = re.findall(r"v_pk_(?:add|mul|fma)_f32", dis)
    assert not packed, f"{len(packed)} packed-fp32 instructions in libivit_hip.so"


def test_no_device_is_an_error_not_a_fallback():
    import ctypes
    import torch
    if torch.cuda.is_available():
        return
    lib = _lib.load()
    h = ctypes.c_void_p()
    assert lib.ivit_create(ctypes.byref(h), 0, None) != 0


def test_freeze_matches_oracle_constants():
    from oracle import oracle as orc
    g = load_golden("micro_vit_b2.npz")
    cfg = iv.CONFIGS[str(g["cfg_name"])]
    w = iv.make_vit_weights(cfg, int(g["seed"]))
    sc = golden_scales(g)
    c, f32 = iv.freeze.freeze_vit(cfg, w, sc)
    o = orc.OracleViT(cfg, w, sc)
    def dy_eq(a, d):
        return all(a[i, 0] == d[i].m and a[i, 1] == d[i].r for i in range(len(d)))
    assert np.array_equal(c["patch_embed.proj.w"], o.c["pe"][0])
    assert np.array_equal(c["patch_embed.proj.b"], o.c["pe"][1])
    assert dy_eq(c["patch_embed.proj.dy"], o.c["pe"][2])
    assert np.array_equal(c["z_cls"], o.c["z_cls"])
    assert np.array_equal(c["pos"].astype(np.int32), o.c["pos"])
    for i, b in enumerate(o.blocks):
        p = f"blocks.{i}."
        assert np.array_equal(c[p + "attn.qkv.w"], b["qkv"][0])
        assert np.array_equal(c[p + "attn.qkv.b"], b["qkv"][1])
        assert dy_eq(c[p + "attn.qkv.dy"], b["qkv"][2])
        assert dy_eq(c[p + "attn.dy_qk"], b["dy_qk"]) and dy_eq(c[p + "attn.dy_pv"], b["dy_av"])
        assert dy_eq(c[p + "norm1.dy"], b["dy_ln1"]) and dy_eq(c[p + "norm2.dy"], b["dy_ln2"])
        assert np.array_equal(c[p + "norm1.bias_int"], b["ln1"][0]) and np.array_equal(c[p + "norm1.sc"], b["ln1"][1])
        assert dy_eq(c[p + "res1.dy_main"], b["dy_res1"][0]) and dy_eq(c[p + "res1.dy_res"], b["dy_res1"][1])
        assert dy_eq(c[p + "mlp.dy_gelu"], b["dy_gelu"])
        assert dy_eq(c[p + "res2.dy_main"], b["dy_res2"][0]) and dy_eq(c[p + "res2.dy_res"], b["dy_res2"][1])
    assert np.array_equal(c["head.w"], o.c["head"][0]) and np.array_equal(c["head.scale"], o.c["head"][2])


def test_dyadic_edge_cases():
    d = iv.freeze.dyadic(np.array([1.0, -0.5, 3e-9, 1e12], np.float32), np.float32(0.37))
    for (m, r), s in zip(d, [1.0, -0.5, 3e-9, 1e12]):
        assert abs(m) >= 2 ** 30 and abs(m) <= 2 ** 31 and m == int(m)
        assert np.isclose(m * r, np.float64(np.float32(s)) / np.float64(np.float32(0.37)), rtol=1e-9)


def test_shiftmax_rowtable_restates_the_two_level_tables():
    """freeze.shiftmax_rowtable (round 6: one 64-entry line of exp_int per row maximum, the host restatement the GPU tests compare
    ivit_shiftmax_rowtable with) against the definition of the two-level tables and against the fp32 arithmetic of
    IntSoftmax.int_exp_shift (quant_modules.py:469-481) for every (row maximum, score) pair, for scales with 1 ... 13 requotient classes;
    None exactly when a line does not fit 64 entries."""
    from ivit_amd import freeze
    for s in (0.3036, 0.2508, 0.2306, 0.1947, 0.52, 0.6203, 0.1059, 0.0902):
        s = np.float32(s)
        tabs = freeze.shiftmax_tables(s)
        assert tabs is not None
        rt = freeze.shiftmax_rowtable(tabs)
        if tabs["R"] > 64:
            assert rt is None
            continue
        assert rt.shape == (256, 64) and rt.dtype == np.float32
        v = np.arange(-128, 128).astype(np.float32)
        f = ((v * s).astype(np.float32) / s).astype(np.float32)                    # x~(v)
        dmin, R = int(tabs["dmin"]), int(tabs["R"])
        for q in range(256):
            direct = freeze._shift_exp_f32((f[: q + 1] - f[q]).astype(np.float32), s)     # exp_int of every score <= the row maximum
            dd = np.maximum(np.arange(q + 1) - q, dmin) - dmin
            assert np.array_equal(rt[q, dd], direct), (float(s), q)
        assert not rt[:, R:].any()
