"""GPU stress at operator level: the same C-ABI call issued on NS streams at once (separate inputs of equal content, separate
outputs), repeated; every output is compared with the single-stream result.  Finds kernels whose result depends on what
shares the chip with them.   python tools/op_stress.py [repeats] [streams]"""
import ctypes, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import conftest  # noqa: F401
import ivit_amd as iv
from ivit_amd import _lib
_P = ctypes.c_void_p
P = lambda t: _P(t.data_ptr())
REP = int(sys.argv[1]) if len(sys.argv) > 1 else 30
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 8
streams = [torch.cuda.Stream() for _ in range(NS)]
hs = [_lib.Handle(0, s.cuda_stream) for s in streams]
rng = np.random.default_rng(3)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
dyv = lambda d: _lib.Dyadic(float(d[0, 0]), float(d[0, 1]))

OPS = []
def stress(name, make_out, call, shared):
    OPS.append((name, make_out, call))
    if os.environ.get("MIXED_ONLY"): return
    _stress(name, make_out, call)

def _stress(name, make_out, call):
    """shared: dict of device tensors (inputs, read-only); make_out() -> fresh output tensor; call(h, out)"""
    ref = make_out(); call(hs[0], ref); torch.cuda.synchronize(); ref = ref.clone()
    bad = 0
    for _ in range(REP):
        outs = [make_out() for _ in range(NS)]
        torch.cuda.synchronize()
        for i in range(NS):
            call(hs[i], outs[i])
        torch.cuda.synchronize()
        bad += sum(0 if torch.equal(o, ref) else 1 for o in outs)
    print(f"{name:46s}: {bad} of {REP * NS} concurrent launches differ from the single-stream result")

B, R = 32, 56                                   # a Swin-T stage-0 slice of 32 images
M, C = B * R * R, 96
# ---- unplanned QuantLinear (gemm_glds): qkv of stage 0 (K = 96), fc-like K = 384
for (K, N, tag) in ((96, 288, "K=96 N=288"), (384, 1152, "K=384 N=1152"), (192, 576, "K=192 N=576"), (768, 2304, "K=768 N=2304")):
    Mk = M if K == 96 else M // (K // 96) ** 2 * 1
    x = dev(rng.integers(-128, 128, (Mk, K), dtype=np.int8)); w = dev(rng.integers(-128, 128, (N, K), dtype=np.int8))
    b = dev(rng.integers(-3000, 3000, N).astype(np.int32))
    d = dev(iv.freeze.dyadic((10 ** rng.uniform(-5.6, -5.2, N)).astype(np.float32), np.float32(0.012)))
    stress(f"ivit_linear_i8_requant {tag} M={Mk}", lambda: torch.empty(Mk, N, dtype=torch.int8, device="cuda"),
           lambda h, o: h.call("ivit_linear_i8_requant", P(x), P(w), P(b), P(d), 8, P(o), Mk, N, K), None)
    res = dev(rng.integers(-30000, 30000, (Mk, K)).astype(np.int16))
    w2 = dev(rng.integers(-128, 128, (K, K), dtype=np.int8)); b2 = dev(rng.integers(-3000, 3000, K).astype(np.int32))
    d2 = dev(iv.freeze.dyadic((10 ** rng.uniform(-5.9, -5.5, K)).astype(np.float32), np.float32(2e-4)))
    dm = iv.freeze.dyadic(np.float32(2e-4), np.float32(3.1e-4)); dr = iv.freeze.dyadic(np.float32(2.7e-4), np.float32(3.1e-4))
    stress(f"ivit_linear_i8_requant_residual K=N={K} M={Mk}", lambda: torch.empty(Mk, K, dtype=torch.int16, device="cuda"),
           lambda h, o: h.call("ivit_linear_i8_requant_residual", P(x), P(w2), P(b2), P(d2), dyv(dm), dyv(dr), P(res), P(o), Mk, K, K), None)
# ---- window attention, stage 0 (3 heads) shifted
heads = 3
qkv = dev(rng.integers(-128, 128, (B, R, R, 3 * C), dtype=np.int8))
relb = dev(rng.integers(-200, 200, (heads, 49, 49)).astype(np.int16))
dq = iv.freeze.dyadic(np.float32(3e-4), np.float32(0.05)); da = iv.freeze.dyadic(np.float32(0.05), np.float32(0.06)); dp = iv.freeze.dyadic(np.float32(2e-4), np.float32(0.03))
for shift in (0, 3):
    stress(f"ivit_window_attention_fused shift={shift}", lambda: torch.empty(B, R * R, C, dtype=torch.int8, device="cuda"),
           lambda h, o: h.call("ivit_window_attention_fused", P(qkv), dyv(dq), dyv(da), P(relb), 0.06, dyv(dp), P(o), B, R, 7, shift, heads, 32), None)
# ---- LayerNorms
x16 = dev(rng.integers(-20000, 20000, (M, C)).astype(np.int16))
bi = dev(rng.normal(0, 3e5, C).astype(np.float32)); sc = dev((10 ** rng.uniform(-10.2, -9.8, C)).astype(np.float32))
dln = dev(iv.freeze.dyadic((10 ** rng.uniform(-10.2, -9.8, C)).astype(np.float32), np.float32(0.03)))
stress("ivit_layernorm_tokenorder_requant C=96", lambda: torch.empty(M, C, dtype=torch.int8, device="cuda"),
       lambda h, o: h.call("ivit_layernorm_tokenorder_requant", P(x16), M, C, 0.01, P(bi), P(sc), P(dln), R * R, P(o)), None)
stress("ivit_layernorm_requant C=96", lambda: torch.empty(M, C, dtype=torch.int8, device="cuda"),
       lambda h, o: h.call("ivit_layernorm_requant", P(x16), M, C, C, 0.01, P(bi), P(sc), P(dln), P(o)), None)
stress("ivit_patch_merge_gather", lambda: torch.empty(B * (R // 2) ** 2, 4 * C, dtype=torch.int16, device="cuda"),
       lambda h, o: h.call("ivit_patch_merge_gather", P(x16), 16, B, R, C, P(o)), None)

# ---- the rest of the Swin path
img = dev(rng.integers(-128, 128, (B, 3, 224, 224), dtype=np.int8))
stress("ivit_im2col_patch P=4", lambda: torch.empty(B * 56 * 56, 48, dtype=torch.int8, device="cuda"),
       lambda h, o: h.call("ivit_im2col_patch", P(img), B, 3, 224, 224, 4, P(o)), None)
xp = dev(rng.integers(-128, 128, (M, 48), dtype=np.int8)); wp = dev(rng.integers(-128, 128, (96, 48), dtype=np.int8))
bp = dev(rng.integers(-3000, 3000, 96).astype(np.int32)); dpe = dev(iv.freeze.dyadic((10 ** rng.uniform(-4.6, -4.2, 96)).astype(np.float32), np.float32(0.012)))
stress("ivit_linear_i8_requant K=48 N=96 (patch embed)", lambda: torch.empty(M, 96, dtype=torch.int8, device="cuda"),
       lambda h, o: h.call("ivit_linear_i8_requant", P(xp), P(wp), P(bp), P(dpe), 8, P(o), M, 96, 48), None)
for (Rr, hd) in ((28, 6), (14, 12), (7, 24)):
    Cc = hd * 32
    qk = dev(rng.integers(-128, 128, (B, Rr, Rr, 3 * Cc), dtype=np.int8)); rb = dev(rng.integers(-200, 200, (hd, 49, 49)).astype(np.int16))
    for shift in ((0, 3) if Rr > 7 else (0,)):
        stress(f"ivit_window_attention_fused R={Rr} heads={hd} shift={shift}", (lambda Rr=Rr, Cc=Cc: torch.empty(B, Rr * Rr, Cc, dtype=torch.int8, device="cuda")),
               (lambda h, o, qk=qk, rb=rb, Rr=Rr, hd=hd, shift=shift: h.call("ivit_window_attention_fused", P(qk), dyv(dq), dyv(da), P(rb), 0.06, dyv(dp), P(o), B, Rr, 7, shift, hd, 32)), None)
# LN_BIG=1: the S = 4 register LayerNorm forms (packed-fp32 ISA in the default build) at DeiT-B / ViT-L sized inputs
_ln_shapes = ((192, B * 28 * 28), (384, B * 14 * 14), (768, B * 7 * 7), (1536, B * 7 * 7))
if os.environ.get("LN_BIG"):
    _ln_shapes += ((768, 64 * 197), (1024, 32 * 197))
if os.environ.get("LN_ODD"):            # a channel count outside the register kernel's list: layernorm16_kernel, the round-2 LayerNorm
    _ln_shapes += ((200, B * 28 * 28),)
for (Cc, Mm) in _ln_shapes:
    xx = dev(rng.integers(-20000, 20000, (Mm, Cc)).astype(np.int16))
    bb = dev(rng.normal(0, 3e5, Cc).astype(np.float32)); ss = dev((10 ** rng.uniform(-10.2, -9.8, Cc)).astype(np.float32))
    dd = dev(iv.freeze.dyadic((10 ** rng.uniform(-10.2, -9.8, Cc)).astype(np.float32), np.float32(0.03)))
    stress(f"ivit_layernorm_requant C={Cc} M={Mm}", (lambda Mm=Mm, Cc=Cc: torch.empty(Mm, Cc, dtype=torch.int8, device="cuda")),
           (lambda h, o, xx=xx, bb=bb, ss=ss, dd=dd, Mm=Mm, Cc=Cc: h.call("ivit_layernorm_requant", P(xx), Mm, Cc, Cc, 0.01, P(bb), P(ss), P(dd), P(o))), None)
a8w = dev(rng.integers(-128, 128, (B * 28 * 28 * 192,), dtype=np.int8))
stress("ivit_widen_i8_i16", lambda: torch.empty(B * 28 * 28 * 192, dtype=torch.int16, device="cuda"),
       lambda h, o: h.call("ivit_widen_i8_i16", P(a8w), P(o), B * 28 * 28 * 192), None)
ap = dev(rng.integers(-128, 128, (B, 49, 768), dtype=np.int8)); dpool = iv.freeze.dyadic(np.float32(0.02), np.float32(0.03))
stress("ivit_avgpool_requant", lambda: torch.empty(B, 768, dtype=torch.int8, device="cuda"),
       lambda h, o: h.call("ivit_avgpool_requant", P(ap), B, 49, 768, dyv(dpool), P(o)), None)
pw = dev(rng.integers(-128, 128, (B, 768), dtype=np.int8)); hw = dev(rng.integers(-128, 128, (1000, 768), dtype=np.int8)); hb = dev(rng.integers(-3000, 3000, 1000).astype(np.int32))
stress("ivit_linear_i8 (head)", lambda: torch.empty(B, 1000, dtype=torch.int32, device="cuda"),
       lambda h, o: h.call("ivit_linear_i8", P(pw), P(hw), P(hb), P(o), B, 1000, 768), None)

# ---- the ViT path's kernels (DeiT-S shapes, 32 images): planned GEMMs, fused attention, fused Mlp, front end
if not os.environ.get("SWIN_ONLY"):
    Bv, T, D, Hh, Hd = 32, 197, 384, 6, 1536
    Mv, ldv = Bv * T, 208
    xa = dev(rng.integers(-128, 128, (Mv, D), dtype=np.int8))
    def lin(N, K, lo=-5.6, hi=-5.2, so=0.012):
        w = dev(rng.integers(-128, 128, (N, K), dtype=np.int8)); b = dev(rng.integers(-3000, 3000, N).astype(np.int32))
        d = dev(iv.freeze.dyadic((10 ** rng.uniform(lo, hi, N)).astype(np.float32), np.float32(so)))
        pl = _P(); hs[0].call("ivit_linear_plan_create", P(w), P(b), P(d), N, K, ctypes.byref(pl))
        return (w, b, d), pl
    keep_q, pq = lin(3 * D, D)
    stress("ivit_linear_i8_qkv_planned", lambda: torch.zeros(3 * Bv * Hh * 64 * ldv, dtype=torch.int8, device="cuda"),
           lambda h, o: h.call("ivit_linear_i8_qkv_planned", pq, P(xa), P(o), _P(o.data_ptr() + Bv * Hh * T * 64), _P(o.data_ptr() + 2 * Bv * Hh * T * 64), Bv, T, Hh, 64, ldv), None)
    keep_p, pp = lin(D, D, -5.9, -5.5, 2e-4)
    resv = dev(rng.integers(-30000, 30000, (Mv, D)).astype(np.int16))
    dmv = iv.freeze.dyadic(np.float32(2e-4), np.float32(3.1e-4)); drv = iv.freeze.dyadic(np.float32(2.7e-4), np.float32(3.1e-4))
    stress("ivit_linear_i8_requant_residual_planned", lambda: torch.empty(Mv, D, dtype=torch.int16, device="cuda"),
           lambda h, o: h.call("ivit_linear_i8_requant_residual_planned", pp, P(xa), dyv(dmv), dyv(drv), P(resv), P(o), Mv), None)
    keep_1, p1 = lin(Hd, D)
    keep_2, p2 = lin(D, Hd, -5.9, -5.5, 2e-4)
    stress("ivit_linear_i8_requant_planned (fc1)", lambda: torch.empty(Mv, Hd, dtype=torch.int8, device="cuda"),
           lambda h, o: h.call("ivit_linear_i8_requant_planned", p1, P(xa), 8, P(o), Mv), None)
    tabv = torch.empty(65536, dtype=torch.int8, device="cuda")
    hs[0].call("ivit_shiftgelu_build_table", 0.03, dyv(iv.freeze.dyadic(np.float32(0.03 * 2.0 ** -7), np.float32(0.02))), P(tabv))
    mpv = _P(); hs[0].call("ivit_mlp_plan_create", p1, p2, ctypes.byref(mpv))
    stress("ivit_mlp_fused_planned", lambda: torch.empty(Mv, D, dtype=torch.int16, device="cuda"),
           lambda h, o: h.call("ivit_mlp_fused_planned", mpv, P(xa), P(tabv), dyv(dmv), dyv(drv), P(resv), P(o), Mv), None)
    # round 5: row counts that reach the role-split kernels (two 80-token units per CU; two 64-token tiles per workgroup)
    Mr = 128 * T
    xr = dev(rng.integers(-128, 128, (Mr, D), dtype=np.int8)); resr = dev(rng.integers(-30000, 30000, (Mr, D)).astype(np.int16))
    stress("ivit_mlp_fused_planned M=25216 (role-split)", lambda: torch.empty(Mr, D, dtype=torch.int16, device="cuda"),
           lambda h, o: h.call("ivit_mlp_fused_planned", mpv, P(xr), P(tabv), dyv(dmv), dyv(drv), P(resr), P(o), Mr), None)
    h8v = dev(rng.integers(-128, 128, (Mv, Hd), dtype=np.int8))
    stress("ivit_shiftgelu_requant_lut", lambda: torch.empty(Mv, Hd, dtype=torch.int8, device="cuda"),
           lambda h, o: h.call("ivit_shiftgelu_requant_lut", P(h8v), Mv, Hd, P(tabv), P(o)), None)
    qv = dev(rng.integers(-128, 128, (Bv * Hh, T, 64), dtype=np.int8)); kv = dev(rng.integers(-128, 128, (Bv * Hh, T, 64), dtype=np.int8))
    vtn = np.zeros((Bv * Hh, 64, ldv), np.int8); vtn[:, :, :T] = rng.integers(-128, 128, (Bv * Hh, 64, T), dtype=np.int8)
    vtv = dev(vtn)
    dqkv, dpvv = iv.freeze.dyadic(np.float32(2e-4), np.float32(6e-2)), iv.freeze.dyadic(np.float32(3e-6), np.float32(9e-3))
    stress("ivit_attention_fused T=197", lambda: torch.empty(Bv, T, D, dtype=torch.int8, device="cuda"),
           lambda h, o: h.call("ivit_attention_fused", P(qv), P(kv), P(vtv), dyv(dqkv), 0.06, dyv(dpvv), P(o), Bv, Hh, T, 64, ldv), None)
    etab = iv.freeze.shiftmax_tables(np.float32(0.06))
    eaq, eet, ecl = dev(etab["aq"]), dev(etab["t"]), dev(etab["cls"])
    stress("ivit_attention_fused_lut T=197", lambda: torch.empty(Bv, T, D, dtype=torch.int8, device="cuda"),
           lambda h, o: h.call("ivit_attention_fused_lut", P(qv), P(kv), P(vtv), dyv(dqkv), 0.06, P(eaq), P(eet), P(ecl), int(etab["NC"]),
                               int(etab["t"].size), int(etab["dmin"]), dyv(dpvv), P(o), Bv, Hh, T, 64, ldv), None)
    imgv = dev(rng.integers(-128, 128, (Bv, 3, 224, 224), dtype=np.int8))
    stress("ivit_im2col_patch P=16", lambda: torch.empty(Bv * 196, 768, dtype=torch.int8, device="cuda"),
           lambda h, o: h.call("ivit_im2col_patch", P(imgv), Bv, 3, 224, 224, 16, P(o)), None)
    p16v = dev(rng.integers(-20000, 20000, (Bv, T - 1, D)).astype(np.int16)); zc = dev(rng.integers(-10 ** 6, 10 ** 6, D).astype(np.int32))
    posv = dev(rng.integers(-20000, 20000, (T, D)).astype(np.int16))
    stress("ivit_embed_finish", lambda: torch.empty(Bv, T, D, dtype=torch.int16, device="cuda"),
           lambda h, o: h.call("ivit_embed_finish", P(p16v), P(zc), P(posv), dyv(dmv), dyv(drv), P(o), Bv, T, D), None)
    x96 = dev(rng.integers(-128, 128, (M // 4, 96), dtype=np.int8))
    w1s = dev(rng.integers(-128, 128, (384, 96), dtype=np.int8)); b1s = dev(rng.integers(-3000, 3000, 384).astype(np.int32))
    w2s = dev(rng.integers(-128, 128, (96, 384), dtype=np.int8)); b2s = dev(rng.integers(-3000, 3000, 96).astype(np.int32))
    d1s = dev(iv.freeze.dyadic((10 ** rng.uniform(-5.0, -4.6, 384)).astype(np.float32), np.float32(0.012)))
    d2s = dev(iv.freeze.dyadic((10 ** rng.uniform(-5.3, -4.9, 96)).astype(np.float32), np.float32(2e-4)))
    res96 = dev(rng.integers(-30000, 30000, (M // 4, 96)).astype(np.int16))
    x96b = dev(rng.integers(-128, 128, (M, 96), dtype=np.int8)); res96b = dev(rng.integers(-30000, 30000, (M, 96)).astype(np.int16))
    stress("ivit_mlp_fused C=96 M=100352 (role-split)", lambda: torch.empty(M, 96, dtype=torch.int16, device="cuda"),
           lambda h, o: h.call("ivit_mlp_fused", P(x96b), P(w1s), P(b1s), P(d1s), P(tabv), P(w2s), P(b2s), P(d2s), dyv(dmv), dyv(drv), P(res96b), P(o), M, 96, 384), None)
    stress("ivit_mlp_fused C=96", lambda: torch.empty(M // 4, 96, dtype=torch.int16, device="cuda"),
           lambda h, o: h.call("ivit_mlp_fused", P(x96), P(w1s), P(b1s), P(d1s), P(tabv), P(w2s), P(b2s), P(d2s), dyv(dmv), dyv(drv), P(res96), P(o), M // 4, 96, 384), None)

# ---- mixed: every stream walks the operator list (rotated by its index), all streams at once.  MIX_FILTERS="a;b;c" runs
# one mixed test per filter with only the operators whose name contains it plus the VICTIM operator (name contains MIX_VICTIM)
def mixed(ops, tag):
    refs = []
    for name, mk, call in ops:
        r = mk(); call(hs[0], r); torch.cuda.synchronize(); refs.append(r.clone())
    bad = {name: 0 for name, _, _ in ops}
    for _ in range(REP):
        outs = [[mk() for _, mk, _ in ops] for _ in range(NS)]
        torch.cuda.synchronize()
        for j in range(len(ops)):
            for i in range(NS):
                k = (j + i * 3) % len(ops)
                ops[k][2](hs[i], outs[i][k])
        torch.cuda.synchronize()
        for i in range(NS):
            for k, (name, _, _) in enumerate(ops):
                if not torch.equal(outs[i][k], refs[k]):
                    bad[name] += 1
                    if bad[name] <= 3 and outs[i][k].dim() == 2:
                        dd = (outs[i][k] != refs[k]).cpu().numpy(); rows = np.nonzero(dd.any(1))[0]
                        c0 = np.nonzero(dd[rows[0]])[0]
                        print(f"    {name}: {int(dd.sum())} elements in {len(rows)} rows {rows[:10]}; first row cols {c0[:12]} (n={len(c0)}); got",
                              outs[i][k][rows[0]].cpu().numpy()[c0[:6]], "ref", refs[k][rows[0]].cpu().numpy()[c0[:6]])
    print(f"mixed [{tag}] ({len(ops)} operators side by side): " + ("all equal" if not any(bad.values()) else ""))
    for name, n in bad.items():
        if n: print(f"  {name:46s}: {n} of {REP * NS} differ")

filters = os.environ.get("MIX_FILTERS")
if filters:
    victim = os.environ.get("MIX_VICTIM", "layernorm_requant C=192")
    for flt in filters.split(";"):
        mixed([o for o in OPS if victim in o[0] or flt in o[0]], flt)
else:
    mixed(OPS, "all")
