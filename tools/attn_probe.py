"""Fused attention alone: the library's kernel against the probe build of the same headers (tools/ubench/attn_probe.hip),
DeiT-S b256 (T = 197) and ViT-B@384 b128 (T = 577) shapes, real Shiftmax tables, outputs compared byte for byte; prints the
timings.  ATTN_T=197|577 picks one shape."""
import ctypes, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import ivit_amd as iv
from ivit_amd import _lib
P = lambda t: ctypes.c_void_p(t.data_ptr())
H = _lib.Handle(0, torch.cuda.current_stream().cuda_stream)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
dy = lambda d: _lib.Dyadic(float(d[0, 0]), float(d[0, 1]))
probe = ctypes.CDLL(os.path.join(ROOT, "tools", "ubench", os.environ.get("ATTN_PROBE_SO", "libattn_probe.so")))
D, F, I, V = ctypes.c_double, ctypes.c_float, ctypes.c_int, ctypes.c_void_p
probe.attn_probe.argtypes = [V, V, V, D, D, F, V, V, V, I, I, I, V, D, D, V, I, I, I, I, I, I, ctypes.POINTER(F)]
CASES = {197: (256, 6, ((0, 0.1947, 30), (1, 0.3036, 30), (2, 0.2508, 30), (3, 0.2306, 12))),
         577: (128, 12, ((0, 0.6203, 30), (1, 0.4640, 30), (2, 0.5776, 12)))}
for T in ([int(os.environ["ATTN_T"])] if os.environ.get("ATTN_T") else (197, 577)):
    B, Hh, scales = CASES[T]
    dh, ld = 64, (T + 15) // 16 * 16
    for seed, s, spread in scales:
        rng = np.random.default_rng(seed)
        q = dev(rng.normal(0, spread, (B * Hh, T, dh)).clip(-127, 127).astype(np.int8)); k = dev(rng.normal(0, spread, (B * Hh, T, dh)).clip(-127, 127).astype(np.int8))
        vt = np.zeros((B * Hh, dh, ld), np.int8); vt[:, :, :T] = rng.integers(-128, 128, (B * Hh, dh, T), dtype=np.int8); vt = dev(vt)
        s = np.float32(s)
        tabs = iv.freeze.shiftmax_tables(s)
        aq, et, cls = dev(tabs["aq"]), dev(tabs["t"]), dev(tabs["cls"])
        dqk = iv.freeze.dyadic(np.float32(6e-4 * 30 / spread * (0.1947 / float(s))), s); dpv = iv.freeze.dyadic(np.float32(3e-6), np.float32(9e-3))
        ref = torch.empty(B, T, Hh * dh, dtype=torch.int8, device="cuda"); out = torch.zeros_like(ref)
        f = lambda: H.call("ivit_attention_fused_lut", P(q), P(k), P(vt), dy(dqk), float(s), P(aq), P(et), P(cls), int(tabs["NC"]), int(tabs["t"].size), int(tabs["dmin"]), dy(dpv), P(ref), B, Hh, T, dh, ld)
        for _ in range(3): f()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for rep in range(5):
            a.record()
            for _ in range(10): f()
            b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / 10 * 1000)
        rt = iv.freeze.shiftmax_rowtable(tabs)
        rowtab = dev(rt) if rt is not None else None
        vrow = vt[:, :, :T].transpose(1, 2).contiguous()          # [B*H, T, dh]
        variants = [("two-gather", 0, None, vt, ld), ("row lines", 1, rowtab, vt, ld), ("row lines, v row-major", 1, rowtab, vrow, 0)]
        if T == 577:
            variants += [("stream (packed bytes)", 2, rowtab, vt, ld), ("stream, v row-major", 2, rowtab, vrow, 0)]
        for name, var, rtp, vsrc, ldv in variants:
            if var and rowtab is None:
                continue
            out.zero_()
            us = F(0)
            rc = probe.attn_probe(P(q), P(k), P(vsrc), float(dqk[0, 0]), float(dqk[0, 1]), float(s), P(aq), P(et), P(cls), int(tabs["NC"]), int(tabs["t"].size),
                                  int(tabs["dmin"]), P(rtp) if rtp is not None else None, float(dpv[0, 0]), float(dpv[0, 1]), P(out), B, Hh, T, ldv, var, 10, ctypes.byref(us))
            torch.cuda.synchronize()
            print(f"T {T} s {float(s):.4f} NC {tabs['NC']} R {tabs['R']} spread {spread}: library (two-level tables) {min(ts):.1f} us | probe {name} {us.value:.1f} us rc {rc} | "
                  f"{int((out != ref).sum())} bytes differ, {int((ref != 0).sum()) * 100 // ref.numel()} % non-zero", flush=True)
