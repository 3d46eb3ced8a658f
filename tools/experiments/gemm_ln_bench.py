"""GPU microbench + parity: residual QuantLinear + I-LayerNorm, fused (ivit_linear_i8_residual_layernorm) vs the
two-kernel chain, at DeiT-S shapes (whole batch and one of four slices)."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ivit_amd as iv
from ivit_amd import _lib
P = lambda t: ctypes.c_void_p(t.data_ptr())
H = _lib.Handle(0, torch.cuda.current_stream().cuda_stream)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
rng = np.random.default_rng(0)
N = 384
for M in (50432, 12608):
    for name, K in (("proj", 384), ("fc2", 1536)):
        x = torch.from_numpy(rng.integers(-128, 128, (M, K), dtype=np.int8)).cuda()
        w = torch.from_numpy(rng.integers(-128, 128, (N, K), dtype=np.int8)).cuda()
        b = torch.from_numpy(rng.integers(-1000, 1000, N).astype(np.int32)).cuda()
        s_acc = (10 ** rng.uniform(-5.2, -5, N)).astype(np.float32) * np.float32(np.sqrt(384.0 / K))
        d16 = torch.from_numpy(iv.freeze.dyadic(s_acc, np.float32(2e-4))).cuda()
        r16 = torch.randint(-20000, 20000, (M, N), dtype=torch.int16, device="cuda")
        dmn = iv.freeze.dyadic(np.float32(2e-4), np.float32(3.1e-4)); drn = iv.freeze.dyadic(np.float32(2.7e-4), np.float32(3.1e-4))
        dm = _lib.Dyadic(float(dmn[0, 0]), float(dmn[0, 1])); dr = _lib.Dyadic(float(drn[0, 0]), float(drn[0, 1]))
        s_ln = np.float32(3.1e-4)
        lw = rng.uniform(0.5, 1.5, N).astype(np.float32); lb = rng.uniform(-0.5, 0.5, N).astype(np.float32)
        bias_int, sc = iv.freeze.layernorm_constants(lw, lb) if hasattr(iv.freeze, "layernorm_constants") else (None, None)
        if bias_int is None:
            sf = np.float32(np.sqrt(np.float32(N))) / np.float32(2 ** 30)   # quant_modules.py:352-367
            bias_int = np.floor((lb / lw) / sf).astype(np.float32)
            sc = (sf * lw).astype(np.float32)
        ln_dy = torch.from_numpy(iv.freeze.dyadic(sc, np.float32(0.03))).cuda()
        bi_d, sc_d = torch.from_numpy(bias_int).cuda(), torch.from_numpy(sc).cuda()
        y_ref = torch.empty(M, N, dtype=torch.int16, device="cuda"); a_ref = torch.empty(M, N, dtype=torch.int8, device="cuda")
        y = torch.full((M, N), 7, dtype=torch.int16, device="cuda"); a8 = torch.full((M, N), 7, dtype=torch.int8, device="cuda")
        def unfused():
            H.call("ivit_linear_i8_requant_residual", P(x), P(w), P(b), P(d16), dm, dr, P(r16), P(y_ref), M, N, K)
            H.call("ivit_layernorm_requant", P(y_ref), M, N, N, float(s_ln), P(bi_d), P(sc_d), P(ln_dy), P(a_ref))
        def fused():
            H.call("ivit_linear_i8_residual_layernorm", P(x), P(w), P(b), P(d16), dm, dr, P(r16), P(y), float(s_ln),
                   P(bi_d), P(sc_d), P(ln_dy), P(a8), M, N, K)
        unfused(); fused(); torch.cuda.synchronize()
        ok = bool((y == y_ref).all()) and bool((a8 == a_ref).all())
        sat = float((a_ref.abs() >= 127).float().mean()), float((y_ref.abs() >= 32767).float().mean())
        tg = timeit(lambda: H.call("ivit_linear_i8_requant_residual", P(x), P(w), P(b), P(d16), dm, dr, P(r16), P(y_ref), M, N, K))
        tu = timeit(unfused); tf = timeit(fused)
        print(f"M={M} {name} K={K}: gemm {tg:.1f} us, gemm+LN {tu:.1f} us, fused {tf:.1f} us, equal={ok}, sat(a8,y16)={sat}", flush=True)
