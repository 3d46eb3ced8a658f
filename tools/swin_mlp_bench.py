"""Time ivit_mlp_fused (C = 96, hidden 384: Swin stage 0) at Swin-T b256 / b128 row counts.  usage: python tools/swin_mlp_bench.py"""
import ctypes, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ivit_amd as iv
from ivit_amd import _lib
_P = ctypes.c_void_p
P = lambda t: _P(t.data_ptr())
dyv = lambda d: _lib.Dyadic(float(d[0, 0]), float(d[0, 1]))
H = _lib.Handle(0, torch.cuda.current_stream().cuda_stream)
rng = np.random.default_rng(0)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
C, HD = 96, 384
w1 = dev(rng.integers(-128, 128, (HD, C), dtype=np.int8)); b1 = dev(rng.integers(-3000, 3000, HD).astype(np.int32))
w2 = dev(rng.integers(-128, 128, (C, HD), dtype=np.int8)); b2 = dev(rng.integers(-3000, 3000, C).astype(np.int32))
d1 = dev(iv.freeze.dyadic((10 ** rng.uniform(-5.0, -4.6, HD)).astype(np.float32), np.float32(0.012)))
d2 = dev(iv.freeze.dyadic((10 ** rng.uniform(-5.3, -4.9, C)).astype(np.float32), np.float32(2e-4)))
dm = iv.freeze.dyadic(np.float32(2e-4), np.float32(3.1e-4)); dr = iv.freeze.dyadic(np.float32(2.7e-4), np.float32(3.1e-4))
tab = torch.empty(65536, dtype=torch.int8, device="cuda")
H.call("ivit_shiftgelu_build_table", 0.03, dyv(iv.freeze.dyadic(np.float32(0.03 * 2.0 ** -7), np.float32(0.02))), P(tab))
for M in (802816, 401408):
    x = dev(rng.integers(-128, 128, (M, C), dtype=np.int8)); res = dev(rng.integers(-30000, 30000, (M, C)).astype(np.int16))
    out = torch.empty(M, C, dtype=torch.int16, device="cuda")
    f = lambda: H.call("ivit_mlp_fused", P(x), P(w1), P(b1), P(d1), P(tab), P(w2), P(b2), P(d2), dyv(dm), dyv(dr), P(res), P(out), M, C, HD)
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): f()
    b.record(); torch.cuda.synchronize()
    t = a.elapsed_time(b) / 10 * 1000
    print(f"M {M}: ivit_mlp_fused {t:.1f} us  ({4.0 * M * C * HD / t / 1e6:.0f} TOP/s, {M * C * 5 / t / 1e3:.0f} GB/s algorithmic)")
