"""Time ivit_mlp_fused_planned against the three planned kernels it replaces (DeiT-S shapes).  usage: python tools/mlp_bench.py [M]"""
import ctypes, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ivit_amd as iv
from ivit_amd import _lib
_P = ctypes.c_void_p
M = int(sys.argv[1]) if len(sys.argv) > 1 else 50432
C, HD = 384, 1536
H = _lib.Handle(0, torch.cuda.current_stream().cuda_stream)
rng = np.random.default_rng(0)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
P = lambda t: _P(t.data_ptr())
dyv = lambda d: _lib.Dyadic(float(d[0, 0]), float(d[0, 1]))
x = dev(rng.integers(-128, 128, (M, C), dtype=np.int8))
w1 = dev(rng.integers(-128, 128, (HD, C), dtype=np.int8)); b1 = dev(rng.integers(-3000, 3000, HD).astype(np.int32))
w2 = dev(rng.integers(-128, 128, (C, HD), dtype=np.int8)); b2 = dev(rng.integers(-3000, 3000, C).astype(np.int32))
d1 = dev(iv.freeze.dyadic((10 ** rng.uniform(-5.6, -5.2, HD)).astype(np.float32), np.float32(0.012)))
d2 = dev(iv.freeze.dyadic((10 ** rng.uniform(-5.9, -5.5, C)).astype(np.float32), np.float32(2e-4)))
dm = iv.freeze.dyadic(np.float32(2e-4), np.float32(3.1e-4)); dr = iv.freeze.dyadic(np.float32(2.7e-4), np.float32(3.1e-4))
res = dev(rng.integers(-30000, 30000, (M, C)).astype(np.int16))
tab = torch.empty(65536, dtype=torch.int8, device="cuda")
H.call("ivit_shiftgelu_build_table", 0.03, dyv(iv.freeze.dyadic(np.float32(0.03 * 2.0 ** -7), np.float32(0.02))), P(tab))
p1, p2, mp = _P(), _P(), _P()
H.call("ivit_linear_plan_create", P(w1), P(b1), P(d1), HD, C, ctypes.byref(p1))
H.call("ivit_linear_plan_create", P(w2), P(b2), P(d2), C, HD, ctypes.byref(p2))
H.call("ivit_mlp_plan_create", p1, p2, ctypes.byref(mp))
h8 = torch.empty(M, HD, dtype=torch.int8, device="cuda"); g8 = torch.empty_like(h8)
ref = torch.empty(M, C, dtype=torch.int16, device="cuda"); out = torch.empty_like(ref)
def chain():
    H.call("ivit_linear_i8_requant_planned", p1, P(x), 8, P(h8), M)
    H.call("ivit_shiftgelu_requant_lut", P(h8), M, HD, P(tab), P(g8))
    H.call("ivit_linear_i8_requant_residual_planned", p2, P(g8), dyv(dm), dyv(dr), P(res), P(ref), M)
def fused():
    H.call("ivit_mlp_fused_planned", mp, P(x), P(tab), dyv(dm), dyv(dr), P(res), P(out), M)
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1000
tc, tf = timeit(chain), timeit(fused)
print(f"M {M}: chain {tc:.1f} us, fused {tf:.1f} us, equal {bool(torch.equal(ref, out))}; fused = {2*2*M*C*HD/tf/1e6:.0f} TOP/s")
