"""GPU-side debugging aid: fused Mlp kernel vs the unfused planned chain, mismatch pattern by token / channel.
   python tools/mlp_debug.py [M]"""
import ctypes, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import conftest  # noqa: F401  (puts the package alias on the path)
import ivit_amd as iv
from ivit_amd import _lib
_P = ctypes.c_void_p
P = lambda t: _P(t.data_ptr())
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
dyv = lambda d: _lib.Dyadic(float(d[0, 0]), float(d[0, 1]))
M = int(sys.argv[1]) if len(sys.argv) > 1 else 64
H = _lib.Handle(0, torch.cuda.current_stream().cuda_stream)
rng = np.random.default_rng(M + 5)
C, HD = 384, 1536
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
ref = torch.empty(M, C, dtype=torch.int16, device="cuda")
H.call("ivit_linear_i8_requant_planned", p1, P(x), 8, P(h8), M)
H.call("ivit_shiftgelu_requant_lut", P(h8), M, HD, P(tab), P(g8))
H.call("ivit_linear_i8_requant_residual_planned", p2, P(g8), dyv(dm), dyv(dr), P(res), P(ref), M)
out = torch.full((M, C), -7, dtype=torch.int16, device="cuda")
H.call("ivit_mlp_fused_planned", mp, P(x), P(tab), dyv(dm), dyv(dr), P(res), P(out), M)
torch.cuda.synchronize()
o, r = out.cpu().numpy().astype(np.int64), ref.cpu().numpy().astype(np.int64)
bad = o != r
print("M", M, "mismatches", int(bad.sum()), "of", bad.size, " untouched (-7):", int((o == -7).sum()))
print("bad tokens (first 70):", np.nonzero(bad.any(1))[0][:70])
print("bad channels (first 70):", np.nonzero(bad.any(0))[0][:70])
print("max |diff|", int(np.abs(o - r).max()), " mean |diff| over bad", float(np.abs(o - r)[bad].mean()) if bad.any() else 0)
print("token 0, first 16 channels: got", o[0, :16], "ref", r[0, :16])
