"""GPU microbench: I-LayerNorm + requant kernel at DeiT-S b256 shape (50432 x 384)."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ivit_amd as iv
from ivit_amd import _lib
P = lambda t: ctypes.c_void_p(t.data_ptr())
H = _lib.Handle(0, torch.cuda.current_stream().cuda_stream)
M, C = 50432, int(os.environ.get("LN_C", "384"))
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.integers(-20000, 20000, (M, C)).astype(np.int16)).cuda()
bi = torch.from_numpy(rng.normal(0, 1e6, C).astype(np.float32)).cuda()
sc = torch.from_numpy((1e-8 * (1 + rng.normal(0, .3, C))).astype(np.float32)).cuda()
d = torch.from_numpy(iv.freeze.dyadic(sc.cpu().numpy(), np.float32(0.05))).cuda()
out = torch.empty(M, C, dtype=torch.int8, device="cuda")
def run(): H.call("ivit_layernorm_requant", P(x), M, C, C, 3e-4, P(bi), P(sc), P(d), P(out))
for _ in range(3): run()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): run()
b.record(); torch.cuda.synchronize()
t = a.elapsed_time(b) / 20 * 1e3
print(f"layernorm_requant {M}x{C}: {t:.1f} us  ({M*C*3/t/1e6:.2f} TB/s algorithmic)  checksum {int(out.sum())}")
