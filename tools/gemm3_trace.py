"""GPU: dump the k-step timeline of gemm_as_kernel (library built with -DG3_TRACE=1, IVIT_LIB pointing at it)."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ivit_amd as iv
from ivit_amd import _lib
P = lambda t: ctypes.c_void_p(t.data_ptr())
H = _lib.Handle(0, torch.cuda.current_stream().cuda_stream)
rng = np.random.default_rng(0)
M, N, K = 50432, int(os.environ.get("TR_N", 1536)), 384
x = torch.from_numpy(rng.integers(-128, 128, (M, K), dtype=np.int8)).cuda()
w = torch.from_numpy(np.rint(rng.normal(0, 40, (N, K)).clip(-127, 127)).astype(np.int8)).cuda()
b = torch.from_numpy(rng.integers(-20000, 20000, N).astype(np.int32)).cuda()
d8 = torch.from_numpy(iv.freeze.dyadic((10 ** rng.uniform(-5.2, -5, N)).astype(np.float32), np.float32(0.04))).cuda()
p8 = H.linear_plan(P(w), P(b), P(d8), N, K)
o8 = torch.empty(M, N, dtype=torch.int8, device="cuda")
for _ in range(3):
    H.call("ivit_linear_i8_requant_planned", p8.p, P(x), 8, P(o8), M)
torch.cuda.synchronize()
buf = np.zeros(8 * 3 * 24, np.uint64)
H.lib.ivit_debug_plan_scratch(p8.p, buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes)
t = buf.reshape(8, 3, 6, 4).astype(np.int64)
names = ["post-barrier", "dma issued", "mfma+epi done", "wait done"]
for wv in (0, 1, 5):
    print(f"wave {wv}: per k-step PAIR cycles [barrier->dma | dma->compute done | compute done->wait done | wait done->next barrier release]   step total")
    for u in range(3):
        for kt in range(3):
            p0, p1, p2, p3 = t[wv, u, kt]
            nxt = t[wv, u, kt + 1, 0] if kt < 2 else (t[wv, u + 1, 0, 0] if u < 2 else 0)
            print(f"  unit {u} step {kt}: {p1-p0:6d} {p2-p1:6d} {p3-p2:6d} {(nxt-p3) if nxt else -1:6d}    {(nxt-p0) if nxt else -1:6d}")
