"""GPU microbench: time the QuantLinear GEMM entry points at DeiT-S b256 shapes."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ivit_amd as iv
from ivit_amd import _lib
P = lambda t: ctypes.c_void_p(t.data_ptr())
H = _lib.Handle(0, torch.cuda.current_stream().cuda_stream)
M = int(os.environ.get("GB_M", "50432"))       # Swin-T b256: stage 0 GB_M=802816 (sq0, sp0), stage 1 GB_M=200704 (sq1, sp1, sf1)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
rng = np.random.default_rng(0)
SHAPES = [("qkv", 1152, 384), ("proj", 384, 384), ("fc1", 1536, 384), ("fc2", 384, 1536)]
SWIN = [("sq0", 288, 96), ("sp0", 96, 96), ("sq1", 576, 192), ("sp1", 192, 192), ("sf1", 768, 192)]
if os.environ.get("GB_SHAPES"): SHAPES = [x for x in SHAPES + SWIN if x[0] in os.environ["GB_SHAPES"].split(",")]
for name, N, K in SHAPES:
    x = torch.from_numpy(rng.integers(-128, 128, (M, K), dtype=np.int8)).cuda()
    w = torch.from_numpy(rng.integers(-128, 128, (N, K), dtype=np.int8)).cuda()
    b = torch.from_numpy(rng.integers(-1000, 1000, N).astype(np.int32)).cuda()
    d = torch.from_numpy(iv.freeze.dyadic((10 ** rng.uniform(-5.2, -5, N)).astype(np.float32), np.float32(0.02))).cuda()
    d16 = torch.from_numpy(iv.freeze.dyadic((10 ** rng.uniform(-5.2, -5, N)).astype(np.float32), np.float32(2e-4))).cuda()
    o8 = torch.empty(M, N, dtype=torch.int8, device="cuda")
    o16 = torch.empty(M, N, dtype=torch.int16, device="cuda")
    r16 = torch.randint(-30000, 30000, (M, N), dtype=torch.int16, device="cuda")
    o32 = torch.empty(M, N, dtype=torch.int32, device="cuda")
    ops = 2.0 * M * N * K
    dm = _lib.Dyadic(1.5e9, 2.0 ** -31); dr = _lib.Dyadic(1.2e9, 2.0 ** -30)
    t8 = timeit(lambda: H.call("ivit_linear_i8_requant", P(x), P(w), P(b), P(d), 8, P(o8), M, N, K))
    t16 = timeit(lambda: H.call("ivit_linear_i8_requant", P(x), P(w), P(b), P(d16), 16, P(o16), M, N, K))
    tr = timeit(lambda: H.call("ivit_linear_i8_requant_residual", P(x), P(w), P(b), P(d16), dm, dr, P(r16), P(o16), M, N, K))
    t32 = timeit(lambda: H.call("ivit_linear_i8", P(x), P(w), P(b), P(o32), M, N, K))
    print(f"{name:5s} N={N:5d} K={K:5d}  rq8 {t8:7.1f} us ({ops/t8/1e6:6.0f} TOPS)  rq16 {t16:7.1f} us ({ops/t16/1e6:6.0f})  "
          f"rq16+res {tr:7.1f} us ({ops/tr/1e6:6.0f})  raw32(old kernel) {t32:7.1f} us ({ops/t32/1e6:6.0f})")
