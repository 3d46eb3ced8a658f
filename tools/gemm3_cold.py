"""GPU: the planned residual GEMM on the fc2 / proj shapes with HOT operands (back-to-back launches over the same buffers)
and COLD ones (a 1 GB fill between launches evicts L2 and the Infinity Cache), timed per launch with events.
The dispatch switches are compile-time now: build a scratch library with -DIVIT_OPT_GEMM3=0 (launch-per-tile kernels) or
-DIVIT_OPT_GEMM3_RES_MIN_N=0 (persistent residual flavour) and point IVIT_LIB at it."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ivit_amd as iv
from ivit_amd import _lib
P = lambda t: ctypes.c_void_p(t.data_ptr())
H = _lib.Handle(0, torch.cuda.current_stream().cuda_stream)
rng = np.random.default_rng(0)
big = torch.empty(1 << 30, dtype=torch.int8, device="cuda")
for name, (N, K) in {"fc2": (384, 1536), "proj": (384, 384), "fc1": (1536, 384)}.items():
    M = 50432
    x = torch.from_numpy(rng.integers(-128, 128, (M, K), dtype=np.int8)).cuda()
    w = torch.from_numpy(np.rint(rng.normal(0, 40, (N, K)).clip(-127, 127)).astype(np.int8)).cuda()
    b = torch.from_numpy(rng.integers(-20000, 20000, N).astype(np.int32)).cuda()
    d16 = torch.from_numpy(iv.freeze.dyadic((10 ** rng.uniform(-5.2, -5, N)).astype(np.float32), np.float32(2e-4))).cuda()
    d8 = torch.from_numpy(iv.freeze.dyadic((10 ** rng.uniform(-5.2, -5, N)).astype(np.float32), np.float32(4e-2))).cuda()
    dm = _lib.Dyadic(1.5e9, 2.0 ** -31); dr = _lib.Dyadic(1.2e9, 2.0 ** -30)
    r16 = torch.randint(-30000, 30000, (M, N), dtype=torch.int16, device="cuda")
    out = torch.empty(M, N, dtype=torch.int16, device="cuda")
    o8 = torch.empty(M, N, dtype=torch.int8, device="cuda")
    p16 = H.linear_plan(P(w), P(b), P(d16), N, K)
    p8 = H.linear_plan(P(w), P(b), P(d8), N, K)
    if name == "fc1":
        f = lambda: H.call("ivit_linear_i8_requant_planned", p8.p, P(x), 8, P(o8), M)
    else:
        f = lambda: H.call("ivit_linear_i8_requant_residual_planned", p16.p, P(x), dm, dr, P(r16), P(out), M)
    res = {}
    for mode in ("hot", "cold"):
        ts = []
        for i in range(12):
            if mode == "cold":
                big.fill_(i)
            a = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
            a.record(); f(); e.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(e) * 1e3)
        res[mode] = float(np.median(ts[2:]))
    print(f"{name:5s} hot {res['hot']:6.1f} us   cold {res['cold']:6.1f} us")
