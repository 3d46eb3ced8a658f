"""GPU microbench: the planned qkv QuantLinear (ivit_linear_i8_qkv_planned) at DeiT-S b256 / DeiT-B b64 shapes, V^T scatter
(ldv = padded token count) against row-major V (ldv = 0, round 6).  QB_CHECK=1 compares the two V layouts."""
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
    best = 1e9
    for _ in range(4):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): fn()
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / n * 1e3)
    return best
rng = np.random.default_rng(0)
for name, B, T, D, Hh in (("deit_small b256", 256, 197, 384, 6), ("deit_base b64", 64, 197, 768, 12), ("vit_base_384 b128", 128, 577, 768, 12)):
    M, dh, ld = B * T, 64, (T + 15) // 16 * 16
    x = torch.from_numpy(rng.integers(-128, 128, (M, D), dtype=np.int8)).cuda()
    w = torch.from_numpy(rng.integers(-128, 128, (3 * D, D), dtype=np.int8)).cuda()
    b = torch.from_numpy(rng.integers(-1000, 1000, 3 * D).astype(np.int32)).cuda()
    d = torch.from_numpy(iv.freeze.dyadic((10 ** rng.uniform(-5.2, -5, 3 * D)).astype(np.float32), np.float32(0.02))).cuda()
    plan = H.linear_plan(P(w), P(b), P(d), 3 * D, D)
    q = torch.empty(B * Hh, T, dh, dtype=torch.int8, device="cuda"); k = torch.empty_like(q)
    vt = torch.zeros(B * Hh, dh, ld, dtype=torch.int8, device="cuda"); vr = torch.zeros(B * Hh, T, dh, dtype=torch.int8, device="cuda")
    ops = 2.0 * M * 3 * D * D
    t1 = timeit(lambda: H.call("ivit_linear_i8_qkv_planned", plan.p, P(x), P(q), P(k), P(vt), B, T, Hh, dh, ld))
    line = f"{name:18s} qkv planned, V^T scatter {t1:7.1f} us ({ops/t1/1e6:6.0f} TOPS)"
    if os.environ.get("QB_ROWV", "1") == "1":
        try:
            t2 = timeit(lambda: H.call("ivit_linear_i8_qkv_planned", plan.p, P(x), P(q), P(k), P(vr), B, T, Hh, dh, 0))
            line += f" | row-major V {t2:7.1f} us ({ops/t2/1e6:6.0f} TOPS)"
            if os.environ.get("QB_CHECK"):
                line += f" | V layouts agree: {bool(torch.equal(vt[:, :, :T].transpose(1, 2), vr))}"
        except _lib.IvitError as e:
            line += f" | row-major V: {e}"
    print(line, flush=True)
