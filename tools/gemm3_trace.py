"""GPU: dump the k-step timeline of gemm_as_kernel (library built with -DG3_TRACE=1, IVIT_LIB pointing at it)."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ivit_amd as iv
from ivit_amd import _lib
P = lambda t: ctypes.c_void_p(t.data_ptr())
H = _lib.Handle(0, torch.cuda.current_stream().cuda_stream)
rng = np.random.default_rng(0)
M, N, K = 50432, int(os.environ.get("TR_N", 1536)), int(os.environ.get("TR_K", 384))
x = torch.from_numpy(rng.integers(-128, 128, (M, K), dtype=np.int8)).cuda()
w = torch.from_numpy(np.rint(rng.normal(0, 40, (N, K)).clip(-127, 127)).astype(np.int8)).cuda()
b = torch.from_numpy(rng.integers(-20000, 20000, N).astype(np.int32)).cuda()
d8 = torch.from_numpy(iv.freeze.dyadic((10 ** rng.uniform(-5.2, -5, N)).astype(np.float32), np.float32(0.04))).cuda()
p8 = H.linear_plan(P(w), P(b), P(d8), N, K)
BITS = int(os.environ.get("TR_BITS", 8))
o8 = torch.empty(M, N, dtype=torch.int8 if BITS == 8 else torch.int16, device="cuda")
for _ in range(3):
    H.call("ivit_linear_i8_requant_planned", p8.p, P(x), BITS, P(o8), M)
torch.cuda.synchronize()
buf = np.zeros(8 * 3 * 24, np.uint64)
H.lib.ivit_debug_plan_scratch(p8.p, buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes)
t = buf.reshape(8, 3, 3, 8).astype(np.int64)
print("per 128-column k-step, cycles: [S0 | S1 | S2 | counted wait | barrier | S3]   step total")
for wv in (0, 5):
    print("wave", wv)
    for u in range(3):
        for kt in range(3):
            p = t[wv, u, kt]
            nxt = t[wv, u, kt + 1, 0] if kt < 2 else (t[wv, u + 1, 0, 0] if u < 2 else 0)
            seg = [p[1] - p[0], p[2] - p[1], p[4] - p[2], p[5] - p[4], p[3] - p[5], (nxt - p[3]) if nxt else -1]
            print(f"  unit {u} step {kt}: " + " ".join(f"{int(v):6d}" for v in seg) + f"    {(nxt - p[0]) if nxt else -1:6d}")

print("busy cycles per step (step total - barrier wait), unit 1, all waves: [step0 step1 step2] | S3 of each step")
for wv in range(8):
    row, s3 = [], []
    for kt in range(3):
        p = t[wv, 1, kt]
        nxt = t[wv, 1, kt + 1, 0] if kt < 2 else t[wv, 2, 0, 0]
        row.append(int(nxt - p[0] - (p[3] - p[5])))
        s3.append(int(nxt - p[3]))
    print(f"  wave {wv} (dma section {(wv + (wv >> 2) * 2 + 3) & 3}): {row}  | {s3}")
