"""GPU microbench: Swin window attention (stage-0 shape of Swin-T b256: 56 x 56 tokens, 3 heads), arithmetic vs table Shiftmax."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ivit_amd as iv
from ivit_amd import _lib
P = lambda t: ctypes.c_void_p(t.data_ptr())
H = _lib.Handle(0, torch.cuda.current_stream().cuda_stream)
rng = np.random.default_rng(0)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
B, R, Hh = 256, int(os.environ.get("WA_R", "56")), int(os.environ.get("WA_H", "3"))
scale = np.float32(os.environ.get("WA_S", "0.06"))
qkv = torch.randint(-128, 128, (B, R, R, 3 * Hh * 32), dtype=torch.int8, device="cuda")
relb = dev(rng.integers(-60, 60, (Hh, 49, 49)).astype(np.int16))
tabs = iv.freeze.shiftmax_tables(scale)
aq, et, cl = dev(tabs["aq"]), dev(tabs["t"]), dev(tabs["cls"])
dy = lambda a, b: (lambda d: _lib.Dyadic(float(d[0, 0]), float(d[0, 1])))(iv.freeze.dyadic(np.float32(a), np.float32(b)))
dqk, da, dpv = dy(3.1e-4, scale * 0.8), dy(scale * 0.8, scale), dy(4e-4, 0.03)
out = torch.empty(B, R * R, Hh * 32, dtype=torch.int8, device="cuda")
def arith(sh): H.call("ivit_window_attention_fused", P(qkv), dqk, da, P(relb), float(scale), dpv, P(out), B, R, 7, sh, Hh, 32)
def lut(sh): H.call("ivit_window_attention_fused_lut", P(qkv), dqk, da, P(relb), float(scale), P(aq), P(et), P(cl), int(tabs["NC"]),
                    int(tabs["t"].size), int(tabs["dmin"]), dpv, P(out), B, R, 7, sh, Hh, 32)
print("tables: %d classes, %.1f KB" % (tabs["NC"], (tabs["t"].size * 4 + tabs["aq"].size * 2 + 256) / 1024))
for name, f in (("arithmetic", arith), ("tables", lut)):
    for sh in (0, 3):
        for _ in range(3): f(sh)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for rep in range(5):
            a.record()
            for _ in range(10): f(sh)
            b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / 10 * 1000)
        print("%-10s shift %d: min %.1f median %.1f us" % (name, sh, min(ts), sorted(ts)[2]))
