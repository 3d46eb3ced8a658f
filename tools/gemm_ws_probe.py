"""D = 384 qkv QuantLinear: the library's kernels (ivit_layernorm_requant + ivit_linear_i8_qkv_planned, v row-major) against
i-vit_amd/csrc/ivit_gemm_ws.h built alone (tools/ubench/libgemm_ws_probe.so) — the GEMM on the same 8-bit activations, and
norm1 + GEMM fused on the 16-bit input; outputs compared byte for byte."""
import ctypes, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import ivit_amd as iv
from ivit_amd import _lib
P = lambda t: ctypes.c_void_p(t.data_ptr())
H = _lib.Handle(0, torch.cuda.current_stream().cuda_stream)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
probe = ctypes.CDLL(os.path.join(ROOT, "tools", "ubench", os.environ.get("WS_PROBE_SO", "libgemm_ws_probe.so")))
I, V, F = ctypes.c_int, ctypes.c_void_p, ctypes.c_float
probe.gemm_ws_probe.argtypes = [V] * 7 + [I] * 7 + [ctypes.POINTER(F), I, V, F, V, V, V]
DB = ctypes.c_double
probe.gemm_ws_probe_res.argtypes = [V] * 6 + [DB, DB] + [I] * 5 + [ctypes.POINTER(F)]
def timeit(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    ts = []
    for rep in range(5):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): f()
        e.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(e) * 1000 / n)
    return min(ts)
rng = np.random.default_rng(0)
for name, B, T, D, Hh in (("deit_small b256", 256, 197, 384, 6), ("b3 (ragged)", 3, 197, 384, 6), ("b100", 100, 197, 384, 6), ("b1", 1, 197, 384, 6)):
    M, dh = B * T, 64
    # norm1: the block's 16-bit input and the LayerNorm's constants
    wln = rng.normal(1.0, 0.4, D).astype(np.float32) * rng.choice([-1.0, 1.0], D).astype(np.float32)
    bln = rng.normal(0.0, 0.5, D).astype(np.float32)
    bias_int, sc = iv.freeze.layernorm_constants(wln, bln)
    s_in, s_out = np.float32(7.3e-4), np.float32(0.031)
    dln = dev(iv.freeze.dyadic(sc, s_out))
    x16np = rng.integers(-26000, 26000, (M, D)).astype(np.int16); x16np[:, : D // 2] //= 64
    x16, bi_d, sc_d = dev(x16np), dev(bias_int), dev(sc)
    x = torch.empty(M, D, dtype=torch.int8, device="cuda")
    ln = lambda: H.call("ivit_layernorm_requant", P(x16), M, D, D, float(s_in), P(bi_d), P(sc_d), P(dln), P(x))
    ln(); torch.cuda.synchronize()
    w = dev(rng.integers(-128, 128, (3 * D, D), dtype=np.int8))
    b = dev(rng.integers(-1000, 1000, 3 * D).astype(np.int32))
    dnp = iv.freeze.dyadic((10 ** rng.uniform(-5.2, -4.2, 3 * D)).astype(np.float32), np.float32(0.02))
    d = dev(dnp)
    cq = dev(dnp[:, 0] * dnp[:, 1])
    plan = H.linear_plan(P(w), P(b), P(d), 3 * D, D)
    ref = [torch.zeros(B * Hh, T, dh, dtype=torch.int8, device="cuda") for _ in range(3)]
    out = [torch.zeros(B * Hh, T, dh, dtype=torch.int8, device="cuda") for _ in range(3)]
    f = lambda: H.call("ivit_linear_i8_qkv_planned", plan.p, P(x), P(ref[0]), P(ref[1]), P(ref[2]), B, T, Hh, dh, 0)
    t_ln, t_qkv = timeit(ln), timeit(f)
    sat = sum(int(((r == 127) | (r == -128)).sum()) for r in ref) * 100 // (3 * ref[0].numel())
    ops = 2.0 * M * 3 * D * D
    line = f"{name:16s} library LayerNorm {t_ln:5.1f} + qkv {t_qkv:5.1f} us ({ops/t_qkv/1e6:5.0f} TOPS)"
    for mode in (0, 1):
        for o in out: o.zero_()
        us = F(0)
        rc = probe.gemm_ws_probe(P(x), P(w), P(b), P(cq), P(out[0]), P(out[1]), P(out[2]), M, 3 * D, T, Hh, 1, int(os.environ.get("WS_GRID", "256")), 10,
                                 ctypes.byref(us), mode, P(x16), float(s_in), P(bi_d), P(sc_d), P(dln))
        torch.cuda.synchronize()
        diff = sum(int((o != r).sum()) for o, r in zip(out, ref))
        line += f" | {'LayerNorm + qkv fused' if mode else 'ws qkv'} {us.value:5.1f} us rc {rc}, {diff} bytes differ"
    print(line + f" | {sat} % saturated, {len(torch.unique(x))} LayerNorm levels", flush=True)

# attn.proj + the residual QuantAct on the same kernel (EPI_RES16) against ivit_linear_i8_requant_residual_planned
for name, M in (("deit_small b256", 256 * 197), ("b3", 3 * 197), ("b100", 100 * 197)):
    D = 384
    x = dev(rng.integers(-128, 128, (M, D), dtype=np.int8))
    w = dev(np.rint(rng.normal(0, 45, (D, D)).clip(-128, 127)).astype(np.int8))
    b = dev(rng.integers(-2 ** 14, 2 ** 14, D).astype(np.int32))
    dnp = iv.freeze.dyadic((10 ** rng.uniform(-5.5, -5, D)).astype(np.float32), np.float32(2e-4))
    d, cq = dev(dnp), dev(dnp[:, 0] * dnp[:, 1])
    res = dev(rng.integers(-30000, 30000, (M, D)).astype(np.int16))
    dm, dr = iv.freeze.dyadic(np.float32(2e-4), np.float32(7.3e-4)), iv.freeze.dyadic(np.float32(6.9e-4), np.float32(7.3e-4))
    plan = H.linear_plan(P(w), P(b), P(d), D, D)
    ref = torch.zeros(M, D, dtype=torch.int16, device="cuda"); out = torch.zeros_like(ref)
    f = lambda: H.call("ivit_linear_i8_requant_residual_planned", plan.p, P(x), _lib.Dyadic(float(dm[0, 0]), float(dm[0, 1])),
                       _lib.Dyadic(float(dr[0, 0]), float(dr[0, 1])), P(res), P(ref), M)
    t_lib = timeit(f)
    us = F(0)
    rc = probe.gemm_ws_probe_res(P(x), P(w), P(b), P(cq), P(res), P(out), float(dm[0, 0] * dm[0, 1]), float(dr[0, 0] * dr[0, 1]), M, D, 1,
                                 int(os.environ.get("WS_GRID", "256")), 10, ctypes.byref(us))
    torch.cuda.synchronize()
    print(f"proj + residual {name:16s} library {t_lib:5.1f} us | ws {us.value:5.1f} us rc {rc}, {int((out != ref).sum())} values differ, "
          f"{int(((ref == 32767) | (ref == -32768)).sum()) * 100 // ref.numel()} % saturated, {len(torch.unique(ref))} levels", flush=True)
