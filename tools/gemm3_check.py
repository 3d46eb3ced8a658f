"""GPU: A/B of the persistent pipelined GEMM (ivit_gemm3.h, *_planned entry points) against the
launch-per-tile kernels (bit-exact comparison on random operands) and timing of both.
usage: python tools/gemm3_check.py [--time-only] [--shapes qkv,fc1]"""
import ctypes, sys, os, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ivit_amd as iv
from ivit_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--time-only", action="store_true")
ap.add_argument("--shapes", default="qkv,proj,fc1,fc2")
ap.add_argument("--M", type=int, default=50432)
ap.add_argument("--reps", type=int, default=20)
args = ap.parse_args()

P = lambda t: ctypes.c_void_p(t.data_ptr())
H = _lib.Handle(0, torch.cuda.current_stream().cuda_stream)


def timeit(fn, n):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


rng = np.random.default_rng(0)
SHAPES = {"qkv": (1152, 384), "proj": (384, 384), "fc1": (1536, 384), "fc2": (384, 1536)}
ok_all = True
for name in args.shapes.split(","):
    N, K = SHAPES[name]
    for M in ([args.M] if args.time_only else [args.M, args.M - 37, 300, 128]):
        x = torch.from_numpy(rng.integers(-128, 128, (M, K), dtype=np.int8)).cuda()
        # weights like a quantised layer: per-row max 127, gaussian bulk (keeps sum|w| realistic)
        wf = rng.normal(0, 40, (N, K)).clip(-127, 127)
        w = torch.from_numpy(np.rint(wf).astype(np.int8)).cuda()
        b = torch.from_numpy(rng.integers(-20000, 20000, N).astype(np.int32)).cuda()
        d8 = torch.from_numpy(iv.freeze.dyadic((10 ** rng.uniform(-5.2, -5, N)).astype(np.float32), np.float32(0.04))).cuda()
        d16 = torch.from_numpy(iv.freeze.dyadic((10 ** rng.uniform(-5.2, -5, N)).astype(np.float32), np.float32(2e-4))).cuda()
        dm = _lib.Dyadic(1.5e9, 2.0 ** -31); dr = _lib.Dyadic(1.2e9, 2.0 ** -30)
        r16 = torch.randint(-30000, 30000, (M, N), dtype=torch.int16, device="cuda")
        p8 = H.linear_plan(P(w), P(b), P(d8), N, K)
        p16 = H.linear_plan(P(w), P(b), P(d16), N, K)
        o8a = torch.empty(M, N, dtype=torch.int8, device="cuda"); o8b = torch.zeros_like(o8a)
        o16a = torch.empty(M, N, dtype=torch.int16, device="cuda"); o16b = torch.zeros_like(o16a)
        ora = torch.empty(M, N, dtype=torch.int16, device="cuda"); orb = torch.zeros_like(ora)
        f_old8 = lambda: H.call("ivit_linear_i8_requant", P(x), P(w), P(b), P(d8), 8, P(o8a), M, N, K)
        f_new8 = lambda: H.call("ivit_linear_i8_requant_planned", p8.p, P(x), 8, P(o8b), M)
        f_old16 = lambda: H.call("ivit_linear_i8_requant", P(x), P(w), P(b), P(d16), 16, P(o16a), M, N, K)
        f_new16 = lambda: H.call("ivit_linear_i8_requant_planned", p16.p, P(x), 16, P(o16b), M)
        f_oldr = lambda: H.call("ivit_linear_i8_requant_residual", P(x), P(w), P(b), P(d16), dm, dr, P(r16), P(ora), M, N, K)
        f_newr = lambda: H.call("ivit_linear_i8_requant_residual_planned", p16.p, P(x), dm, dr, P(r16), P(orb), M)
        tag = f"{name:5s} M={M:6d} N={N:5d} K={K:5d} plan(pipelined={p8.pipelined_ok},{p16.pipelined_ok} fma={p8.single_fma_ok},{p16.single_fma_ok})"
        if not args.time_only:
            f_old8(); f_new8(); f_old16(); f_new16(); f_oldr(); f_newr()
            torch.cuda.synchronize()
            e8 = int((o8a != o8b).sum()); e16 = int((o16a != o16b).sum()); er = int((ora != orb).sum())
            line = f"{tag}  mismatches rq8 {e8} rq16 {e16} res {er}"
            if name == "qkv" and M % 197 == 0:
                B_, T, Hh, dh = M // 197, 197, 6, 64
                ld = 208
                qa = torch.zeros(B_ * Hh * T * dh, dtype=torch.int8, device="cuda"); ka = torch.zeros_like(qa)
                va = torch.zeros(B_ * Hh * dh * ld, dtype=torch.int8, device="cuda")
                qb = torch.zeros_like(qa); kb = torch.zeros_like(qa); vb = torch.zeros_like(va)
                H.call("ivit_linear_i8_qkv", P(x), P(w), P(b), P(d8), P(qa), P(ka), P(va), B_, T, Hh, dh, ld)
                H.call("ivit_linear_i8_qkv_planned", p8.p, P(x), P(qb), P(kb), P(vb), B_, T, Hh, dh, ld)
                torch.cuda.synchronize()
                line += f" qkv {int((qa != qb).sum())}/{int((ka != kb).sum())}/{int((va != vb).sum())}"
                e8 += int((qa != qb).sum()) + int((ka != kb).sum()) + int((va != vb).sum())
            print(line, flush=True)
            ok_all &= (e8 == 0 and e16 == 0 and er == 0)
            # repeat the pipelined launches: races show up as run-to-run differences
            for _ in range(5):
                o8c = torch.zeros_like(o8b);
                H.call("ivit_linear_i8_requant_planned", p8.p, P(x), 8, P(o8c), M)
                orc_ = torch.zeros_like(orb)
                H.call("ivit_linear_i8_requant_residual_planned", p16.p, P(x), dm, dr, P(r16), P(orc_), M)
                torch.cuda.synchronize()
                if int((o8c != o8a).sum()) or int((orc_ != ora).sum()):
                    print("   RERUN MISMATCH", int((o8c != o8a).sum()), int((orc_ != ora).sum())); ok_all = False
        if M == args.M:
            ops = 2.0 * M * N * K
            t = [timeit(f, args.reps) for f in (f_old8, f_new8, f_old16, f_new16, f_oldr, f_newr)]
            print(f"{tag}\n      rq8 old {t[0]:6.1f} new {t[1]:6.1f} us ({ops/t[1]/1e6:5.0f} TOPS) | rq16 old {t[2]:6.1f} new {t[3]:6.1f} ({ops/t[3]/1e6:5.0f}) | "
                  f"res old {t[4]:6.1f} new {t[5]:6.1f} ({ops/t[5]/1e6:5.0f})", flush=True)
            if name == "qkv":
                B_, T, Hh, dh, ld = M // 197, 197, 6, 64, 208
                if B_ * 197 == M:
                    qa = torch.zeros(B_ * Hh * T * dh, dtype=torch.int8, device="cuda"); ka = torch.zeros_like(qa)
                    va = torch.zeros(B_ * Hh * dh * ld, dtype=torch.int8, device="cuda")
                    t0 = timeit(lambda: H.call("ivit_linear_i8_qkv", P(x), P(w), P(b), P(d8), P(qa), P(ka), P(va), B_, T, Hh, dh, ld), args.reps)
                    t1 = timeit(lambda: H.call("ivit_linear_i8_qkv_planned", p8.p, P(x), P(qa), P(ka), P(va), B_, T, Hh, dh, ld), args.reps)
                    print(f"      qkv-scatter old {t0:6.1f} new {t1:6.1f} us ({ops/t1/1e6:5.0f} TOPS)", flush=True)
print("ALL EXACT" if ok_all else "MISMATCHES")
