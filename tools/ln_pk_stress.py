"""Concurrency stress for the register-resident LayerNorm: C = 384 and C = 768 launches on 8 streams, each beside MFMA-issuing
aggressors on the same streams — the K = 48 patch-embedding GEMM on gemm_nt_kernel (AGPR accumulators: the aggressor of the
round-4/5 packed-fp32 fault) and a K = 384 QuantLinear — every output compared with the single-stream result.  Written in round 6 for
the hand-packed variant (tools/experiments/ivit_layernorm_pk.h: 0 of 60 000 launches differed while it was integrated); runs against
whatever LayerNorm the library dispatches.  usage: python tools/ln_pk_stress.py [launches per shape, default 20000]."""
import ctypes, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ivit_amd as iv
from ivit_amd import _lib
P = lambda t: ctypes.c_void_p(t.data_ptr())
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
total = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
NS = 8
rng = np.random.default_rng(11)
streams = [torch.cuda.Stream() for _ in range(NS)]
hs = [_lib.Handle(0, st.cuda_stream) for st in streams]
aggr = []
for (K, N, M) in ((48, 96, 100352), (384, 1152, 6272), (48, 384, 50176)):
    x = dev(rng.integers(-128, 128, (M, K), dtype=np.int8)); w = dev(rng.integers(-128, 128, (N, K), dtype=np.int8))
    b = dev(rng.integers(-3000, 3000, N).astype(np.int32))
    d = dev(iv.freeze.dyadic((10 ** rng.uniform(-5.6, -5.2, N)).astype(np.float32), np.float32(0.012)))
    o = [torch.empty(M, N, dtype=torch.int8, device="cuda") for _ in range(NS)]
    aggr.append(lambda h, i, x=x, w=w, b=b, d=d, o=o, M=M, N=N, K=K: h.call("ivit_linear_i8_requant", P(x), P(w), P(b), P(d), 8, P(o[i]), M, N, K))
victims = []
for (C, M) in ((384, 6272), (768, 3136), (384, 25216)):
    xx = dev(rng.integers(-20000, 20000, (M, C)).astype(np.int16))
    bb = dev(rng.normal(0, 3e5, C).astype(np.float32)); ss = dev((10 ** rng.uniform(-10.2, -9.8, C)).astype(np.float32) * rng.choice([-1.0, 1.0], C).astype(np.float32))
    dd = dev(iv.freeze.dyadic(ss.cpu().numpy(), np.float32(0.03)))
    call = lambda h, o, xx=xx, bb=bb, ss=ss, dd=dd, M=M, C=C: h.call("ivit_layernorm_requant", P(xx), M, C, C, 0.01, P(bb), P(ss), P(dd), P(o))
    ref = torch.empty(M, C, dtype=torch.int8, device="cuda")
    call(hs[0], ref); torch.cuda.synchronize()
    victims.append((C, M, call, ref, [torch.empty_like(ref) for _ in range(NS)]))
t0 = time.time()
bad = {v[:2]: 0 for v in victims}
done = {v[:2]: 0 for v in victims}
rounds = (total + NS - 1) // NS
for r in range(rounds):
    for vi, (C, M, call, ref, outs) in enumerate(victims):
        for i in range(NS):
            aggr[(r + i) % len(aggr)](hs[i], i)
            call(hs[i], outs[i])
            aggr[(r + i + 1) % len(aggr)](hs[i], i)
    torch.cuda.synchronize()
    for (C, M, call, ref, outs) in victims:
        for i in range(NS):
            done[(C, M)] += 1
            if not torch.equal(outs[i], ref):
                bad[(C, M)] += 1
for k in bad:
    print(f"layernorm_requant C {k[0]} rows {k[1]}: {bad[k]} of {done[k]} launches beside MFMA aggressors differ from the single-stream result")
print(f"{time.time() - t0:.1f} s")
sys.exit(1 if any(bad.values()) else 0)
