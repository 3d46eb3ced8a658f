"""GPU debugging aid: layernorm_requant on NS streams at once vs single stream; where do the outputs differ?"""
import ctypes, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import conftest  # noqa
import ivit_amd as iv
from ivit_amd import _lib
_P = ctypes.c_void_p
P = lambda t: _P(t.data_ptr())
C = int(sys.argv[1]) if len(sys.argv) > 1 else 192
M = int(sys.argv[2]) if len(sys.argv) > 2 else 25088
REP = int(sys.argv[3]) if len(sys.argv) > 3 else 40
NS = 8
streams = [torch.cuda.Stream() for _ in range(NS)]
hs = [_lib.Handle(0, s.cuda_stream) for s in streams]
rng = np.random.default_rng(3)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
xx = dev(rng.integers(-20000, 20000, (M, C)).astype(np.int16))
bb = dev(rng.normal(0, 3e5, C).astype(np.float32)); ss = dev((10 ** rng.uniform(-10.2, -9.8, C)).astype(np.float32))
dd = dev(iv.freeze.dyadic((10 ** rng.uniform(-10.2, -9.8, C)).astype(np.float32), np.float32(0.03)))
call = lambda h, o: h.call("ivit_layernorm_requant", P(xx), M, C, C, 0.01, P(bb), P(ss), P(dd), P(o))
ref = torch.empty(M, C, dtype=torch.int8, device="cuda"); call(hs[0], ref); torch.cuda.synchronize()
ref2 = torch.empty(M, C, dtype=torch.int8, device="cuda"); call(hs[1], ref2); torch.cuda.synchronize()
print("two single-stream runs equal:", torch.equal(ref, ref2))
nbad = 0
for r in range(REP):
    outs = [torch.full((M, C), 77, dtype=torch.int8, device="cuda") for _ in range(NS)]
    torch.cuda.synchronize()
    for i in range(NS): call(hs[i], outs[i])
    torch.cuda.synchronize()
    for i, o in enumerate(outs):
        if not torch.equal(o, ref):
            nbad += 1
            d = (o != ref).cpu().numpy()
            rows = np.nonzero(d.any(1))[0]
            print(f"rep {r} stream {i}: {int(d.sum())} bytes differ in {len(rows)} rows; rows {rows[:12]} (blocks of 32: {sorted(set(rows // 32))[:8]});"
                  f" cols of first row {np.nonzero(d[rows[0]])[0][:16]}; got {o[rows[0]].cpu().numpy()[np.nonzero(d[rows[0]])[0][:6]]} ref {ref[rows[0]].cpu().numpy()[np.nonzero(d[rows[0]])[0][:6]]}")
print("C", C, "M", M, ":", nbad, "of", REP * NS, "concurrent launches differ")
