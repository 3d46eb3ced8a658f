"""Swin-T on one MI355X through SwinEngine (fused windowed attention): throughput + per-kernel split."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ivit_amd as iv
from ivit_amd.swin_engine import SwinEngine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
g = np.load(os.path.join(ROOT, "tests", "golden", "swin_tiny_b1.npz"))
sc = {k[6:]: np.float32(g[k]) for k in g.files if k.startswith("scale/")}
cfg = iv.SWIN_CONFIGS["swin_tiny"]
eng = SwinEngine(cfg, iv.make_swin_weights(cfg, int(g["seed"])), sc)
for batch in [int(x) for x in os.environ.get("SW_BATCH", "1,64,256").split(",")]:
    imgs = torch.from_numpy(np.concatenate([iv.make_images_int8(cfg, 1, int(g["images_seed"])),
                                            iv.make_images_int8(cfg, batch, 5)])[:batch]).cuda()
    ok = bool(np.array_equal(eng.forward(imgs)[:1].cpu().numpy(), g["logits_int"]))
    for _ in range(2): eng.forward(imgs)
    torch.cuda.synchronize(); t = time.perf_counter()
    n = 10 if batch > 1 else 50
    for _ in range(n): eng.forward(imgs)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
    NS = int(os.environ.get("SW_SLICES", "4")) if batch >= 16 else 1
    rep = eng.capture(imgs, NS)
    for _ in range(2): rep()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): rep()
    torch.cuda.synchronize(); dtg = (time.perf_counter() - t) / n
    okg = bool(np.array_equal(rep()[:1].cpu().numpy(), g["logits_int"]))
    # per-operator HIP-event timing
    recs, orig = [], eng.h.call
    def call(name, *a):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); orig(name, *a); e1.record()
        if name.startswith("ivit_linear_i8_requant"): name += " MNK=" + "x".join(str(v) for v in a[-3:])
        recs.append((name, e0, e1))
    eng.h.call = call
    eng.forward_ops(imgs); torch.cuda.synchronize(); eng.h.call = orig
    per = {}
    for name, e0, e1 in recs:
        d = per.setdefault(name, [0.0, 0]); d[0] += e0.elapsed_time(e1); d[1] += 1
    print(json.dumps({"config": "swin_tiny (SwinEngine, fused windowed attention)", "batch": batch, "ms": round(dt * 1e3, 3),
                      "images_per_s": round(batch / dt, 1), "bit_exact_vs_reference_golden": ok,
                      "hipgraph_slices": NS, "hipgraph_ms": round(dtg * 1e3, 3), "hipgraph_images_per_s": round(batch / dtg, 1), "hipgraph_bit_exact": okg,
                      "kernel_ms": {k: [round(v[0], 3), v[1]] for k, v in sorted(per.items(), key=lambda kv: -kv[1][0])}}), flush=True)
