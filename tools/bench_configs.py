"""Throughput of the other BASELINE.json configs on ONE MI355X (per-GPU shares of configs 3/5,
config 4 through the operator-surface chain).  Scales: reference calibration where a golden
fixture exists.  Prints one JSON line per config."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ivit_amd as iv
from ivit_amd.engine import ViTEngine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def scales_of(f):
    g = np.load(os.path.join(ROOT, "tests", "golden", f))
    return g, {k[6:]: np.float32(g[k]) for k in g.files if k.startswith("scale/")}

def timeit(fn, n):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n

for name, fixture, batch, streams in [("deit_tiny", "deit_tiny_b1.npz", 1, 1), ("deit_base", "deit_base_b2.npz", 64, 2),
                                      ("vit_base_384", "vit_base_384_b1.npz", 128, 4)]:
    g, sc = scales_of(fixture)
    cfg = iv.CONFIGS[name]
    eng = ViTEngine.from_float(cfg, iv.make_vit_weights(cfg, 0), sc)
    imgs = torch.from_numpy(iv.make_images_int8(cfg, batch, 1)).cuda()
    ok = bool(np.array_equal(eng.forward(imgs)[:int(g["batch"])].cpu().numpy(), g["logits_int"])) if batch >= int(g["batch"]) else None
    step = eng.capture(imgs, streams)
    dt = timeit(step, 10 if batch > 1 else 200)
    print(json.dumps({"config": name, "batch": batch, "streams": streams, "hipgraph": True, "ms": round(dt * 1e3, 4),
                      "images_per_s": round(batch / dt, 1), "bit_exact_vs_reference_golden": ok}), flush=True)

# config 4: Swin-T through the reference-shaped operator chain (generic kernels + torch permutations)
from ivit_amd.swin_quant import SwinTransformer
g, sc = scales_of("swin_tiny_b1.npz")
cfg = iv.SWIN_CONFIGS["swin_tiny"]
m = SwinTransformer(img_size=cfg.img_size, patch_size=cfg.patch_size, num_classes=cfg.num_classes, embed_dim=cfg.embed_dim,
                    depths=cfg.depths, num_heads=cfg.num_heads, window_size=cfg.window_size, mlp_ratio=cfg.mlp_ratio)
m.load_float_weights(iv.make_swin_weights(cfg, 0)).load_act_scales(sc)
iv.freeze_model(m)
for batch in (1, 32):
    imgs = torch.from_numpy(iv.make_images_int8(cfg, batch, 1)).cuda()
    with torch.no_grad():
        acc, _ = m(imgs)
        ok = bool(np.array_equal(acc[:1].cpu().numpy(), g["logits_int"]))
        dt = timeit(lambda: m(imgs), 3)
    print(json.dumps({"config": "swin_tiny (operator-surface chain, not fused)", "batch": batch, "ms": round(dt * 1e3, 3),
                      "images_per_s": round(batch / dt, 1), "bit_exact_vs_reference_golden": ok}), flush=True)
