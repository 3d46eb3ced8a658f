"""per-operator HIP-event split of one ViTEngine.forward_ops for any config (VS_CFG, VS_BATCH)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ivit_amd as iv
from ivit_amd.engine import ViTEngine
name, batch = os.environ.get("VS_CFG", "vit_base_384"), int(os.environ.get("VS_BATCH", "128"))
fix = {"vit_base_384": "vit_base_384_b1.npz", "deit_base": "deit_base_b2.npz", "deit_small": "deit_small_b4.npz", "deit_tiny": "deit_tiny_b1.npz"}[name]
g = np.load(os.path.join("tests", "golden", fix))
sc = {k[6:]: np.float32(g[k]) for k in g.files if k.startswith("scale/")}
cfg = iv.CONFIGS[name]
eng = ViTEngine.from_float(cfg, iv.make_vit_weights(cfg, 0), sc)
imgs = torch.from_numpy(iv.make_images_int8(cfg, batch, 1)).cuda()
eng.forward_ops(imgs); torch.cuda.synchronize()
recs, orig = [], eng.h.call
def call(n, *a):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); orig(n, *a); e1.record(); recs.append((n, e0, e1))
eng.h.call = call
eng.forward_ops(imgs); torch.cuda.synchronize(); eng.h.call = orig
per = {}
for n, e0, e1 in recs:
    d = per.setdefault(n, [0.0, 0]); d[0] += e0.elapsed_time(e1); d[1] += 1
tot = sum(v[0] for v in per.values())
print(name, "batch", batch, "total ms", round(tot, 3))
for k, v in sorted(per.items(), key=lambda kv: -kv[1][0]): print(f"  {v[0]:8.3f} ms x{v[1]:3d}  {k}")
