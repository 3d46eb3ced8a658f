import sys, os
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from conftest import load_golden, golden_scales
import ivit_amd as iv
from ivit_amd.engine import ViTEngine
from ivit_amd.swin_engine import SwinEngine
g = load_golden("deit_small_b4.npz"); cfg = iv.CONFIGS[str(g["cfg_name"])]
eng = ViTEngine.from_float(cfg, iv.make_vit_weights(cfg, int(g["seed"])), golden_scales(g))
for B in (1, 5, 37, 130, 255):
    imgs = torch.from_numpy(iv.make_images_int8(cfg, B, 11)).cuda()
    ref = eng.forward(imgs, nslices=1).cpu().numpy()
    for ns in (2, 3, 5, 8):
        if ns > B: continue
        out = eng.forward(imgs, nslices=ns).cpu().numpy()
        assert np.array_equal(out, ref), (B, ns)
    print("deit_small B", B, "ok", flush=True)
g = load_golden("swin_tiny_b1.npz"); cfg = iv.SWIN_CONFIGS[str(g["cfg_name"])]
eng = SwinEngine(cfg, iv.make_swin_weights(cfg, int(g["seed"])), golden_scales(g))
for B in (1, 3, 9, 70, 131):
    imgs = torch.from_numpy(iv.make_images_int8(cfg, B, 12)).cuda()
    ref = eng.forward(imgs, nslices=1).cpu().numpy()
    for ns in (2, 3, 4, 7):
        if ns > B: continue
        out = eng.forward(imgs, nslices=ns).cpu().numpy()
        assert np.array_equal(out, ref), (B, ns)
    print("swin_tiny B", B, "ok", flush=True)
