import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ivit_amd as iv
from ivit_amd.swin_engine import SwinEngine
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "swin_tiny_b1.npz"))
sc = {k[6:]: np.float32(g[k]) for k in g.files if k.startswith("scale/")}
cfg = iv.SWIN_CONFIGS["swin_tiny"]
eng = SwinEngine(cfg, iv.make_swin_weights(cfg, int(g["seed"])), sc)
imgs = torch.from_numpy(iv.make_images_int8(cfg, 256, 5)).cuda()
for _ in range(5): eng.forward(imgs)
torch.cuda.synchronize()
