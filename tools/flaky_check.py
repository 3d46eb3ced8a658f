import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ivit_amd as iv
from ivit_amd.engine import ViTEngine
g = np.load("tests/golden/deit_small_b4.npz")
sc = {k[6:]: np.float32(g[k]) for k in g.files if k.startswith("scale/")}
cfg = iv.CONFIGS["deit_small"]
eng = ViTEngine.from_float(cfg, iv.make_vit_weights(cfg, int(g["seed"])), sc)
B = int(os.environ.get("FB", "8")); NS = int(os.environ.get("FS", "4"))
imgs = torch.from_numpy(np.concatenate([iv.make_images_int8(cfg, 4, int(g["images_seed"])), iv.make_images_int8(cfg, B, 7)])[:B]).cuda()
ref = eng.forward(imgs).cpu().numpy()
print("ref ok", np.array_equal(ref[:4], g["logits_int"]))
bad = 0
for it in range(int(os.environ.get("FN", "40"))):
    out = eng.forward(imgs, nslices=NS).cpu().numpy()
    if not np.array_equal(out, ref):
        rows = np.where((out != ref).any(1))[0]
        print("iter", it, "sliced mismatch rows", rows.tolist(), "n elems", int((out != ref).sum())); bad += 1
rep = eng.capture(imgs, NS)
for it in range(int(os.environ.get("FN", "40"))):
    out = rep(); torch.cuda.synchronize(); out = out.cpu().numpy()
    if not np.array_equal(out, ref):
        rows = np.where((out != ref).any(1))[0]
        print("iter", it, "graph mismatch rows", rows.tolist(), "n elems", int((out != ref).sum())); bad += 1
for it in range(int(os.environ.get("FN", "40"))):
    out = eng.forward(imgs).cpu().numpy()
    if not np.array_equal(out, ref): print("iter", it, "single-stream mismatch"); bad += 1
print("bad", bad)
