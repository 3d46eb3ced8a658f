"""GPU stress: DeiT-S b256 through the native runner, unsliced vs sliced, repeated; reports mismatching runs.
   python tools/vit_stress.py [runs] [slice counts, comma separated]"""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import load_golden, golden_scales
import ivit_amd as iv
g = load_golden(os.environ.get("STRESS_GOLDEN", "deit_small_b4.npz"))
cfg = iv.CONFIGS[str(g["cfg_name"])]
from ivit_amd.engine import ViTEngine
eng = ViTEngine.from_float(cfg, iv.make_vit_weights(cfg, int(g["seed"])), golden_scales(g))
B = int(os.environ.get("STRESS_BATCH", "256"))
imgs = np.concatenate([iv.make_images_int8(cfg, int(g["batch"]), int(g["images_seed"])), iv.make_images_int8(cfg, B - int(g["batch"]), seed=11)])
d = torch.from_numpy(imgs).cuda()
ref = eng.forward(d).clone().cpu().numpy()
print("golden prefix ok:", np.array_equal(ref[:int(g["batch"])], g["logits_int"]))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for ns in [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "4").split(",")]:
    bad, where = 0, []
    for i in range(n):
        o = eng.forward(d, nslices=ns).cpu().numpy()
        if not np.array_equal(o, ref):
            bad += 1
            where += list(np.nonzero((o != ref).any(1))[0])
    print(f"nslices {ns}: {bad} of {n} runs differ; images: {sorted(int(w) for w in where)}")
