import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import ivit_amd as iv
from conftest import load_golden, golden_scales
from ivit_amd.engine import ViTEngine
g = load_golden("deit_small_b4.npz")
cfg = iv.CONFIGS[str(g["cfg_name"])]
eng = ViTEngine.from_float(cfg, iv.make_vit_weights(cfg, int(g["seed"])), golden_scales(g), device="cuda:0")
eng.build_op_plans()
for k, p in list(eng._plans.items())[:8]:
    print(k, "pipelined", p.pipelined_ok, "single_fma", p.single_fma_ok)
print("all fma:", all(p.single_fma_ok for p in eng._plans.values()), " fc2 fma:", all(p.single_fma_ok for k, p in eng._plans.items() if k.endswith("fc2")))
