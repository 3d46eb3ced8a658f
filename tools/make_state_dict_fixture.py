"""Build-container tool: save what the REFERENCE's `model.state_dict()` looks like after a calibrated,
frozen forward (micro_vit) as a data fixture — key names, shapes, and the values of every buffer the
importer reads (scales, integer buffers).  Float parameters are the seeded synthetic ones and are not
stored (regenerated from the seed).  Output: tests/golden/micro_vit_state_dict.npz"""
import os, sys
import numpy as np, torch
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
import ivit_amd as iv
import ref_harness as rh

models = rh.load_reference()
SWIN = len(sys.argv) > 1 and sys.argv[1] == "swin"          # python tools/make_state_dict_fixture.py swin -> micro_swin
if SWIN:
    cfg = iv.SWIN_CONFIGS["micro_swin"]
    w = iv.make_swin_weights(cfg, 0)
    m = rh.build_ref_swin(models, cfg, w)
else:
    cfg = iv.CONFIGS["micro_vit"]
    w = iv.make_vit_weights(cfg, 0)
    m = rh.build_ref_vit(models, cfg, w)
rh.calibrate_and_freeze(models, m, iv.make_calibration_batch(cfg, 4))
with torch.no_grad():
    m(torch.from_numpy(iv.make_calibration_batch(cfg, 2, seed=3)))       # one frozen forward: buffers take their post-forward shapes
sd = m.state_dict()
out = {"seed": np.int64(0), "cfg_name": np.array(cfg.name)}
names, shapes = [], []
for k, v in sd.items():
    names.append(k); shapes.append(list(v.shape))
    if k not in w:                                   # buffers only (scales, integer tensors)
        out["buf/" + k] = v.detach().cpu().numpy()
out["keys"] = np.array(names)
out["shapes"] = np.array([",".join(map(str, s)) for s in shapes])
np.savez_compressed(os.path.join(os.path.dirname(HERE), "tests", "golden", f"{cfg.name}_state_dict.npz"), **out)
print(len(names), "keys;", sum(1 for k in out if k.startswith("buf/")), "buffers")
print([ (k, s) for k, s in zip(names, shapes) if "scaling_factor" in k][:6])
