"""N1 diagnostic: calibrate each fixture's model here (operator surface, running_stat branch) and compare with the
reference-calibrated scales stored in the fixture: the forward-ordered list of QuantAct sites, the first site that differs,
and the float logits of the two calibrations on the fixture's images.   usage (GPU): python tools/calib_diag.py"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ivit_amd as iv
from conftest import load_golden, golden_scales
CALIB_BATCH = {"micro_vit_b2.npz": 4, "micro_vit2h_b3.npz": 4, "deit_tiny_b1.npz": 2, "micro_swin_b2.npz": 4}


def build(g):
    name = str(g["cfg_name"])
    if name in iv.SWIN_CONFIGS:
        from ivit_amd.swin_quant import SwinTransformer
        cfg = iv.SWIN_CONFIGS[name]
        m = SwinTransformer(img_size=cfg.img_size, patch_size=cfg.patch_size, in_chans=cfg.in_chans, num_classes=cfg.num_classes,
                            embed_dim=cfg.embed_dim, depths=cfg.depths, num_heads=cfg.num_heads, window_size=cfg.window_size,
                            mlp_ratio=cfg.mlp_ratio)
        m.load_float_weights(iv.make_swin_weights(cfg, int(g["seed"])))
    else:
        cfg = iv.CONFIGS[name]
        m = iv.VisionTransformer(img_size=cfg.img_size, patch_size=cfg.patch_size, num_classes=cfg.num_classes,
                                 embed_dim=cfg.embed_dim, depth=cfg.depth, num_heads=cfg.num_heads, mlp_ratio=4)
        m.load_float_weights(iv.make_vit_weights(cfg, int(g["seed"])))
    return cfg, m


for fname in sorted(CALIB_BATCH):
    g = load_golden(fname)
    cfg, m = build(g)
    order = []
    for n, mod in m.named_modules():
        if type(mod) is iv.QuantAct:
            mod.register_forward_hook(lambda mod, i, o, n=n: order.append(n) if n not in order else None)
    with torch.no_grad():
        m(torch.from_numpy(iv.make_calibration_batch(cfg, CALIB_BATCH[fname])).cuda())
    iv.freeze_model(m)
    ref = golden_scales(g)
    got = {k: np.float32(mod.act_scaling_factor.reshape(-1)[0].item()) for k, mod in m.named_modules() if type(mod) is iv.QuantAct}
    sites = [k for k in order if k in ref and ref[k] > 0]
    diff = [k for k in sites if got[k] != ref[k]]
    first = diff[0] if diff else None
    nb = sites.index(first) if first else len(sites)
    imgs = torch.from_numpy(iv.make_images_int8(cfg, int(g["batch"]), int(g["images_seed"]))).cuda()
    with torch.no_grad():
        acc, sc = m(imgs)
    here = acc.cpu().numpy().astype(np.float64) * np.asarray(sc, np.float64)
    there = g["logits_int"].astype(np.float64) * g["logits_scale"].astype(np.float64)
    rel = max(abs(got[k] - ref[k]) / ref[k] for k in sites)
    print(f"{fname}: {len(sites)} sites, {len(diff)} differ, first = {first!r} after {nb} equal sites; max rel scale diff {rel:.3e}; "
          f"argmax equal {bool((here.argmax(1) == there.argmax(1)).all())}; max |dlogit| {np.abs(here - there).max():.4e} "
          f"(logit range {np.abs(there).max():.3f}, head LSB {np.asarray(g['logits_scale']).max():.3e})")
    print("   first ten in forward order:", sites[:10])
    if first:
        i = sites.index(first)
        print("   around the first difference:", sites[max(0, i - 3):i + 2])
