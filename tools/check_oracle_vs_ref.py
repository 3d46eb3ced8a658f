"""Build-container check: oracle vs the imported reference, every op site."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import ivit_amd as iv
import ref_harness as rh
from oracle import oracle as orc

def run(cfg_name, batch, calib_batch=4, seed=0, verbose=True):
    models = rh.load_reference()
    cfg = iv.CONFIGS[cfg_name]
    w = iv.make_vit_weights(cfg, seed)
    m = rh.build_ref_vit(models, cfg, w)
    rh.calibrate_and_freeze(models, m, iv.make_calibration_batch(cfg, calib_batch))
    with torch.no_grad():
        m(torch.zeros(1, 3, cfg.img_size, cfg.img_size))
    sc = rh.act_scales(models, m)
    q = iv.make_images_int8(cfg, batch)
    x = q.astype(np.float32) * sc['qact_input']
    t = time.time(); y, recs = rh.capture(models, m, x); t_ref = time.time() - t
    o = orc.OracleViT(cfg, w, sc)
    capd = {}
    t = time.time(); logits, s_head = o.forward(q, capd); t_orc = time.time() - t
    bad = 0
    for r in recs:
        n = r['name']
        if n in ('qact_input', 'qact_pos'):
            continue
        if r['type'] in ('QuantLinear', 'QuantConv2d', 'QuantMatMul'):
            ref = r['acc']
        elif r['type'] == 'IntLayerNorm':
            ref = r['z']
        else:
            ref = r['out']
        got = capd[n]
        if n == 'patch_embed.proj':
            ref = ref.reshape(ref.shape[0], ref.shape[1], -1).transpose(0, 2, 1)
        if n == 'norm':
            ref = ref[:, 0]
        if n == 'qact2':
            pass
        ref = np.asarray(ref).reshape(-1).astype(np.float64)
        got = np.asarray(got).reshape(-1).astype(np.float64)
        nbad = int((ref != got).sum())
        if nbad or verbose:
            print(f"{n:32s} {r['type']:14s} n={ref.size:9d} mismatches={nbad} maxabs={np.abs(ref-got).max() if nbad else 0}")
        bad += nbad
    ref_logits_int = recs[-1]['acc']
    print('logits equal', np.array_equal(ref_logits_int, logits.astype(np.int64)), 'argmax', y.argmax(1)[:8], logits.argmax(1)[:8])
    print(f'total mismatches {bad}; ref {t_ref:.2f}s oracle {t_orc:.2f}s')
    return bad

if __name__ == '__main__':
    run(sys.argv[1] if len(sys.argv) > 1 else 'micro_vit', int(sys.argv[2]) if len(sys.argv) > 2 else 2,
        verbose=(len(sys.argv) <= 3))
