import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import ivit_amd as iv, ref_harness as rh
from oracle import oracle as orc
models = rh.load_reference()
def run(name, batch, calib):
    cfg = iv.SWIN_CONFIGS[name]
    w = iv.make_swin_weights(cfg, 0)
    m = rh.build_ref_swin(models, cfg, w)
    rng = np.random.Generator(np.random.PCG64(2))
    rh.calibrate_and_freeze(models, m, rng.standard_normal((calib,3,cfg.img_size,cfg.img_size)).astype(np.float32))
    with torch.no_grad(): m(torch.zeros(1,3,cfg.img_size,cfg.img_size))
    sc = rh.act_scales(models, m)
    q = np.random.Generator(np.random.PCG64(1)).integers(-128,128,(batch,3,cfg.img_size,cfg.img_size),dtype=np.int8)
    y, recs = rh.capture(models, m, q.astype(np.float32)*sc['qact_input'])
    o = orc.OracleSwin(cfg, w, sc)
    capd = {}
    logits, _ = o.forward(q, capd)
    bad = 0
    for r in recs:
        n = r['name']
        if n == 'qact_input': continue
        ref = r['acc'] if r['type'] in ('QuantLinear','QuantConv2d','QuantMatMul') else (r['z'] if r['type']=='IntLayerNorm' else r['out'])
        if n == 'patch_embed.proj': ref = ref.reshape(ref.shape[0], ref.shape[1], -1).transpose(0,2,1)
        got = capd.get(n)
        ref = np.asarray(ref).reshape(-1).astype(np.float64); got = np.asarray(got).reshape(-1).astype(np.float64)
        nb = int((ref != got).sum())
        if nb: print(n, r['type'], ref.size, 'mismatch', nb, np.abs(ref-got).max())
        bad += nb
    print(name, 'total mismatches', bad, 'logits equal', np.array_equal(recs[-1]['acc'], logits.astype(np.int64)))
for a in sys.argv[1:]:
    n, b, c = a.split(':'); run(n, int(b), int(c))
