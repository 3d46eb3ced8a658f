"""State-dict importer against the reference's own post-forward state dict
(tests/golden/micro_vit_state_dict.npz, written by tools/make_state_dict_fixture.py)."""
import io

import numpy as np
import pytest
import torch

import ivit_amd as iv
from ivit_amd import checkpoint as ck
from conftest import load_golden, golden_scales


def _reference_state_dict():
    f = load_golden("micro_vit_state_dict.npz")
    cfg = iv.CONFIGS[str(f["cfg_name"])]
    w = iv.make_vit_weights(cfg, int(f["seed"]))
    sd = {}
    for k, shp in zip(f["keys"], f["shapes"]):
        k = str(k)
        v = w[k] if k in w else f["buf/" + k]
        t = torch.from_numpy(np.asarray(v).copy())
        want = tuple(int(x) for x in str(shp).split(",") if x != "")
        assert tuple(t.shape) == want, (k, t.shape, want)
        sd[k] = t
    return cfg, w, sd


def _fresh(cfg):
    return iv.VisionTransformer(img_size=cfg.img_size, patch_size=cfg.patch_size, num_classes=cfg.num_classes,
                                embed_dim=cfg.embed_dim, depth=cfg.depth, num_heads=cfg.num_heads, mlp_ratio=4)


def test_import_reference_state_dict_scales_and_weights():
    cfg, w, sd = _reference_state_dict()
    # the shape quirk the importer exists for: post-forward scales are 0-dim, LN scales are [C]
    assert sd["qact1.act_scaling_factor"].dim() == 0
    assert sd["blocks.0.norm1.norm_scaling_factor"].shape == (cfg.embed_dim,)
    m = _fresh(cfg)
    unset = ck.load_reference_state_dict(m, sd)
    ref = golden_scales(load_golden("micro_vit_b2.npz"))
    got = m.act_scales()
    for k, v in ref.items():
        if v > 0:
            assert got[k] == v, k
    # only the sites the reference never calls carry no scale
    assert all(u.endswith("attn.qact_softmax") or u == "act_out" for u in unset), unset
    for k, p in m.named_parameters():
        assert np.array_equal(p.detach().numpy(), w[k]), k
    assert all(not mod.running_stat for mod in m.modules() if type(mod) is iv.QuantAct)


def test_import_accepts_checkpoint_wrappers(tmp_path):
    cfg, w, sd = _reference_state_dict()
    wrapped = {"epoch": 3, "state_dict": {"module." + k: v for k, v in sd.items()}}
    path = tmp_path / "checkpoint.pth.tar"
    torch.save(wrapped, path)
    m = _fresh(cfg)
    ck.load_reference_state_dict(m, str(path))
    assert m.act_scales()["qact1"] == np.float32(sd["qact1.act_scaling_factor"].item())
    buf = io.BytesIO()
    torch.save(sd, buf)
    buf.seek(0)
    params, scales, derived = ck.split_state_dict(buf)
    assert "blocks.0.attn.qkv.weight_integer" in derived and "blocks.0.attn.qkv.weight" in params
    assert scales["blocks.0.attn.qact1"] > 0


def test_import_rejects_inconsistent_checkpoints():
    cfg, w, sd = _reference_state_dict()
    bad = dict(sd)
    bad["blocks.0.attn.qkv.weight_integer"] = sd["blocks.0.attn.qkv.weight_integer"] + 1
    with pytest.raises(ValueError, match="weight_integer"):
        ck.load_reference_state_dict(_fresh(cfg), bad)
    short = {k: v for k, v in sd.items() if k != "blocks.1.mlp.fc2.weight"}
    with pytest.raises(KeyError, match="absent"):
        ck.load_reference_state_dict(_fresh(cfg), short)
    extra = dict(sd)
    extra["blocks.7.norm1.weight"] = torch.zeros(cfg.embed_dim)
    with pytest.raises(KeyError, match="without a home"):
        ck.load_reference_state_dict(_fresh(cfg), extra)
    wrong = dict(sd)
    wrong["head.weight"] = torch.zeros(3, 3)
    with pytest.raises(ValueError, match="shape"):
        ck.load_reference_state_dict(_fresh(cfg), wrong)
