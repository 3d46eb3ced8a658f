"""Reference harness — runs ONLY in the build container (needs /root/reference).

Imports the reference I-ViT `models` package on CPU with the two shims of
SURVEY.md Appendix B (tkinter stub, Tensor.cuda() -> identity), loads this
repo's seeded synthetic weights into it, calibrates once, freezes, and captures
integer-domain inputs/outputs of every quantized operator with forward hooks.

Nothing from the reference is copied: this file only *calls* it.  The outputs
(integer tensors, scales) become the data fixtures under tests/golden/.
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_HERE))


def load_reference():
    if not os.path.isdir(REF):
        raise RuntimeError("reference tree not present (this tool only runs in the build container)")
    if "tkinter" not in sys.modules:
        tk = types.ModuleType("tkinter")
        tk.X = None
        sys.modules["tkinter"] = tk
    torch.Tensor.cuda = lambda self, *a, **k: self
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import models  # noqa: the reference package
    return models


def build_ref_vit(models, cfg, weights):
    from functools import partial
    m = models.vit_quant.VisionTransformer(
        img_size=cfg.img_size, patch_size=cfg.patch_size, in_chans=cfg.in_chans,
        num_classes=cfg.num_classes, embed_dim=cfg.embed_dim, depth=cfg.depth,
        num_heads=cfg.num_heads, mlp_ratio=cfg.mlp_ratio, qkv_bias=True,
        norm_layer=partial(models.IntLayerNorm, eps=1e-6))
    sd = {k: torch.from_numpy(v.copy()) for k, v in weights.items()}
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    # only buffers (weight_integer, *_scaling_factor, bias_integer) may be missing
    for k in missing:
        assert ("integer" in k) or ("scaling_factor" in k), k
    m.eval()
    return m


def build_ref_swin(models, cfg, weights):
    from functools import partial
    m = models.swin_quant.SwinTransformer(
        img_size=cfg.img_size, patch_size=cfg.patch_size, in_chans=cfg.in_chans, num_classes=cfg.num_classes,
        embed_dim=cfg.embed_dim, depths=list(cfg.depths), num_heads=list(cfg.num_heads),
        window_size=cfg.window_size, mlp_ratio=cfg.mlp_ratio, norm_layer=partial(models.IntLayerNorm, eps=1e-6))
    sd = {k: torch.from_numpy(v.copy()) for k, v in weights.items()}
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    for k in missing:
        assert ("integer" in k) or ("scaling_factor" in k) or ("relative_position_index" in k) or ("attn_mask" in k), k
    m.eval()
    return m


def calibrate_and_freeze(models, m, calib_fp32):
    with torch.no_grad():
        m(torch.from_numpy(calib_fp32))
        models.freeze_model(m)


def act_scales(models, m):
    """fp32 activation scales of every QuantAct (after a frozen forward)."""
    out = {}
    for name, mod in m.named_modules():
        if type(mod) is models.QuantAct:
            out[name] = np.float32(mod.act_scaling_factor.detach().reshape(-1)[0].item())
    return out


def _np(t):
    return t.detach().cpu().numpy()


def _bshape(s, x):
    """broadcast a scale over the channel (= last) dim of x (conv scale is [1,C,1,1])."""
    s = s.detach()
    if s.dim() == 4:
        return s
    return s.reshape(-1) if s.numel() > 1 else s.reshape(())


def capture(models, m, x_fp32):
    """Frozen forward with hooks. Returns (logits fp32, list of per-op records)."""
    recs = []
    hooks = []
    kinds = (models.QuantLinear, models.QuantAct, models.QuantMatMul, models.QuantConv2d,
             models.IntLayerNorm, models.IntGELU, models.IntSoftmax)

    def mk(name):
        def hook(mod, inp, out):
            r = {"name": name, "type": type(mod).__name__}
            y, s_out = out
            if isinstance(mod, models.QuantAct):
                x = inp[0]
                s_out = mod.act_scaling_factor
                r["s_out"] = np.float32(s_out.reshape(-1)[0].item())
                r["bits"] = mod.activation_bit
                r["out"] = _np(torch.round(y / s_out)).astype(np.int32)
                if len(inp) > 1 and inp[1] is not None:
                    s_pre = inp[1]
                    r["s_pre"] = _np(s_pre.reshape(-1)).astype(np.float32)
                    # exactly the z_int the reference computes (quant_utils.py:220)
                    sp = s_pre.reshape(-1) if s_pre.numel() > 1 else s_pre.reshape(())
                    if x.dim() == 4 and s_pre.numel() > 1:
                        sp = s_pre.reshape(1, -1, 1, 1)
                    r["z"] = _np(torch.round(x / sp)).astype(np.float32)
                    if len(inp) > 2 and inp[2] is not None:
                        idt, s_id = inp[2], inp[3]
                        si = s_id.reshape(())
                        r["s_id"] = _np(s_id.reshape(-1)).astype(np.float32)
                        r["z_id"] = _np(torch.round(idt / si)).astype(np.float32)
                else:
                    r["x_fp32"] = _np(x).astype(np.float32)
            elif isinstance(mod, (models.QuantLinear,)):
                x, s_in = inp
                r["s_in"] = np.float32(s_in.reshape(-1)[0].item())
                r["x"] = _np(torch.round(x / s_in)).astype(np.int32)
                r["s_out"] = _np(s_out.reshape(-1)).astype(np.float32)
                r["acc"] = _np(torch.round(y / s_out.reshape(1, -1) if y.dim() == 2
                                           else y / s_out.reshape(1, 1, -1))).astype(np.int64)
                r["w_int"] = _np(mod.weight_integer).astype(np.int32)
                if mod.bias_integer is not None:
                    r["b_int"] = _np(mod.bias_integer).astype(np.int64)
                r["s_w"] = _np(mod.fc_scaling_factor).astype(np.float32)
                r["y_fp32"] = _np(y).astype(np.float32)
            elif isinstance(mod, models.QuantConv2d):
                x, s_in = inp
                r["s_in"] = np.float32(s_in.reshape(-1)[0].item())
                r["x"] = _np(torch.round(x / s_in)).astype(np.int32)
                r["s_out"] = _np(s_out.reshape(-1)).astype(np.float32)
                r["acc"] = _np(torch.round(y / s_out)).astype(np.int64)
                r["w_int"] = _np(mod.weight_integer).astype(np.int32)
                r["b_int"] = _np(mod.bias_integer).astype(np.int64)
                r["s_w"] = _np(mod.conv_scaling_factor).astype(np.float32)
            elif isinstance(mod, models.QuantMatMul):
                A, sA, B, sB = inp
                r["sA"] = np.float32(sA.reshape(-1)[0].item())
                r["sB"] = np.float32(sB.reshape(-1)[0].item())
                r["A"] = _np(torch.round(A / sA)).astype(np.int32)
                r["B"] = _np(torch.round(B / sB)).astype(np.int32)
                r["s_out"] = np.float32(s_out.reshape(-1)[0].item())
                r["acc"] = _np(torch.round(y / s_out)).astype(np.int64)
            elif isinstance(mod, models.IntLayerNorm):
                x, s_in = inp
                r["s_in"] = np.float32(s_in.reshape(-1)[0].item())
                r["x"] = _np(torch.round(x / s_in)).astype(np.int32)
                r["s_out"] = _np(s_out.reshape(-1)).astype(np.float32)
                r["ln_w"] = _np(mod.weight).astype(np.float32)
                r["ln_b"] = _np(mod.bias).astype(np.float32)
                # z the next QuantAct will see: round(fl(fl(out*sc)/sc))
                r["z"] = _np(torch.round(y / s_out.reshape(1, 1, -1))).astype(np.float32)
                r["bias_integer"] = _np(mod.bias_integer).astype(np.float32)
            elif isinstance(mod, models.IntGELU):
                x, s_in = inp
                r["s_in"] = np.float32(s_in.reshape(-1)[0].item())
                r["x"] = _np(torch.round(x / s_in)).astype(np.int32)
                r["s_out"] = np.float32(s_out.reshape(-1)[0].item())
                r["out"] = _np(torch.round(y / s_out)).astype(np.int32)
            elif isinstance(mod, models.IntSoftmax):
                x, s_in = inp
                r["s_in"] = np.float32(s_in.reshape(-1)[0].item())
                r["x"] = _np(torch.round(x / s_in)).astype(np.int32)
                r["x_raw"] = _np(x).astype(np.float32)
                r["s_out"] = np.float32(s_out.reshape(-1)[0].item())
                r["bits"] = mod.output_bit
                r["out"] = _np(torch.round(y / s_out)).astype(np.int32)
            recs.append(r)
        return hook

    for name, mod in m.named_modules():
        if isinstance(mod, kinds):
            hooks.append(mod.register_forward_hook(mk(name)))
    with torch.no_grad():
        y = m(torch.from_numpy(x_fp32))
    for h in hooks:
        h.remove()
    return _np(y), recs
