"""DeiT / ViT on the integer operator surface — mirrors reference `models/vit_quant.py`
(Attention :22-88, Block :91-143, VisionTransformer :146-282, factories :285-381).

`forward` chains the per-operator modules exactly like the reference (one C-ABI call per
operator); `VisionTransformer.compile()` returns the fused `ViTEngine` (same integers, the
path bench.py measures)."""
from functools import partial

import numpy as np
import torch
import torch.nn as nn

from .layers_quant import PatchEmbed, Mlp, DropPath
from .quant_modules import (QuantLinear, QuantAct, IntLayerNorm, IntSoftmax, IntGELU, QuantMatMul, _f32, _is_fake, to_fake)
from .synth import ViTConfig

__all__ = ["deit_tiny_patch16_224", "deit_small_patch16_224", "deit_base_patch16_224",
           "vit_base_patch16_224", "vit_large_patch16_224", "VisionTransformer"]


class Attention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0.0, proj_drop=0.0):
        super().__init__()
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.scale = qk_scale or head_dim ** -0.5
        self.qkv = QuantLinear(dim, dim * 3, bias=qkv_bias)
        self.qact1 = QuantAct()
        self.qact_attn1 = QuantAct()
        self.qact2 = QuantAct()
        self.proj = QuantLinear(dim, dim)
        self.qact3 = QuantAct(16)
        self.qact_softmax = QuantAct()
        self.int_softmax = IntSoftmax(16)
        self.matmul_1 = QuantMatMul()
        self.matmul_2 = QuantMatMul()

    def forward(self, x, act_scaling_factor):
        B, N, C = x.shape
        x, s = self.qkv(x, act_scaling_factor)
        x, s1 = self.qact1(x, s)
        qkv = x.reshape(B, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        attn, s = self.matmul_1(q, s1, k.transpose(-2, -1), s1)
        # attn * scale and scale * scale (vit_quant.py:72-73): the integers are unchanged; a fake-quant fp32 `attn`
        # (reference convention) is scaled exactly like the reference scales it
        if _is_fake(attn):
            attn = attn * self.scale
        s = torch.from_numpy((_f32(s) * np.float32(self.scale)).astype(np.float32))
        attn, s = self.qact_attn1(attn, s)
        attn, s = self.int_softmax(attn, s)
        x, s = self.matmul_2(attn, s, v, s1)
        x = x.transpose(1, 2).reshape(B, N, C)
        x, s = self.qact2(x, s)
        x, s = self.proj(x, s)
        x, s = self.qact3(x, s)
        return x, s


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False, qk_scale=None, drop=0.0, attn_drop=0.0,
                 drop_path=0.0, act_layer=IntGELU, norm_layer=IntLayerNorm):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.qact1 = QuantAct()
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.qact2 = QuantAct(16)
        self.norm2 = norm_layer(dim)
        self.qact3 = QuantAct()
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)
        self.qact4 = QuantAct(16)

    def forward(self, x_1, s_1):
        x, s = self.norm1(x_1, s_1)
        x, s = self.qact1(x, s)
        x, s = self.attn(x, s)
        x = self.drop_path(x)
        x_2, s_2 = self.qact2(x, s, x_1, s_1)
        x, s = self.norm2(x_2, s_2)
        x, s = self.qact3(x, s)
        x, s = self.mlp(x, s)
        x = self.drop_path(x)
        x, s = self.qact4(x, s, x_2, s_2)
        return x, s


class VisionTransformer(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12,
                 num_heads=12, mlp_ratio=4.0, qkv_bias=True, qk_scale=None, representation_size=None,
                 drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.0, norm_layer=None):
        super().__init__()
        if representation_size:
            raise NotImplementedError("pre_logits representation layer is not on the integer path")
        self.num_classes = num_classes
        self.num_features = self.embed_dim = embed_dim
        norm_layer = norm_layer or IntLayerNorm
        self.cfg = ViTConfig("custom", img_size, patch_size, in_chans, num_classes, embed_dim, depth, num_heads,
                             int(mlp_ratio))
        self.qact_input = QuantAct()
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=in_chans,
                                      embed_dim=embed_dim)
        num_patches = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + 1, embed_dim))
        self.qact_pos = QuantAct(16)
        self.qact1 = QuantAct(16)
        self.blocks = nn.ModuleList([
            Block(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale,
                  act_layer=IntGELU, norm_layer=norm_layer) for _ in range(depth)])
        self.norm = norm_layer(embed_dim)
        self.qact2 = QuantAct()
        self.head = QuantLinear(self.num_features, num_classes) if num_classes > 0 else nn.Identity()
        self.act_out = QuantAct()
        # False (default): integer tensors between the operators.  True: the reference's fake-quant fp32 tensors
        # X = fl(Q*s) travel between them (same integers inside every kernel) and forward returns fp32 logits.
        self.fake_quant = False

    # ---- frozen-model plumbing -------------------------------------------------------
    def load_float_weights(self, weights):
        """numpy/tensor dict keyed like the reference state dict (fp32 parameters)."""
        sd = {k: torch.as_tensor(np.asarray(v)) for k, v in weights.items()}
        missing, unexpected = self.load_state_dict(sd, strict=False)
        assert not unexpected, unexpected
        return self

    def load_act_scales(self, scales):
        """{QuantAct module name -> act_scaling_factor} from a calibrated reference model."""
        mods = dict(self.named_modules())
        for k, v in scales.items():
            if k in mods and isinstance(mods[k], QuantAct):
                mods[k].set_scale(v)
        return self

    def act_scales(self):
        return {n: np.float32(m.act_scaling_factor.reshape(-1)[0].item()) for n, m in self.named_modules()
                if isinstance(m, QuantAct) and float(m.act_scaling_factor.reshape(-1)[0]) > 0}

    def compile(self, device="cuda:0"):
        """fused engine over the same frozen integers (ivit_amd.engine.ViTEngine)"""
        from .engine import ViTEngine
        w = {k: v.detach().cpu().numpy() for k, v in self.state_dict().items()
             if not ("integer" in k or "scaling_factor" in k)}
        return ViTEngine.from_float(self.cfg, w, self.act_scales(), device=device)

    # ---- reference forward (vit_quant.py:254-282) ------------------------------------
    def _forward_features_fake(self, x):
        """the reference's forward_features (vit_quant.py:254-276) statement by statement, on fake-quant fp32 tensors"""
        B = x.shape[0]
        if x.dtype == torch.int8:
            s = self.qact_input.act_scaling_factor
            x = to_fake(x, s)
        else:
            # per call, not module state: a later fake_quant = False forward must get integers from qact_input again
            q, s = self.qact_input(x)
            x = q if q.is_floating_point() else to_fake(q, s)
        x, s = self.patch_embed(x, s)
        cls_tokens = self.cls_token.to(x.device).expand(B, -1, -1)     # a float: rounded inside qact1 like in the reference
        x = torch.cat((cls_tokens, x), dim=1)
        pos_int = self.qact_pos.quantize_param(self.pos_embed[0]).astype(np.float32)
        x_pos = to_fake(torch.from_numpy(pos_int).to(x.device).unsqueeze(0), self.qact_pos.act_scaling_factor)
        x, s = self.qact1(x, s, x_pos, self.qact_pos.act_scaling_factor)
        for blk in self.blocks:
            x, s = blk(x, s)
        x, s = self.norm(x, s)
        x = x[:, 0]
        x, s = self.qact2(x.contiguous(), s)
        return x, s

    def forward_features(self, x):
        if self.fake_quant:
            return self._forward_features_fake(x)
        B = x.shape[0]
        if x.dtype == torch.int8:      # already-quantised image batch
            s = self.qact_input.act_scaling_factor
        else:
            x, s = self.qact_input(x)
        x, s = self.patch_embed(x, s)
        # the float class token rides the patch scale: rne(cls / s) (vit_quant.py:259-265)
        s_pe = _f32(s)[0]
        cls = np.rint((self.cls_token.detach().cpu().numpy().reshape(-1).astype(np.float32) / s_pe)
                      .astype(np.float32)).astype(np.int32)
        cls_t = torch.from_numpy(cls).to(x.device).reshape(1, 1, -1).expand(B, -1, -1)
        z = torch.cat((cls_t, x.to(torch.int32)), dim=1)
        # position embedding: a parameter, quantised on the host once (qact_pos, 16 bit)
        pos = self.qact_pos.quantize_param(self.pos_embed[0]).astype(np.int32)
        x_pos = torch.from_numpy(pos).to(x.device).unsqueeze(0)
        x, s = self.qact1(z, s, x_pos, self.qact_pos.act_scaling_factor)
        for blk in self.blocks:
            x, s = blk(x, s)
        x, s = self.norm(x, s)
        x = x[:, 0]
        x, s = self.qact2(x.contiguous(), s)
        return x, s

    def forward(self, x):
        """returns (int32 head accumulators [B, classes], per-class fp32 scale); the reference
        returns fp32 acc*scale (vit_quant.py:278-282) = `logits_fp32(acc, scale)`."""
        x, s = self.forward_features(x)
        x, s = self.head(x, s)
        return x, s

    def int8_logits(self, acc, scale):
        """The 8-bit output requant the reference defines but never runs (`act_out`, vit_quant.py:240 and the
        commented-out call at :281; same in swin_quant.py): QuantAct(8) on the head's int32 accumulators with
        their per-class scale.  While `act_out.running_stat` it calibrates its range from these logits like
        any other site; once frozen it is one per-channel dyadic requant.  Not pinned by the reference (it has
        no scale for this site) — a build-side extra.  -> (int8 [B, classes], scale)"""
        return self.act_out(acc, scale)

    @staticmethod
    def logits_fp32(acc, scale):
        return acc.float() * torch.as_tensor(_f32(scale), device=acc.device)


def _vit(embed_dim, depth, num_heads, pretrained=False, **kwargs):
    if pretrained:
        raise NotImplementedError("pretrained checkpoints need network access; load a state dict instead")
    return VisionTransformer(patch_size=16, embed_dim=embed_dim, depth=depth, num_heads=num_heads, mlp_ratio=4,
                             qkv_bias=True, norm_layer=partial(IntLayerNorm, eps=1e-6), **kwargs)


def deit_tiny_patch16_224(pretrained=False, **kwargs):
    return _vit(192, 12, 3, pretrained, **kwargs)


def deit_small_patch16_224(pretrained=False, **kwargs):
    return _vit(384, 12, 6, pretrained, **kwargs)


def deit_base_patch16_224(pretrained=False, **kwargs):
    return _vit(768, 12, 12, pretrained, **kwargs)


def vit_base_patch16_224(pretrained=False, **kwargs):
    return _vit(768, 12, 12, pretrained, **kwargs)


def vit_large_patch16_224(pretrained=False, **kwargs):
    return _vit(1024, 24, 16, pretrained, **kwargs)
