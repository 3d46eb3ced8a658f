"""Swin Transformer on the integer operator surface — mirrors reference `models/swin_quant.py`
(window_partition/reverse :18-50, WindowAttention :53-169, SwinTransformerBlock :172-301,
PatchMerging :304-358, BasicLayer :361-416, SwinTransformer :419-564, factories :567-627).

Roll / window partition / patch-merge gathers are pure permutations of integer tensors (torch
indexing, as in the reference); every arithmetic operator is a C-ABI call.  Two reference quirks
are reproduced explicitly: the float shift mask is added before Shiftmax
(`IntSoftmax(mask=...)`), and stage-0 LayerNorms use torch's token-contiguous summation order
(`IntLayerNorm.sum_order = "token"`).
"""
from functools import partial

import numpy as np
import torch
import torch.nn as nn

from . import freeze as fz
from .layers_quant import PatchEmbed, Mlp, DropPath, to_2tuple
from .quant_modules import (QuantLinear, QuantAct, IntLayerNorm, IntSoftmax, IntGELU, QuantMatMul, _f32, _ptr,
                            _dyv, handle, _is_fake, to_fake)
from .synth import SwinConfig

__all__ = ["swin_tiny_patch4_window7_224", "swin_small_patch4_window7_224", "swin_base_patch4_window7_224",
           "SwinTransformer"]


def window_partition(x, window_size):
    B, H, W, C = x.shape
    x = x.view(B, H // window_size, window_size, W // window_size, window_size, C)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, window_size, window_size, C)


def window_reverse(windows, window_size, H, W):
    B = int(windows.shape[0] / (H * W / window_size / window_size))
    x = windows.view(B, H // window_size, W // window_size, window_size, window_size, -1)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(B, H, W, -1)


class WindowAttention(nn.Module):
    def __init__(self, dim, window_size, num_heads, qkv_bias=True, attn_drop=0.0, proj_drop=0.0):
        super().__init__()
        self.dim, self.window_size, self.num_heads = dim, window_size, num_heads
        head_dim = dim // num_heads
        self.scale = head_dim ** -0.5
        self.relative_position_bias_table = nn.Parameter(
            torch.zeros((2 * window_size[0] - 1) * (2 * window_size[1] - 1), num_heads))
        coords = torch.stack(torch.meshgrid([torch.arange(window_size[0]), torch.arange(window_size[1])],
                                            indexing="ij"))
        cf = torch.flatten(coords, 1)
        rel = (cf[:, :, None] - cf[:, None, :]).permute(1, 2, 0).contiguous()
        rel[:, :, 0] += window_size[0] - 1
        rel[:, :, 1] += window_size[1] - 1
        rel[:, :, 0] *= 2 * window_size[1] - 1
        self.register_buffer("relative_position_index", rel.sum(-1))
        self.qkv = QuantLinear(dim, dim * 3, bias=qkv_bias)
        self.qact1 = QuantAct()
        self.qact_attn1 = QuantAct()
        self.qact_table = QuantAct()
        self.qact2 = QuantAct()
        self.log_int_softmax = IntSoftmax()
        self.qact3 = QuantAct()
        self.qact4 = QuantAct(16)
        self.proj = QuantLinear(dim, dim)
        self.matmul_1 = QuantMatMul()
        self.matmul_2 = QuantMatMul()

    def forward(self, x, act_scaling_factor, mask=None):
        B_, N, C = x.shape
        x, s = self.qkv(x, act_scaling_factor)
        x, s1 = self.qact1(x, s)
        qkv = x.reshape(B_, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        attn, s = self.matmul_1(q, s1, k.transpose(-2, -1), s1)
        # attn * scale, scale * scale (swin_quant.py:133-134): integers unchanged, fp32 scale product; a fake-quant fp32
        # `attn` (the reference's convention) is scaled exactly like the reference scales it
        fake = _is_fake(attn)
        if fake:
            attn = attn * self.scale
        s_mm = s
        s = torch.from_numpy((_f32(s) * np.float32(self.scale)).astype(np.float32))
        if self.qact_attn1.running_stat and not fake:
            # calibration: the reference tracks the range of fl(fl(acc * s) * scale) — TWO roundings (swin_quant.py:132-133:
            # matmul_1 returns acc * s, then `attn * self.scale`), not fl(acc * fl(s * scale)); with head dim 32 the factor
            # 32^-0.5 is not a power of two and the two differ by an ulp on the tracked maximum
            Xr = (attn.float() * torch.as_tensor(_f32(s_mm), device=attn.device)) * self.scale
            self.qact_attn1._collect_range(Xr, None, None, None)
            self.qact_attn1.running_stat = False
            try:
                attn, s = self.qact_attn1(attn, s)
            finally:
                self.qact_attn1.running_stat = True
        else:
            attn, s = self.qact_attn1(attn, s)
        tab_q, s_tab = self.qact_table(self.relative_position_bias_table.detach().to(x.device))
        if fake and not _is_fake(tab_q):
            tab_q = to_fake(tab_q, s_tab)
        bias = tab_q[self.relative_position_index.view(-1).to(x.device)].view(N, N, -1).permute(2, 0, 1).contiguous()
        attn, s = self.qact2(attn, s, bias.unsqueeze(0), s_tab)
        if fake and mask is not None:
            # the reference's own statements (swin_quant.py:151-156): the float mask is ADDED to the fake-quant logits and
            # Shiftmax gets the sum — IntSoftmax recognises the masked entries and treats the mask as the side input it is
            nW = mask.shape[0]
            attn = attn.view(B_ // nW, nW, self.num_heads, N, N) + mask.to(attn.device).unsqueeze(1).unsqueeze(0)
            attn = attn.view(-1, self.num_heads, N, N)
            attn, s = self.log_int_softmax(attn, s)
        else:
            attn, s = self.log_int_softmax(attn, s, mask=mask, num_heads=self.num_heads)
        x, s = self.matmul_2(attn, s, v, s1)
        x = x.transpose(1, 2).reshape(B_, N, C)
        x, s = self.qact3(x, s)
        x, s = self.proj(x, s)
        x, s = self.qact4(x, s)
        return x, s


class SwinTransformerBlock(nn.Module):
    def __init__(self, dim, input_resolution, num_heads, window_size=7, shift_size=0, mlp_ratio=4.0, qkv_bias=True,
                 drop=0.0, attn_drop=0.0, drop_path=0.0, act_layer=IntGELU, norm_layer=IntLayerNorm):
        super().__init__()
        self.dim, self.input_resolution, self.num_heads = dim, input_resolution, num_heads
        self.window_size, self.shift_size, self.mlp_ratio = window_size, shift_size, mlp_ratio
        if min(self.input_resolution) <= self.window_size:
            self.shift_size = 0
            self.window_size = min(self.input_resolution)
        assert 0 <= self.shift_size < self.window_size, "shift_size must in 0-window_size"
        self.norm1 = norm_layer(dim)
        self.qact1 = QuantAct()
        self.attn = WindowAttention(dim, window_size=to_2tuple(self.window_size), num_heads=num_heads,
                                    qkv_bias=qkv_bias)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.qact2 = QuantAct(16)
        self.norm2 = norm_layer(dim)
        self.qact3 = QuantAct()
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)
        self.qact4 = QuantAct(16)
        if self.shift_size > 0:
            H, W = self.input_resolution
            img_mask = torch.zeros((1, H, W, 1))
            cnt = 0
            for h in (slice(0, -self.window_size), slice(-self.window_size, -self.shift_size),
                      slice(-self.shift_size, None)):
                for w in (slice(0, -self.window_size), slice(-self.window_size, -self.shift_size),
                          slice(-self.shift_size, None)):
                    img_mask[:, h, w, :] = cnt
                    cnt += 1
            mw = window_partition(img_mask, self.window_size).view(-1, self.window_size * self.window_size)
            am = mw.unsqueeze(1) - mw.unsqueeze(2)
            attn_mask = am.masked_fill(am != 0, float(-100.0)).masked_fill(am == 0, float(0.0))
        else:
            attn_mask = None
        self.register_buffer("attn_mask", attn_mask)

    def forward(self, x_1, s_1):
        H, W = self.input_resolution
        B, L, C = x_1.shape
        assert L == H * W, "input feature has wrong size"
        x, s = self.norm1(x_1, s_1)
        x, s = self.qact1(x, s)
        x = x.view(B, H, W, C)
        shifted = torch.roll(x, shifts=(-self.shift_size, -self.shift_size), dims=(1, 2)) if self.shift_size > 0 else x
        xw = window_partition(shifted, self.window_size).view(-1, self.window_size * self.window_size, C)
        aw, s = self.attn(xw, s, mask=self.attn_mask)
        aw = aw.view(-1, self.window_size, self.window_size, C)
        shifted = window_reverse(aw, self.window_size, H, W)
        x = torch.roll(shifted, shifts=(self.shift_size, self.shift_size), dims=(1, 2)) if self.shift_size > 0 else shifted
        x = self.drop_path(x.view(B, H * W, C))
        x_2, s_2 = self.qact2(x, s, x_1, s_1)
        x, s = self.norm2(x_2, s_2)
        x, s = self.qact3(x, s)
        x, s = self.mlp(x, s)
        x = self.drop_path(x)
        x, s = self.qact4(x, s, x_2, s_2)
        return x, s


class PatchMerging(nn.Module):
    def __init__(self, input_resolution, dim, norm_layer=IntLayerNorm):
        super().__init__()
        self.input_resolution, self.dim = input_resolution, dim
        self.norm = norm_layer(4 * dim)
        self.qact1 = QuantAct()
        self.reduction = QuantLinear(4 * dim, 2 * dim, bias=False)
        self.qact2 = QuantAct()

    def forward(self, x, s):
        H, W = self.input_resolution
        B, L, C = x.shape
        assert L == H * W, "input feature has wrong size"
        assert H % 2 == 0 and W % 2 == 0, f"x size ({H}*{W}) are not even."
        x = x.view(B, H, W, C)
        x = torch.cat([x[:, 0::2, 0::2, :], x[:, 1::2, 0::2, :], x[:, 0::2, 1::2, :], x[:, 1::2, 1::2, :]], -1)
        x = x.view(B, -1, 4 * C)
        x, s = self.norm(x, s)
        x, s = self.qact1(x, s)
        x, s = self.reduction(x, s)
        x, s = self.qact2(x, s)
        return x, s


class BasicLayer(nn.Module):
    def __init__(self, dim, input_resolution, depth, num_heads, window_size, mlp_ratio=4.0, qkv_bias=True, drop=0.0,
                 attn_drop=0.0, drop_path=0.0, norm_layer=IntLayerNorm, downsample=None, use_checkpoint=False):
        super().__init__()
        self.dim, self.input_resolution, self.depth = dim, input_resolution, depth
        self.blocks = nn.ModuleList([
            SwinTransformerBlock(dim=dim, input_resolution=input_resolution, num_heads=num_heads,
                                 window_size=window_size, shift_size=0 if (i % 2 == 0) else window_size // 2,
                                 mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, act_layer=IntGELU, norm_layer=norm_layer)
            for i in range(depth)])
        self.downsample = downsample(input_resolution, dim=dim, norm_layer=norm_layer) if downsample is not None else None

    def forward(self, x, s):
        for blk in self.blocks:
            x, s = blk(x, s)
        if self.downsample is not None:
            x, s = self.downsample(x, s)
        return x, s


class SwinTransformer(nn.Module):
    def __init__(self, img_size=224, patch_size=4, in_chans=3, num_classes=1000, embed_dim=96, depths=(2, 2, 6, 2),
                 num_heads=(3, 6, 12, 24), window_size=7, mlp_ratio=4.0, qkv_bias=True, drop_rate=0.0,
                 attn_drop_rate=0.0, drop_path_rate=0.1, norm_layer=IntLayerNorm, ape=False, patch_norm=True,
                 use_checkpoint=False, **kwargs):
        super().__init__()
        if ape:
            raise NotImplementedError("absolute position embedding is off in every reference factory")
        self.num_classes, self.num_layers, self.embed_dim = num_classes, len(depths), embed_dim
        self.patch_norm = patch_norm
        self.num_features = int(embed_dim * 2 ** (self.num_layers - 1))
        self.cfg = SwinConfig("custom", img_size, patch_size, in_chans, num_classes, embed_dim, tuple(depths),
                              tuple(num_heads), window_size, int(mlp_ratio))
        self.qact_input = QuantAct()
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=in_chans,
                                      embed_dim=embed_dim, norm_layer=norm_layer if patch_norm else None)
        self.patch_grid = self.patch_embed.grid_size
        self.qact1 = QuantAct(16)
        layers = []
        for i in range(self.num_layers):
            layers.append(BasicLayer(dim=int(embed_dim * 2 ** i),
                                     input_resolution=(self.patch_grid[0] // 2 ** i, self.patch_grid[1] // 2 ** i),
                                     depth=depths[i], num_heads=num_heads[i], window_size=window_size,
                                     mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, norm_layer=norm_layer,
                                     downsample=PatchMerging if i < self.num_layers - 1 else None))
        self.layers = nn.Sequential(*layers)
        self.norm = norm_layer(self.num_features)
        self.qact2 = QuantAct()
        self.qact3 = QuantAct()
        self.head = QuantLinear(self.num_features, num_classes) if num_classes > 0 else nn.Identity()
        self.act_out = QuantAct()
        self.fake_quant = False      # True: forward on the reference's fake-quant fp32 tensors, statement by statement
        # reference layout quirk: until the first PatchMerging (torch.cat) the fp32 activations keep the
        # token-contiguous layout of PatchEmbed's flatten(2).transpose(1,2), which changes torch's
        # summation order inside IntLayerNorm
        if patch_norm:
            self.patch_embed.norm.sum_order = "token"
        for blk in self.layers[0].blocks:
            blk.norm1.sum_order = "token"
            blk.norm2.sum_order = "token"

    def load_float_weights(self, weights):
        sd = {k: torch.as_tensor(np.asarray(v)) for k, v in weights.items()}
        missing, unexpected = self.load_state_dict(sd, strict=False)
        assert not unexpected, unexpected
        return self

    def load_act_scales(self, scales):
        mods = dict(self.named_modules())
        for k, v in scales.items():
            if k in mods and isinstance(mods[k], QuantAct):
                mods[k].set_scale(v)
        return self

    def _forward_features_fake(self, x):
        """the reference's forward_features (swin_quant.py:540-556) statement by statement, on fake-quant fp32 tensors"""
        if x.dtype == torch.int8:
            s = self.qact_input.act_scaling_factor
            x = to_fake(x, s)
        else:
            q, s = self.qact_input(x)
            x = q if q.is_floating_point() else to_fake(q, s)
        x, s = self.patch_embed(x, s)
        x, s = self.qact1(x, s)
        for layer in self.layers:
            x, s = layer(x, s)
        x, s = self.norm(x, s)
        x, s = self.qact2(x, s)
        x = torch.nn.functional.adaptive_avg_pool1d(x.transpose(1, 2), 1)      # B C 1, fp32 mean of fl(Q*s)
        x = torch.flatten(x, 1)
        x, s = self.qact3(x, s)
        return x, s

    def forward_features(self, x):
        if getattr(self, "fake_quant", False):
            return self._forward_features_fake(x)
        if x.dtype == torch.int8:
            s = self.qact_input.act_scaling_factor
        else:
            x, s = self.qact_input(x)
        x, s = self.patch_embed(x, s)
        x, s = self.qact1(x, s)
        for layer in self.layers:
            x, s = layer(x, s)
        x, s = self.norm(x, s)
        x, s = self.qact2(x, s)
        # avgpool over tokens + qact3 (swin_quant.py:553-555), fused: one C-ABI call
        B, L, C = x.shape
        if self.qact3.running_stat:
            # calibration: the range of the pooled fp32 activations, pooled with the reference's own op
            # (swin_quant.py:553-554) on the device
            X = x.float() * torch.as_tensor(_f32(s), device=x.device)
            pooled = torch.nn.functional.adaptive_avg_pool1d(X.transpose(1, 2), 1).flatten(1)
            self.qact3._collect_range(pooled, None, None, None)
        s3 = np.float32(self.qact3.act_scaling_factor.reshape(-1)[0].item())
        d = fz.dyadic(_f32(s), s3)
        out = torch.empty(B, C, dtype=torch.int8, device=x.device)
        xc = x.contiguous()
        handle(x.device).call("ivit_avgpool_requant", _ptr(xc), B, L, C, _dyv(d), _ptr(out))
        return out, self.qact3.act_scaling_factor

    def forward(self, x):
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


def _swin(pretrained=False, **kw):
    if pretrained:
        raise NotImplementedError("pretrained checkpoints need network access; load a state dict instead")
    kw.setdefault("norm_layer", partial(IntLayerNorm, eps=1e-6))
    return SwinTransformer(patch_size=4, window_size=7, **kw)


def swin_tiny_patch4_window7_224(pretrained=False, **kwargs):
    return _swin(pretrained, embed_dim=96, depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24), **kwargs)


def swin_small_patch4_window7_224(pretrained=False, **kwargs):
    return _swin(pretrained, embed_dim=96, depths=(2, 2, 18, 2), num_heads=(3, 6, 12, 24), **kwargs)


def swin_base_patch4_window7_224(pretrained=False, **kwargs):
    return _swin(pretrained, embed_dim=128, depths=(2, 2, 18, 2), num_heads=(4, 8, 16, 32), **kwargs)
