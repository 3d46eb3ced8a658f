"""Composite layers of the reference `models/layers_quant.py` (Mlp :116-153, PatchEmbed
:156-196, DropPath :105-113) on top of the integer operator surface."""
import torch
import torch.nn as nn

from .quant_modules import QuantLinear, QuantAct, QuantConv2d, IntGELU


def to_2tuple(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


class DropPath(nn.Module):
    """identity at inference (reference layers_quant.py:105-113)"""

    def __init__(self, drop_prob=None):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        if self.training and self.drop_prob:
            raise NotImplementedError("stochastic depth is a training feature; inference only")
        return x


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=IntGELU, drop=0.0):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = QuantLinear(in_features, hidden_features)
        self.act = act_layer()
        self.qact1 = QuantAct()
        self.fc2 = QuantLinear(hidden_features, out_features)
        self.qact2 = QuantAct(16)
        self.qact_gelu = QuantAct()

    def forward(self, x, act_scaling_factor):
        x, s = self.fc1(x, act_scaling_factor)
        x, s = self.qact_gelu(x, s)
        x, s = self.act(x, s)
        x, s = self.qact1(x, s)
        x, s = self.fc2(x, s)
        x, s = self.qact2(x, s)
        return x, s


class PatchEmbed(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, norm_layer=None):
        super().__init__()
        img_size, patch_size = to_2tuple(img_size), to_2tuple(patch_size)
        self.img_size, self.patch_size = img_size, patch_size
        self.grid_size = (img_size[0] // patch_size[0], img_size[1] // patch_size[1])
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.norm_layer = norm_layer
        self.proj = QuantConv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        if self.norm_layer:
            self.qact_before_norm = QuantAct()
            self.norm = norm_layer(embed_dim)
        self.qact = QuantAct(16)

    def forward(self, x, act_scaling_factor):
        B, C, H, W = x.shape
        assert H == self.img_size[0] and W == self.img_size[1], \
            f"Input image size ({H}*{W}) doesn't match model ({self.img_size[0]}*{self.img_size[1]})."
        x, s = self.proj(x, act_scaling_factor)
        x = x.flatten(2).transpose(1, 2)
        s = s.reshape(-1)
        if self.norm_layer:
            x, s = self.qact_before_norm(x, s)
            x, s = self.norm(x if x.is_floating_point() else x.to(torch.int16), s)     # fake-quant fp32 stays as it is
        x, s = self.qact(x, s)
        return x, s
