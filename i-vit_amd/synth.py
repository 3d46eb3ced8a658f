"""Seeded synthetic weights / inputs for the integer-only ViT path.

There is no network for checkpoints or datasets, so every benchmark and parity
test uses weights drawn here (numpy PCG64, bit-reproducible across machines).
The dict keys follow the reference state-dict schema (SURVEY.md Appendix D;
reference models/vit_quant.py:176-240) so the same arrays can be loaded into the
reference model when golden vectors are generated (tools/make_golden.py).
"""
from dataclasses import dataclass, asdict
import numpy as np


@dataclass(frozen=True)
class ViTConfig:
    name: str
    img_size: int = 224
    patch_size: int = 16
    in_chans: int = 3
    num_classes: int = 1000
    embed_dim: int = 384
    depth: int = 12
    num_heads: int = 6
    mlp_ratio: int = 4

    @property
    def grid(self):
        return self.img_size // self.patch_size

    @property
    def num_patches(self):
        return self.grid * self.grid

    @property
    def num_tokens(self):
        return self.num_patches + 1

    @property
    def head_dim(self):
        return self.embed_dim // self.num_heads

    @property
    def hidden_dim(self):
        return int(self.embed_dim * self.mlp_ratio)

    def to_dict(self):
        return asdict(self)


# factories of reference models/vit_quant.py:285-381 (+ micro config for fixtures)
CONFIGS = {
    "micro_vit": ViTConfig("micro_vit", img_size=32, patch_size=8, num_classes=10,
                           embed_dim=64, depth=2, num_heads=1),
    "micro_vit2h": ViTConfig("micro_vit2h", img_size=32, patch_size=8, num_classes=10,
                             embed_dim=128, depth=2, num_heads=2),
    "deit_tiny": ViTConfig("deit_tiny", embed_dim=192, depth=12, num_heads=3),
    "deit_small": ViTConfig("deit_small", embed_dim=384, depth=12, num_heads=6),
    "deit_base": ViTConfig("deit_base", embed_dim=768, depth=12, num_heads=12),
    "vit_base": ViTConfig("vit_base", embed_dim=768, depth=12, num_heads=12),
    "vit_base_384": ViTConfig("vit_base_384", img_size=384, embed_dim=768, depth=12,
                              num_heads=12),
    "vit_large": ViTConfig("vit_large", embed_dim=1024, depth=24, num_heads=16),
}


def _tn(rng, shape, std, gain=1.0):
    w = rng.standard_normal(shape, dtype=np.float64) * std
    w = np.clip(w, -2.0 * std, 2.0 * std) * gain
    return w.astype(np.float32)


def make_vit_weights(cfg: ViTConfig, seed: int = 0):
    """Float32 parameters of a ViT/DeiT, keyed like the reference state dict."""
    rng = np.random.Generator(np.random.PCG64(seed))
    D, Hd = cfg.embed_dim, cfg.hidden_dim
    P = cfg.patch_size
    w = {}
    w["cls_token"] = _tn(rng, (1, 1, D), 0.02)
    w["pos_embed"] = _tn(rng, (1, cfg.num_tokens, D), 0.02)
    w["patch_embed.proj.weight"] = _tn(rng, (D, cfg.in_chans, P, P), 0.02, 2.0)
    w["patch_embed.proj.bias"] = (rng.standard_normal(D) * 0.3).astype(np.float32)

    def ln(prefix):
        w[prefix + ".weight"] = (1.0 + rng.standard_normal(D) * 0.4).astype(np.float32)
        w[prefix + ".bias"] = (rng.standard_normal(D) * 0.5).astype(np.float32)

    def lin(prefix, out_f, in_f, gain):
        w[prefix + ".weight"] = _tn(rng, (out_f, in_f), 0.02, gain)
        w[prefix + ".bias"] = (rng.standard_normal(out_f) * 0.3).astype(np.float32)

    for i in range(cfg.depth):
        b = f"blocks.{i}."
        ln(b + "norm1")
        lin(b + "attn.qkv", 3 * D, D, 6.0)
        lin(b + "attn.proj", D, D, 2.0)
        ln(b + "norm2")
        lin(b + "mlp.fc1", Hd, D, 2.0)
        lin(b + "mlp.fc2", D, Hd, 2.0)
    ln("norm")
    lin("head", cfg.num_classes, D, 2.0)
    return w


def make_images_int8(cfg: ViTConfig, batch: int, seed: int = 1):
    """Synthetic int8 NCHW image batch, i.i.d. uniform in [-128, 127]."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.integers(-128, 128, size=(batch, cfg.in_chans, cfg.img_size, cfg.img_size),
                        dtype=np.int8)


def make_calibration_batch(cfg: ViTConfig, batch: int, seed: int = 2):
    """Seeded fp32 batch used for the single calibration forward (values in ~N(0,1))."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.standard_normal((batch, cfg.in_chans, cfg.img_size, cfg.img_size)).astype(np.float32)


# ---------------------------------------------------------------------------
# Swin (reference models/swin_quant.py:419-627)
@dataclass(frozen=True)
class SwinConfig:
    name: str
    img_size: int = 224
    patch_size: int = 4
    in_chans: int = 3
    num_classes: int = 1000
    embed_dim: int = 96
    depths: tuple = (2, 2, 6, 2)
    num_heads: tuple = (3, 6, 12, 24)
    window_size: int = 7
    mlp_ratio: int = 4

    @property
    def grid(self):
        return self.img_size // self.patch_size

    @property
    def num_layers(self):
        return len(self.depths)

    @property
    def num_features(self):
        return self.embed_dim * 2 ** (self.num_layers - 1)

    def to_dict(self):
        return asdict(self)


SWIN_CONFIGS = {
    "micro_swin": SwinConfig("micro_swin", img_size=56, num_classes=10, embed_dim=32, depths=(2, 2),
                             num_heads=(1, 2)),
    "swin_tiny": SwinConfig("swin_tiny"),
    "swin_small": SwinConfig("swin_small", depths=(2, 2, 18, 2)),
    "swin_base": SwinConfig("swin_base", embed_dim=128, depths=(2, 2, 18, 2), num_heads=(4, 8, 16, 32)),
}


def make_swin_weights(cfg: SwinConfig, seed: int = 0):
    """Float32 parameters of a Swin model keyed like the reference state dict."""
    rng = np.random.Generator(np.random.PCG64(seed))
    w = {}
    P, E = cfg.patch_size, cfg.embed_dim

    def ln(prefix, C):
        w[prefix + ".weight"] = (1.0 + rng.standard_normal(C) * 0.4).astype(np.float32)
        w[prefix + ".bias"] = (rng.standard_normal(C) * 0.5).astype(np.float32)

    def lin(prefix, out_f, in_f, gain, bias=True):
        w[prefix + ".weight"] = _tn(rng, (out_f, in_f), 0.02, gain)
        if bias:
            w[prefix + ".bias"] = (rng.standard_normal(out_f) * 0.3).astype(np.float32)

    w["patch_embed.proj.weight"] = _tn(rng, (E, cfg.in_chans, P, P), 0.02, 6.0)
    w["patch_embed.proj.bias"] = (rng.standard_normal(E) * 0.3).astype(np.float32)
    ln("patch_embed.norm", E)
    ws = cfg.window_size
    for i, (depth, heads) in enumerate(zip(cfg.depths, cfg.num_heads)):
        C = E * 2 ** i
        res = cfg.grid // 2 ** i
        for j in range(depth):
            b = f"layers.{i}.blocks.{j}."
            ln(b + "norm1", C)
            wsz = min(ws, res)
            w[b + "attn.relative_position_bias_table"] = _tn(rng, ((2 * wsz - 1) ** 2, heads), 0.02, 20.0)
            lin(b + "attn.qkv", 3 * C, C, 6.0)
            lin(b + "attn.proj", C, C, 2.0)
            ln(b + "norm2", C)
            lin(b + "mlp.fc1", C * cfg.mlp_ratio, C, 2.0)
            lin(b + "mlp.fc2", C, C * cfg.mlp_ratio, 2.0)
        if i < cfg.num_layers - 1:
            ln(f"layers.{i}.downsample.norm", 4 * C)
            lin(f"layers.{i}.downsample.reduction", 2 * C, 4 * C, 2.0, bias=False)
    ln("norm", cfg.num_features)
    lin("head", cfg.num_classes, cfg.num_features, 2.0)
    return w
