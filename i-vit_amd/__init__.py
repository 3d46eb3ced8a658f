"""ivit_amd — MI355X-native integer-only ViT inference path (drop-in for the
operator surface of zkkli/I-ViT models/quantization_utils + models/layers_quant.py)."""
from .synth import (ViTConfig, CONFIGS, make_vit_weights, make_images_int8, make_calibration_batch,  # noqa
                    SwinConfig, SWIN_CONFIGS, make_swin_weights)
from . import freeze  # noqa
from . import _lib  # noqa
from ._lib import IvitError, build  # noqa


_LAZY = {
    "ViTEngine": ("engine", "ViTEngine"),
    "QuantLinear": ("quant_modules", "QuantLinear"), "QuantAct": ("quant_modules", "QuantAct"),
    "QuantMatMul": ("quant_modules", "QuantMatMul"), "QuantConv2d": ("quant_modules", "QuantConv2d"),
    "IntLayerNorm": ("quant_modules", "IntLayerNorm"), "IntGELU": ("quant_modules", "IntGELU"),
    "IntSoftmax": ("quant_modules", "IntSoftmax"), "to_int": ("quant_modules", "to_int"),
    "Mlp": ("layers_quant", "Mlp"), "PatchEmbed": ("layers_quant", "PatchEmbed"),
    "VisionTransformer": ("vit_quant", "VisionTransformer"),
    "deit_tiny_patch16_224": ("vit_quant", "deit_tiny_patch16_224"),
    "deit_small_patch16_224": ("vit_quant", "deit_small_patch16_224"),
    "deit_base_patch16_224": ("vit_quant", "deit_base_patch16_224"),
    "vit_base_patch16_224": ("vit_quant", "vit_base_patch16_224"),
    "vit_large_patch16_224": ("vit_quant", "vit_large_patch16_224"),
    "freeze_model": ("model_utils", "freeze_model"), "unfreeze_model": ("model_utils", "unfreeze_model"),
    "SwinTransformer": ("swin_quant", "SwinTransformer"),
    "swin_tiny_patch4_window7_224": ("swin_quant", "swin_tiny_patch4_window7_224"),
    "swin_small_patch4_window7_224": ("swin_quant", "swin_small_patch4_window7_224"),
    "swin_base_patch4_window7_224": ("swin_quant", "swin_base_patch4_window7_224"),
    "SwinEngine": ("swin_engine", "SwinEngine"),
    "load_reference_state_dict": ("checkpoint", "load_reference_state_dict"),
}


def __getattr__(name):
    if name in _LAZY:
        import importlib
        mod, attr = _LAZY[name]
        return getattr(importlib.import_module("." + mod, __name__), attr)
    raise AttributeError(name)
