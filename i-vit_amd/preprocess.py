"""Device side of the reference's eval transform (utils/data_utils.py:82-92) after the PIL resize + centre
crop: ToTensor -> Normalize(IMAGENET_DEFAULT_MEAN/STD) -> the model's input QuantAct, in one kernel from
uint8 HWC pixels.  The host ships 150 KB per 224x224 image instead of 602 KB of fp32."""
import ctypes

import numpy as np
import torch

from .quant_modules import handle

IMAGENET_DEFAULT_MEAN = (0.485, 0.456, 0.406)
IMAGENET_DEFAULT_STD = (0.229, 0.224, 0.225)


def normalize_quantize(u8_hwc, scale, mean=IMAGENET_DEFAULT_MEAN, std=IMAGENET_DEFAULT_STD):
    """u8_hwc: uint8 device tensor [B, H, W, 3]; scale: the input QuantAct's scale (qact_input).
    Returns int8 [B, 3, H, W] — what `model(...)` / `engine.forward(...)` take."""
    if u8_hwc.dtype != torch.uint8 or u8_hwc.dim() != 4 or u8_hwc.shape[-1] != 3:
        raise TypeError("expected a uint8 tensor [B, H, W, 3]")
    x = u8_hwc.contiguous()
    B, H, W, _ = x.shape
    out = torch.empty(B, 3, H, W, dtype=torch.int8, device=x.device)
    m = (ctypes.c_float * 3)(*[float(np.float32(v)) for v in mean])
    s = (ctypes.c_float * 3)(*[float(np.float32(v)) for v in std])
    handle(x.device).call("ivit_normalize_quantize_u8", ctypes.c_void_p(x.data_ptr()), B, H, W, m, s,
                          float(np.float32(scale)), ctypes.c_void_p(out.data_ptr()))
    return out


def resize_center_crop(u8_hwc, size=256, crop=224):
    """Resize(size, bicubic) + CenterCrop(crop) on the device (utils/data_utils.py:82-88; the reference uses
    size = int(crop / 0.875)): uint8 [B, H0, W0, 3] -> uint8 [B, crop, crop, 3]."""
    if u8_hwc.dtype != torch.uint8 or u8_hwc.dim() != 4 or u8_hwc.shape[-1] != 3:
        raise TypeError("expected a uint8 tensor [B, H, W, 3]")
    x = u8_hwc.contiguous()
    B, H0, W0, _ = x.shape
    ws = torch.empty(B * H0 * crop * 3, dtype=torch.float32, device=x.device)
    out = torch.empty(B, crop, crop, 3, dtype=torch.uint8, device=x.device)
    handle(x.device).call("ivit_resize_center_crop_u8", ctypes.c_void_p(x.data_ptr()), B, H0, W0, int(size), int(crop),
                          ctypes.c_void_p(ws.data_ptr()), ctypes.c_void_p(out.data_ptr()))
    return out


def eval_transform(u8_hwc, scale, size=256, crop=224):
    """the whole eval transform of the reference on the device: resize -> centre crop -> ToTensor -> Normalize -> input QuantAct"""
    return normalize_quantize(resize_center_crop(u8_hwc, size, crop), scale)
