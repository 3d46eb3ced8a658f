"""ivit_amd — MI355X-native integer-only ViT inference path (drop-in for the
operator surface of zkkli/I-ViT models/quantization_utils + models/layers_quant.py)."""
from .synth import ViTConfig, CONFIGS, make_vit_weights, make_images_int8, make_calibration_batch  # noqa
from . import freeze  # noqa
from . import _lib  # noqa
from ._lib import IvitError, build  # noqa


def __getattr__(name):
    if name == "ViTEngine":
        from .engine import ViTEngine
        return ViTEngine
    raise AttributeError(name)
