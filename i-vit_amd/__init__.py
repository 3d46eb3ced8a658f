from .synth import ViTConfig, CONFIGS, make_vit_weights, make_images_int8, make_calibration_batch  # noqa
