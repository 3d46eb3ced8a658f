"""Reference checkpoint -> this build.

The reference's `state_dict()` is both its checkpoint (`quant_train.py:261`,
`torch.save(model.state_dict())`) and the wire format its integer runtime reads
(`TVM_benchmark/convert_model.py:12-148`; key schema SURVEY.md Appendix D).  This importer takes such a
dict (or a `checkpoint.pth.tar` holding it under 'state_dict' / 'model') and loads it into the
same-named modules of this package:

  * float parameters (`*.weight`, `*.bias`, `cls_token`, `pos_embed`,
    `relative_position_bias_table`) are copied;
  * `<site>.act_scaling_factor` of every QuantAct becomes that site's frozen scale.  The reference
    re-assigns these buffers in `forward` with other shapes than it registered (zeros(1) -> 0-dim;
    `norm_scaling_factor` zeros(1) -> [C]) — a post-forward state dict does not even load back into
    a fresh reference model — so scales are read by VALUE, whatever the shape;
  * derived integer buffers (`weight_integer`, `bias_integer`, `fc_scaling_factor`,
    `conv_scaling_factor`, `norm_scaling_factor`) are NOT trusted: they are recomputed from the float
    parameters and the scales by `ivit_amd.freeze`, and compared with the stored ones when
    `verify=True` (bit-equal for a checkpoint saved after a frozen forward).

Host-only; no GPU needed until the model runs.
"""
import numpy as np
import torch

from . import freeze as fz

_DERIVED = ("weight_integer", "bias_integer", "fc_scaling_factor", "conv_scaling_factor", "norm_scaling_factor",
            "relative_position_index", "attn_mask")


def _unwrap(obj, trusted=False):
    if isinstance(obj, (str, bytes)) or hasattr(obj, "read"):
        # the reference's checkpoint is a plain state_dict (quant_train.py:261): tensors only.  Full unpickling
        # (arbitrary code execution) only for a file the caller explicitly vouches for.
        obj = torch.load(obj, map_location="cpu", weights_only=not trusted)
    for key in ("state_dict", "model", "model_state_dict"):
        if isinstance(obj, dict) and key in obj and isinstance(obj[key], dict):
            obj = obj[key]
    if not isinstance(obj, dict):
        raise TypeError("expected a state dict or a checkpoint holding one under 'state_dict' / 'model'")
    out = {}
    for k, v in obj.items():
        k = k[7:] if k.startswith("module.") else k        # nn.DataParallel prefix
        out[k] = v.detach().cpu() if torch.is_tensor(v) else torch.as_tensor(np.asarray(v))
    return out


def split_state_dict(state_dict, trusted=False):
    """-> (float_params, act_scales, derived): numpy arrays keyed like the reference modules."""
    sd = _unwrap(state_dict, trusted)
    params, scales, derived = {}, {}, {}
    for k, v in sd.items():
        leaf = k.rsplit(".", 1)[-1]
        if leaf == "act_scaling_factor":
            flat = v.reshape(-1)
            if flat.numel() >= 1:
                scales[k[: -len(".act_scaling_factor")]] = np.float32(flat[0].item())
        elif leaf in _DERIVED:
            derived[k] = v.numpy()
        else:
            params[k] = v.float().numpy() if v.is_floating_point() else v.numpy()
    return params, scales, derived


def load_reference_state_dict(model, state_dict, verify=True, freeze=True, trusted=False):
    """Load a reference state dict into `model` (ivit_amd VisionTransformer / SwinTransformer).
    Returns the list of QuantAct sites that carry no usable scale in the checkpoint (never-called sites
    such as `act_out` / `attn.qact_softmax` hold 0 and are skipped)."""
    from .model_utils import freeze_model
    from .quant_modules import QuantAct, QuantLinear, QuantConv2d
    params, scales, derived = split_state_dict(state_dict, trusted)
    own = dict(model.named_parameters())
    unknown = [k for k in params if k not in own]
    if unknown:
        raise KeyError(f"checkpoint parameters without a home in this model: {unknown[:8]}")
    missing = [k for k in own if k not in params]
    if missing:
        raise KeyError(f"model parameters absent from the checkpoint: {missing[:8]}")
    with torch.no_grad():
        for k, p in own.items():
            v = torch.as_tensor(params[k])
            if tuple(v.shape) != tuple(p.shape):
                raise ValueError(f"{k}: checkpoint shape {tuple(v.shape)} != model {tuple(p.shape)}")
            p.copy_(v)
    mods = dict(model.named_modules())
    unset = []
    for name, mod in mods.items():
        if type(mod) is QuantAct:
            s = scales.get(name)
            if s is not None and s > 0:
                mod.set_scale(s)
            else:
                unset.append(name)
    if freeze:
        freeze_model(model)
    if verify:
        for name, mod in mods.items():
            if isinstance(mod, (QuantLinear, QuantConv2d)) and name + ".weight_integer" in derived:
                w_int, s_w = fz.quantize_weight(params[name + ".weight"])
                ref_w = derived[name + ".weight_integer"]
                if not np.array_equal(w_int.astype(np.float32).reshape(ref_w.shape), ref_w.astype(np.float32)):
                    raise ValueError(f"{name}: weight_integer in the checkpoint is not the quantisation of its float weight")
                key = name + (".conv_scaling_factor" if isinstance(mod, QuantConv2d) else ".fc_scaling_factor")
                if key in derived and derived[key].size == s_w.size and not np.array_equal(derived[key].reshape(-1), s_w):
                    raise ValueError(f"{key} disagrees with the float weight's per-channel range")
    return unset
