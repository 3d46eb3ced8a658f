"""reference `models/model_utils.py:5-40`: freeze / unfreeze = toggle QuantAct.running_stat (and, here, IntGELU's calibration
side output: the reference's fp32 value for the range statistics of the QuantAct behind it, quant_modules.IntGELU.forward)."""
import torch.nn as nn

from .quant_modules import QuantAct, IntGELU


def freeze_model(model):
    """fix the activation ranges (inference mode): recursively QuantAct.fix()"""
    for m in model.modules() if isinstance(model, nn.Module) else []:
        if type(m) in (QuantAct, IntGELU):
            m.fix()


def unfreeze_model(model):
    for m in model.modules() if isinstance(model, nn.Module) else []:
        if type(m) in (QuantAct, IntGELU):
            m.unfix()
