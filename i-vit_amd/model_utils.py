"""reference `models/model_utils.py:5-40`: freeze / unfreeze = toggle QuantAct.running_stat."""
import torch.nn as nn

from .quant_modules import QuantAct


def freeze_model(model):
    """fix the activation ranges (inference mode): recursively QuantAct.fix()"""
    for m in model.modules() if isinstance(model, nn.Module) else []:
        if type(m) is QuantAct:
            m.fix()


def unfreeze_model(model):
    for m in model.modules() if isinstance(model, nn.Module) else []:
        if type(m) is QuantAct:
            m.unfix()
