"""Architecture table and the motion head, mirroring reference models/module.py.

ENCODER_ARCH (reference models/module.py:17-32): the BasicBlock archs and the Bottleneck archs
(resnet50/101, wide_resnet50_2/101_2); EfficientNet students are out of scope.
FCNet (reference models/module.py:133-156) as built at train_vpd_model.py:61-65:
Linear(D,128)-ReLU-Linear(128,128)-Dropout(0)-ReLU-Linear(128,2D); state_dict
keys layers.{0,2,5}.{weight,bias}.
"""
import math

import torch
import torch.nn as nn

from ..engine import DECODER_PARAM_NAMES, ENCODER_LAYERS

ENCODER_ARCH = dict(ENCODER_LAYERS)


def attach_views(root, names, make_tensor, as_param=True):
    """Create nested holder modules so that state_dict() yields `names` in order."""
    for name in names:
        parts = name.split(".")
        mod = root
        for p in parts[:-1]:
            if p not in mod._modules:
                mod.add_module(p, nn.Module())
            mod = mod._modules[p]
        t = make_tensor(name)
        if as_param:
            mod.register_parameter(parts[-1], nn.Parameter(t, requires_grad=True))
        else:
            mod.register_buffer(parts[-1], t)


class FCNet(nn.Module):
    """Motion head D -> 128 -> 128 -> 2D.  Its tensors are views of the encoder
    engine's flat buffers (tail section), so the fused HIP train step and the
    fused AdamW see them; `forward` is the HIP head inside ModelTrainer.epoch."""

    def __init__(self, engine, input_dim, hidden_dims, output_dim, dropout=0):
        super().__init__()
        assert list(hidden_dims) == [128, 128] and output_dim == 2 * input_dim and dropout == 0, \
            "the HIP head implements FCNet(D,[128,128],2D,dropout=0) (train_vpd_model.py:61-65)"
        self._engine = [engine]      # list: keep the engine out of nn.Module registration
        attach_views(self, DECODER_PARAM_NAMES, lambda k: engine.view("decoder." + k))
        for k in DECODER_PARAM_NAMES:    # .grad aliases the flat gradient buffer
            self.get_parameter(k).grad = engine.view("decoder." + k, engine.grads)
        self.reset_parameters()

    def reset_parameters(self):
        # torch nn.Linear default: kaiming_uniform(a=sqrt(5)) == U(+-1/sqrt(fan_in)) for W and b
        with torch.no_grad():
            for idx in (0, 2, 5):
                w = self.get_parameter("layers.%d.weight" % idx)
                b = self.get_parameter("layers.%d.bias" % idx)
                bound = 1.0 / math.sqrt(w.shape[1])
                w.uniform_(-bound, bound)
                b.uniform_(-bound, bound)
