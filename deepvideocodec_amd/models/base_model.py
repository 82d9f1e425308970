"""Base class of the two context models (reference: dmc/models/base_model.py:43-112)."""
import math

import torch
import torch.nn as nn

from .entropy_models import EntropyBottleneck
from .utils import resize_registered_buffers

SCALES_MIN = 0.11
SCALES_MAX = 256
SCALES_LEVELS = 64


def get_scale_table(min=SCALES_MIN, max=SCALES_MAX, levels=SCALES_LEVELS):
    """64 log-spaced Gaussian scales (reference: base_model.py:43-49)."""
    return torch.exp(torch.linspace(math.log(min), math.log(max), levels))


class CompressionModel(nn.Module):
    """Owns one EntropyBottleneck; aux loss and CDF-table update over it."""

    def __init__(self, entropy_bottleneck_channels):
        super().__init__()
        self.entropy_bottleneck = EntropyBottleneck(entropy_bottleneck_channels)

    def aux_loss(self):
        return sum(m.loss() for m in self.modules() if isinstance(m, EntropyBottleneck))

    def update(self, force=False):
        updated = False
        for m in self.children():
            if isinstance(m, EntropyBottleneck):
                updated |= m.update(force=force)
        return updated

    def load_state_dict(self, state_dict, strict=True):
        resize_registered_buffers(self.entropy_bottleneck, "entropy_bottleneck",
                                  ["_quantized_cdf", "_offset", "_cdf_length"], state_dict)
        return super().load_state_dict(state_dict, strict)
