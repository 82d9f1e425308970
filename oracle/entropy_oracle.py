"""CPU restatement of the two entropy models the reference takes from the
third-party package `compressai` (InterDigital, PyPI `compressai`; the
reference pins no version, its `CompressionModel` copy matches the >=1.2 API).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  **Parity unpinned**: the
`compressai` source is not available in this build environment and the
reference has no tests / golden vectors for it; this file restates the
published algorithm and is anchored on the reference's call sites:

  EntropyBottleneck   : dmc/models/base_model.py:63,71-78,96-102
                        dmc/models/video_model.py:220-224,238-239,257
  GaussianConditional : dmc/models/video_model.py:150,232,248-251,272-283
  scale table         : dmc/models/base_model.py:43-49

The classes are `nn.Module`s with the parameter / buffer names `compressai`
uses (`_matrix{i}`, `_bias{i}`, `_factor{i}`, `quantiles`, `target`,
`_offset`, `_quantized_cdf`, `_cdf_length`, `scale_table`) so that
(a) state dicts are interchangeable with the reference's checkpoints
    (dmc/models/video_model.py:626-656 lists the buffers), and
(b) tests/golden/make_golden.py can register this module as
    `compressai.entropy_models` and import the reference's own graph.
"""
import math
from typing import List, Optional

import numpy as np
import scipy.stats
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import rans_oracle

LIKELIHOOD_BOUND = 1e-9
SCALE_BOUND = 0.11
TAIL_MASS = 1e-9
CDF_PRECISION = 16


# --------------------------------------------------------------------------
# helpers
# --------------------------------------------------------------------------
class _LowerBoundFn(torch.autograd.Function):
    """max(x, bound) whose gradient also passes where it pushes x upward
    (compressai.ops.LowerBound)."""

    @staticmethod
    def forward(ctx, x, bound):
        ctx.save_for_backward(x, bound)
        return torch.max(x, bound)

    @staticmethod
    def backward(ctx, grad_output):
        x, bound = ctx.saved_tensors
        pass_through = (x >= bound) | (grad_output < 0)
        return pass_through.type(grad_output.dtype) * grad_output, None


def lower_bound(x: torch.Tensor, bound: float) -> torch.Tensor:
    return _LowerBoundFn.apply(x, torch.tensor([float(bound)], dtype=x.dtype, device=x.device))


def standardized_cumulative(x: torch.Tensor) -> torch.Tensor:
    """Phi(x) = 0.5 * erfc(-x / sqrt(2))."""
    return 0.5 * torch.erfc(-(2 ** -0.5) * x)


def pmf_to_quantized_cdf(pmf: List[float], precision: int = CDF_PRECISION) -> List[int]:
    """Integer CDF with total 2**precision and every symbol frequency >= 1
    (restates compressai's C++ `pmf_to_quantized_cdf`)."""
    cdf = [0] * (len(pmf) + 1)
    for i, p in enumerate(pmf):
        # C++ std::round: half away from zero (p >= 0 here)
        cdf[i + 1] = int(math.floor(p * (1 << precision) + 0.5))
    total = sum(cdf)
    if total == 0:
        raise ValueError("Invalid `pmf`: at least one element must have a non-zero probability.")
    for i in range(len(cdf)):
        cdf[i] = ((1 << precision) * cdf[i]) // total
    for i in range(1, len(cdf)):
        cdf[i] += cdf[i - 1]
    cdf[-1] = 1 << precision
    for i in range(len(cdf) - 1):
        if cdf[i] == cdf[i + 1]:
            # steal one count from the symbol with the smallest frequency > 1
            best_freq = (1 << 32) - 1
            best_steal = -1
            for j in range(len(cdf) - 1):
                freq = cdf[j + 1] - cdf[j]
                if 1 < freq < best_freq:
                    best_freq = freq
                    best_steal = j
            assert best_steal != -1
            if best_steal < i:
                for j in range(best_steal + 1, i + 1):
                    cdf[j] -= 1
            else:
                for j in range(i + 1, best_steal + 1):
                    cdf[j] += 1
    return cdf


def _pmf_to_cdf(pmf, tail_mass, pmf_length, max_length):
    cdf = torch.zeros((len(pmf_length), max_length + 2), dtype=torch.int32)
    for i, p in enumerate(pmf):
        prob = torch.cat((p[: pmf_length[i]], tail_mass[i]), dim=0)
        _cdf = pmf_to_quantized_cdf(prob.tolist(), CDF_PRECISION)
        cdf[i, : len(_cdf)] = torch.tensor(_cdf, dtype=torch.int32)
    return cdf


def quantize(inputs, mode, means=None):
    if mode == "noise":
        noise = torch.empty_like(inputs).uniform_(-0.5, 0.5)
        return inputs + noise
    outputs = inputs.clone()
    if means is not None:
        outputs = outputs - means
    outputs = torch.round(outputs)
    if mode == "dequantize":
        if means is not None:
            outputs = outputs + means
        return outputs
    assert mode == "symbols", mode
    return outputs.int()


class _EntropyModelBase(nn.Module):
    def __init__(self):
        super().__init__()
        self.register_buffer("_offset", torch.IntTensor())
        self.register_buffer("_quantized_cdf", torch.IntTensor())
        self.register_buffer("_cdf_length", torch.IntTensor())

    def _check_tables(self):
        if self._quantized_cdf.numel() == 0:
            raise ValueError("Uninitialized CDFs. Run update() first")

    def compress(self, inputs, indexes, means=None):
        symbols = quantize(inputs, "symbols", means)
        self._check_tables()
        cdf = self._quantized_cdf.tolist()
        cdf_len = self._cdf_length.reshape(-1).int().tolist()
        off = self._offset.reshape(-1).int().tolist()
        strings = []
        for i in range(symbols.size(0)):
            strings.append(rans_oracle.encode_with_indexes(
                symbols[i].reshape(-1).int().tolist(),
                indexes[i].reshape(-1).int().tolist(), cdf, cdf_len, off))
        return strings

    def decompress(self, strings, indexes, dtype=torch.float, means=None):
        self._check_tables()
        cdf = self._quantized_cdf.tolist()
        cdf_len = self._cdf_length.reshape(-1).int().tolist()
        off = self._offset.reshape(-1).int().tolist()
        outputs = torch.empty(indexes.size(), dtype=dtype)
        for i, s in enumerate(strings):
            values = rans_oracle.decode_with_indexes(
                s, indexes[i].reshape(-1).int().tolist(), cdf, cdf_len, off)
            outputs[i] = torch.tensor(values, dtype=dtype).reshape(outputs[i].size())
        if means is not None:
            outputs = outputs + means
        return outputs.to(indexes.device)


# --------------------------------------------------------------------------
# factorized prior
# --------------------------------------------------------------------------
class EntropyBottleneck(_EntropyModelBase):
    """Per-channel monotone cumulative 1-3-3-3-3-1 (Balle et al. 2018, app. 6.1)."""

    def __init__(self, channels, tail_mass=TAIL_MASS, init_scale=10.0, filters=(3, 3, 3, 3)):
        super().__init__()
        self.channels = int(channels)
        self.filters = tuple(int(f) for f in filters)
        self.init_scale = float(init_scale)
        self.tail_mass = float(tail_mass)

        filters = (1,) + self.filters + (1,)
        scale = self.init_scale ** (1 / (len(self.filters) + 1))
        channels = self.channels
        for i in range(len(self.filters) + 1):
            init = np.log(np.expm1(1 / scale / filters[i + 1]))
            matrix = torch.Tensor(channels, filters[i + 1], filters[i])
            matrix.data.fill_(init)
            self.register_parameter(f"_matrix{i:d}", nn.Parameter(matrix))
            bias = torch.Tensor(channels, filters[i + 1], 1)
            nn.init.uniform_(bias, -0.5, 0.5)
            self.register_parameter(f"_bias{i:d}", nn.Parameter(bias))
            if i < len(self.filters):
                factor = torch.Tensor(channels, filters[i + 1], 1)
                nn.init.zeros_(factor)
                self.register_parameter(f"_factor{i:d}", nn.Parameter(factor))

        self.quantiles = nn.Parameter(torch.Tensor(channels, 1, 3))
        init = torch.Tensor([-self.init_scale, 0, self.init_scale])
        self.quantiles.data = init.repeat(self.quantiles.size(0), 1, 1)
        target = np.log(2 / self.tail_mass - 1)
        self.register_buffer("target", torch.Tensor([-target, 0, target]))

    def _get_medians(self):
        return self.quantiles[:, :, 1:2].detach()

    def _logits_cumulative(self, inputs, stop_gradient=False):
        logits = inputs
        for i in range(len(self.filters) + 1):
            matrix = getattr(self, f"_matrix{i:d}")
            bias = getattr(self, f"_bias{i:d}")
            if stop_gradient:
                matrix, bias = matrix.detach(), bias.detach()
            logits = torch.matmul(F.softplus(matrix), logits) + bias
            if i < len(self.filters):
                factor = getattr(self, f"_factor{i:d}")
                if stop_gradient:
                    factor = factor.detach()
                logits = logits + torch.tanh(factor) * torch.tanh(logits)
        return logits

    def _likelihood(self, inputs):
        lower = self._logits_cumulative(inputs - 0.5)
        upper = self._logits_cumulative(inputs + 0.5)
        sign = -torch.sign(lower + upper).detach()
        return torch.abs(torch.sigmoid(sign * upper) - torch.sigmoid(sign * lower))

    def forward(self, x, training=None):
        if training is None:
            training = self.training
        # channels first, everything else flattened
        perm = [1, 0] + list(range(2, x.dim()))
        inv_perm = [1, 0] + list(range(2, x.dim()))
        x = x.permute(*perm).contiguous()
        shape = x.size()
        values = x.reshape(x.size(0), 1, -1)
        outputs = quantize(values, "noise" if training else "dequantize", self._get_medians())
        likelihood = self._likelihood(outputs)
        likelihood = lower_bound(likelihood, LIKELIHOOD_BOUND)
        outputs = outputs.reshape(shape).permute(*inv_perm).contiguous()
        likelihood = likelihood.reshape(shape).permute(*inv_perm).contiguous()
        return outputs, likelihood

    def loss(self):
        logits = self._logits_cumulative(self.quantiles, stop_gradient=True)
        return torch.abs(logits - self.target).sum()

    def update(self, force=False):
        if self._offset.numel() > 0 and not force:
            return False
        medians = self.quantiles[:, 0, 1]
        minima = torch.ceil(medians - self.quantiles[:, 0, 0]).int()
        minima = torch.clamp(minima, min=0)
        maxima = torch.ceil(self.quantiles[:, 0, 2] - medians).int()
        maxima = torch.clamp(maxima, min=0)
        self._offset = -minima
        pmf_start = medians - minima
        pmf_length = maxima + minima + 1
        max_length = int(pmf_length.max().item())
        samples = torch.arange(max_length, device=pmf_start.device)
        samples = samples[None, :] + pmf_start[:, None, None]
        lower = self._logits_cumulative(samples - 0.5, stop_gradient=True)
        upper = self._logits_cumulative(samples + 0.5, stop_gradient=True)
        sign = -torch.sign(lower + upper)
        pmf = torch.abs(torch.sigmoid(sign * upper) - torch.sigmoid(sign * lower))
        pmf = pmf[:, 0, :]
        tail_mass = torch.sigmoid(lower[:, 0, :1]) + torch.sigmoid(-upper[:, 0, -1:])
        self._quantized_cdf = _pmf_to_cdf(pmf.detach(), tail_mass.detach(), pmf_length, max_length)
        self._cdf_length = pmf_length + 2
        return True

    @staticmethod
    def _build_indexes(size):
        n, c = size[0], size[1]
        view = [1, c] + [1] * (len(size) - 2)
        return torch.arange(c).view(*view).int().repeat(n, 1, *size[2:])

    def compress(self, x):
        indexes = self._build_indexes(x.size())
        medians = self._get_medians().detach()
        medians = medians.reshape(1, -1, *([1] * (x.dim() - 2))).expand(x.size(0), *([-1] * (x.dim() - 1)))
        return super().compress(x.cpu(), indexes, medians.cpu())

    def decompress(self, strings, size):
        output_size = (len(strings), self._quantized_cdf.size(0), *size)
        indexes = self._build_indexes(output_size)
        medians = self._get_medians().detach()
        medians = medians.reshape(1, -1, *([1] * len(size))).expand(len(strings), *([-1] * (len(size) + 1)))
        return super().decompress(strings, indexes, medians.dtype, medians.cpu()).to(self.quantiles.device)


# --------------------------------------------------------------------------
# Gaussian conditional
# --------------------------------------------------------------------------
class GaussianConditional(_EntropyModelBase):
    def __init__(self, scale_table, scale_bound=SCALE_BOUND, tail_mass=TAIL_MASS):
        super().__init__()
        self.tail_mass = float(tail_mass)
        self.scale_bound = float(scale_bound)
        self.register_buffer(
            "scale_table",
            torch.Tensor(tuple(float(s) for s in scale_table)) if scale_table is not None else torch.Tensor())

    def _likelihood(self, inputs, scales, means=None):
        values = inputs if means is None else inputs - means
        scales = lower_bound(scales, self.scale_bound)
        values = torch.abs(values)
        upper = standardized_cumulative((0.5 - values) / scales)
        lower = standardized_cumulative((-0.5 - values) / scales)
        return upper - lower

    def forward(self, inputs, scales, means=None, training=None):
        if training is None:
            training = self.training
        outputs = quantize(inputs, "noise" if training else "dequantize", means)
        likelihood = self._likelihood(outputs, scales, means)
        likelihood = lower_bound(likelihood, LIKELIHOOD_BOUND)
        return outputs, likelihood

    def build_indexes(self, scales):
        scales = torch.max(scales, torch.tensor(self.scale_bound, dtype=scales.dtype, device=scales.device))
        indexes = scales.new_full(scales.size(), len(self.scale_table) - 1).int()
        for s in self.scale_table[:-1]:
            indexes -= (scales <= s).int()
        return indexes

    def update_scale_table(self, scale_table, force=False):
        if self._offset.numel() > 0 and not force:
            return False
        device = self.scale_table.device
        self.scale_table = torch.Tensor(tuple(float(s) for s in scale_table)).to(device)
        self.update()
        return True

    def update(self):
        multiplier = -scipy.stats.norm.ppf(self.tail_mass / 2)
        pmf_center = torch.ceil(self.scale_table * multiplier).int()
        pmf_length = 2 * pmf_center + 1
        max_length = int(torch.max(pmf_length).item())
        device = pmf_center.device
        samples = torch.abs(torch.arange(max_length, device=device).int() - pmf_center[:, None])
        samples_scale = self.scale_table.unsqueeze(1)
        samples = samples.float()
        samples_scale = samples_scale.float()
        upper = standardized_cumulative((0.5 - samples) / samples_scale)
        lower = standardized_cumulative((-0.5 - samples) / samples_scale)
        pmf = upper - lower
        tail_mass = 2 * lower[:, :1]
        self._quantized_cdf = _pmf_to_cdf(pmf, tail_mass, pmf_length, max_length)
        self._offset = -pmf_center
        self._cdf_length = pmf_length + 2

    def compress(self, inputs, indexes, means=None):
        return super().compress(inputs.cpu(), indexes.cpu(), None if means is None else means.cpu())

    def decompress(self, strings, indexes, dtype=torch.float, means=None):
        return super().decompress(strings, indexes.cpu(), dtype, None if means is None else means.cpu()).to(indexes.device)
