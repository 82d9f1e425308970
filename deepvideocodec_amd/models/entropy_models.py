"""Entropy models of the DMC path on MI355X.

These two `nn.Module`s take the place of `compressai.entropy_models`
(`EntropyBottleneck`, `GaussianConditional`; used by the reference at
dmc/models/base_model.py:63 and dmc/models/video_model.py:150,220-251,257-283)
with the same parameter / buffer names and method surface.  Likelihoods,
quantisation and index search run as HIP kernels (csrc/entropy.hip); the range
coder and the CDF-table quantiser are the host C++ entry points of the same
shared library.  CDF tables are built on the host from the parameters -- a
setup step (`update()`), not part of the per-frame path.
"""
import ctypes as C
import math
from typing import List, Optional

import numpy as np
import scipy.stats
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib, ops

CDF_PRECISION = 16


def _quantized_cdf_table(pmf: torch.Tensor, tail_mass: torch.Tensor, pmf_length: torch.Tensor,
                         max_length: int) -> torch.Tensor:
    """Rows of integer CDFs (total 2^16, every frequency >= 1), zero padded to max_length + 2."""
    lib = _lib.load()
    pmf = pmf.detach().cpu().float()
    tail_mass = tail_mass.detach().cpu().float()
    cdf = np.zeros((len(pmf_length), max_length + 2), dtype=np.int32)
    for i in range(len(pmf_length)):
        n = int(pmf_length[i])
        prob = np.ascontiguousarray(np.concatenate([pmf[i, :n].numpy(), tail_mass[i].numpy().reshape(-1)]),
                                    dtype=np.float32)
        row = np.zeros(n + 2, dtype=np.int32)
        _lib.check(lib.dvc_pmf_to_quantized_cdf(prob.ctypes.data, prob.size, CDF_PRECISION, row.ctypes.data),
                   "dvc_pmf_to_quantized_cdf")
        cdf[i, :n + 2] = row
    return torch.from_numpy(cdf)


_noise_hook = None   # tests: callable(shape, device) -> tensor used instead of the random draw


def uniform_noise(shape, device) -> torch.Tensor:
    """U(-1/2, 1/2) quantisation noise of the training path (compressai's "noise" mode)."""
    if _noise_hook is not None:
        return _noise_hook(tuple(shape), device)
    return torch.empty(tuple(shape), dtype=torch.float32, device=device).uniform_(-0.5, 0.5)


class _CoderTables:
    """Contiguous host copies of the CDF tables for the range coder."""

    def __init__(self, cdf: torch.Tensor, cdf_length: torch.Tensor, offset: torch.Tensor):
        self.cdf = np.ascontiguousarray(cdf.detach().cpu().numpy().astype(np.int32))
        self.sizes = np.ascontiguousarray(cdf_length.detach().cpu().numpy().astype(np.int32).reshape(-1))
        self.offsets = np.ascontiguousarray(offset.detach().cpu().numpy().astype(np.int32).reshape(-1))
        self.n, self.stride = self.cdf.shape

    def encode(self, symbols: np.ndarray, indexes: np.ndarray) -> bytes:
        symbols = np.ascontiguousarray(symbols, dtype=np.int32).reshape(-1)
        indexes = np.ascontiguousarray(indexes, dtype=np.int32).reshape(-1)
        if symbols.size != indexes.size:
            raise ValueError("encode: symbols / indexes size mismatch")
        cap = 8 + 4 * (symbols.size + 16) + 64
        lib = _lib.load()
        while True:
            out = np.empty(cap, dtype=np.uint8)
            rc = lib.dvc_rans_encode(symbols.ctypes.data, indexes.ctypes.data, symbols.size, self.cdf.ctypes.data,
                                     self.n, self.stride, self.sizes.ctypes.data, self.offsets.ctypes.data,
                                     out.ctypes.data, cap)
            if rc == -3:  # DVC_ENOSPC: many escape-coded symbols
                cap *= 4
                continue
            _lib.check(rc, "dvc_rans_encode")
            return out[:rc].tobytes()

    def decode(self, data: bytes, indexes: np.ndarray) -> np.ndarray:
        indexes = np.ascontiguousarray(indexes, dtype=np.int32).reshape(-1)
        out = np.empty(indexes.size, dtype=np.int32)
        buf = np.frombuffer(data, dtype=np.uint8)
        _lib.check(_lib.load().dvc_rans_decode(buf.ctypes.data, buf.size, indexes.ctypes.data, indexes.size,
                                               self.cdf.ctypes.data, self.n, self.stride, self.sizes.ctypes.data,
                                               self.offsets.ctypes.data, out.ctypes.data), "dvc_rans_decode")
        return out


class EntropyModel(nn.Module):
    def __init__(self):
        super().__init__()
        self.register_buffer("_offset", torch.IntTensor())
        self.register_buffer("_quantized_cdf", torch.IntTensor())
        self.register_buffer("_cdf_length", torch.IntTensor())
        self._tables: Optional[_CoderTables] = None

    def coder_tables(self) -> _CoderTables:
        if self._quantized_cdf.numel() == 0:
            raise ValueError("Uninitialized CDFs. Run update() first")
        if self._tables is None:
            self._tables = _CoderTables(self._quantized_cdf, self._cdf_length, self._offset)
        return self._tables

    def _invalidate_tables(self):
        self._tables = None

    def _load_from_state_dict(self, *args, **kwargs):
        super()._load_from_state_dict(*args, **kwargs)
        self._invalidate_tables()


class EntropyBottleneck(EntropyModel):
    """Factorized prior: per-channel cumulative 1-3-3-3-3-1 network."""

    def __init__(self, channels: int, tail_mass: float = 1e-9, init_scale: float = 10, filters=(3, 3, 3, 3)):
        super().__init__()
        if tuple(filters) != (3, 3, 3, 3):
            raise ValueError("the HIP factorized-prior kernel is specialised for filters=(3,3,3,3)")
        self.channels = int(channels)
        self.filters = tuple(int(f) for f in filters)
        self.init_scale = float(init_scale)
        self.tail_mass = float(tail_mass)
        filters = (1,) + self.filters + (1,)
        scale = self.init_scale ** (1 / (len(self.filters) + 1))
        for i in range(len(self.filters) + 1):
            init = np.log(np.expm1(1 / scale / filters[i + 1]))
            self.register_parameter(f"_matrix{i:d}", nn.Parameter(
                torch.full((channels, filters[i + 1], filters[i]), float(init))))
            self.register_parameter(f"_bias{i:d}", nn.Parameter(
                torch.empty(channels, filters[i + 1], 1).uniform_(-0.5, 0.5)))
            if i < len(self.filters):
                self.register_parameter(f"_factor{i:d}", nn.Parameter(torch.zeros(channels, filters[i + 1], 1)))
        self.quantiles = nn.Parameter(
            torch.tensor([-self.init_scale, 0.0, self.init_scale]).repeat(channels, 1, 1))
        target = np.log(2 / self.tail_mass - 1)
        self.register_buffer("target", torch.Tensor([-target, 0, target]))

    # -- parameters for the kernel: [C][58] raw values (softplus/tanh are applied on the GPU)
    def packed_parameters(self, differentiable: bool = False) -> torch.Tensor:
        """(C, 58) raw parameters in the kernels' order: per layer matrix, bias, factor.  With
        `differentiable` the concat stays on the autograd tape, so a gradient w.r.t. the packed
        tensor reaches _matrix*/_bias*/_factor*."""
        parts = []
        for i in range(5):
            names = (f"_matrix{i}", f"_bias{i}") + ((f"_factor{i}",) if i < 4 else ())
            for nm in names:
                p = getattr(self, nm)
                parts.append((p if differentiable else p.detach()).reshape(self.channels, -1))
        packed = torch.cat(parts, dim=1).contiguous().float()
        assert packed.shape[1] == _lib.EB_PARAMS_PER_CHANNEL
        return packed

    def _get_medians(self) -> torch.Tensor:
        return self.quantiles[:, :, 1:2].detach()

    def _medians_flat(self) -> torch.Tensor:
        return self.quantiles[:, 0, 1].detach().contiguous().float()

    def forward(self, x: torch.Tensor, training: Optional[bool] = None):
        """-> (outputs, likelihoods); outputs = x + U(-.5,.5) in training, else round(x-med)+med."""
        if training is None:
            training = self.training
        noise = None
        if training:
            noise = uniform_noise(x.shape, x.device)
        if torch.is_grad_enabled():
            from .. import autograd as A
            # evaluation mode under autograd: the likelihood is taken at round(x - median) + median, so no
            # gradient reaches x through it (round), but the density parameters still get theirs
            return A.eb_likelihood(x if noise is not None else x.detach(), self.packed_parameters(differentiable=True),
                                   self._medians_flat(), noise)
        out, lik, _ = ops.eb_forward(x, self.packed_parameters(), self._medians_flat(), noise=noise)
        return out, lik

    def quantize_symbols(self, x: torch.Tensor):
        """-> (z_hat, int32 symbols) without the likelihood (compress path)."""
        z_hat, _, sym = ops.eb_forward(x, self.packed_parameters(), self._medians_flat(),
                                       want_likelihood=False, want_symbols=True)
        return z_hat, sym

    # -- host-side parameter math (setup / aux loss; tiny: C x 3 values)
    def _logits_cumulative(self, inputs, stop_gradient=False):
        logits = inputs
        for i in range(5):
            matrix = getattr(self, f"_matrix{i}")
            bias = getattr(self, f"_bias{i}")
            if stop_gradient:
                matrix, bias = matrix.detach(), bias.detach()
            logits = torch.matmul(F.softplus(matrix), logits) + bias
            if i < 4:
                factor = getattr(self, f"_factor{i}")
                if stop_gradient:
                    factor = factor.detach()
                logits = logits + torch.tanh(factor) * torch.tanh(logits)
        return logits

    def loss(self) -> torch.Tensor:
        logits = self._logits_cumulative(self.quantiles, stop_gradient=True)
        return torch.abs(logits - self.target).sum()

    @torch.no_grad()
    def update(self, force: bool = False) -> bool:
        if self._offset.numel() > 0 and not force:
            return False
        dev = self.quantiles.device
        q = self.quantiles.detach().cpu()
        medians = q[:, 0, 1]
        minima = torch.clamp(torch.ceil(medians - q[:, 0, 0]).int(), min=0)
        maxima = torch.clamp(torch.ceil(q[:, 0, 2] - medians).int(), min=0)
        pmf_start = medians - minima
        pmf_length = maxima + minima + 1
        max_length = int(pmf_length.max().item())
        samples = torch.arange(max_length)[None, :] + pmf_start[:, None, None]
        cpu_self = {n: p.detach().cpu() for n, p in self.named_parameters()}

        def logits_cum(x):
            for i in range(5):
                x = torch.matmul(F.softplus(cpu_self[f"_matrix{i}"]), x) + cpu_self[f"_bias{i}"]
                if i < 4:
                    x = x + torch.tanh(cpu_self[f"_factor{i}"]) * torch.tanh(x)
            return x

        lower = logits_cum(samples - 0.5)
        upper = logits_cum(samples + 0.5)
        sign = -torch.sign(lower + upper)
        pmf = torch.abs(torch.sigmoid(sign * upper) - torch.sigmoid(sign * lower))[:, 0, :]
        tail_mass = torch.sigmoid(lower[:, 0, :1]) + torch.sigmoid(-upper[:, 0, -1:])
        self._quantized_cdf = _quantized_cdf_table(pmf, tail_mass, pmf_length, max_length).to(dev)
        self._cdf_length = (pmf_length + 2).int().to(dev)
        self._offset = (-minima).int().to(dev)
        self._invalidate_tables()
        return True

    def _channel_indexes(self, c, h, w) -> np.ndarray:
        return np.repeat(np.arange(c, dtype=np.int32), h * w)

    def compress(self, x: torch.Tensor) -> List[bytes]:
        _, sym = self.quantize_symbols(x)
        return self.compress_symbols(sym.cpu().numpy())

    def compress_symbols(self, sym: np.ndarray) -> List[bytes]:
        n, c, h, w = sym.shape
        tables = self.coder_tables()
        idx = self._channel_indexes(c, h, w)
        return [tables.encode(sym[i], idx) for i in range(n)]

    def decompress_symbols(self, strings: List[bytes], size) -> np.ndarray:
        tables = self.coder_tables()
        c = self._quantized_cdf.size(0)
        h, w = int(size[0]), int(size[1])
        idx = self._channel_indexes(c, h, w)
        return np.stack([tables.decode(s, idx).reshape(c, h, w) for s in strings])

    def decompress(self, strings: List[bytes], size) -> torch.Tensor:
        sym = torch.from_numpy(self.decompress_symbols(strings, size)).to(self.quantiles.device)
        return ops.eb_dequantize(sym, self._medians_flat())


class GaussianConditional(EntropyModel):
    def __init__(self, scale_table, scale_bound: float = 0.11, tail_mass: float = 1e-9):
        super().__init__()
        if abs(scale_bound - 0.11) > 1e-12:
            raise ValueError("the HIP Gaussian kernels are specialised for scale_bound=0.11")
        self.tail_mass = float(tail_mass)
        self.scale_bound = float(scale_bound)
        self.register_buffer(
            "scale_table",
            torch.Tensor(tuple(float(s) for s in scale_table)) if scale_table is not None else torch.Tensor())

    def forward(self, inputs, scales, means=None, training: Optional[bool] = None):
        if training is None:
            training = self.training
        n, c, h, w = inputs.shape
        noise = None
        if training:
            noise = uniform_noise(inputs.shape, inputs.device)
        out = torch.empty((n, c, h, w), dtype=torch.float32, device=inputs.device)
        lik = torch.empty_like(out)
        mp = C.byref(ops._src(means, "means")) if means is not None else None
        _lib.check(_lib.load().dvc_gaussian_conditional(
            C.byref(ops._src(inputs, "inputs")), C.byref(ops._src(scales, "scales")), mp, ops._ptr(noise),
            C.byref(ops._dst(out)), C.byref(ops._dst(lik)), n, h, w, ops._stream()), "dvc_gaussian_conditional")
        return out, lik

    def build_indexes(self, scales: torch.Tensor) -> torch.Tensor:
        n, c, h, w = scales.shape
        idx = torch.empty((n, c, h, w), dtype=torch.int32, device=scales.device)
        table = self.scale_table.float().contiguous()
        _lib.check(_lib.load().dvc_build_indexes(C.byref(ops._src(scales, "scales")), idx.data_ptr(),
                                                 table.data_ptr(), table.numel(), n, h, w, ops._stream()),
                   "dvc_build_indexes")
        return idx

    def update_scale_table(self, scale_table, force: bool = False) -> bool:
        if self._offset.numel() > 0 and not force:
            return False
        device = self.scale_table.device
        self.scale_table = torch.Tensor(tuple(float(s) for s in scale_table)).to(device)
        self.update()
        return True

    @torch.no_grad()
    def update(self):
        dev = self.scale_table.device
        table = self.scale_table.detach().cpu().float()
        multiplier = -scipy.stats.norm.ppf(self.tail_mass / 2)
        pmf_center = torch.ceil(table * multiplier).int()
        pmf_length = 2 * pmf_center + 1
        max_length = int(torch.max(pmf_length).item())
        samples = torch.abs(torch.arange(max_length).int() - pmf_center[:, None]).float()
        scale = table.unsqueeze(1)
        const = -(2 ** -0.5)
        upper = 0.5 * torch.erfc(const * ((0.5 - samples) / scale))
        lower = 0.5 * torch.erfc(const * ((-0.5 - samples) / scale))
        pmf = upper - lower
        tail_mass = 2 * lower[:, :1]
        self._quantized_cdf = _quantized_cdf_table(pmf, tail_mass, pmf_length, max_length).to(dev)
        self._offset = (-pmf_center).int().to(dev)
        self._cdf_length = (pmf_length + 2).int().to(dev)
        self._invalidate_tables()

    def compress(self, inputs: torch.Tensor, indexes: torch.Tensor, means=None) -> List[bytes]:
        """inputs: values to code (already mean-removed when means is None).  Generic entry
        point: rounding happens on the host copy; the DMC path feeds `compress_symbols`
        with the int32 symbols its fused quantisation kernel already produced."""
        x = inputs.detach().cpu().numpy()
        if means is not None:
            x = x - means.detach().cpu().numpy()
        sym = x if x.dtype == np.int32 else np.rint(x).astype(np.int32)
        return self.compress_symbols(sym, indexes.cpu().numpy())

    def compress_symbols(self, sym: np.ndarray, idx: np.ndarray) -> List[bytes]:
        tables = self.coder_tables()
        return [tables.encode(sym[i], idx[i]) for i in range(sym.shape[0])]

    def decompress_symbols(self, strings: List[bytes], idx: np.ndarray) -> np.ndarray:
        tables = self.coder_tables()
        return np.stack([tables.decode(s, idx[i]).reshape(idx[i].shape) for i, s in enumerate(strings)])

    def decompress(self, strings: List[bytes], indexes: torch.Tensor, dtype=torch.float, means=None):
        sym = self.decompress_symbols(strings, indexes.cpu().numpy()).astype(np.float32)
        if means is not None:
            sym = sym + means.detach().cpu().numpy()
        return torch.from_numpy(sym).to(indexes.device).to(dtype)
