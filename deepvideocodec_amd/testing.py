"""Deterministic synthetic weights and frames shared by the tests, the fixture
generator, the smoke test and the benchmark (there are no trained DMC weights and
no dataset offline, so every run uses seeded random-init weights and synthetic video).

Everything here is generated on the CPU with explicitly seeded generators, so the
same tensors are reproduced on any machine with the same torch build.
"""
import hashlib
import math
from typing import Dict, List

import torch
import torch.nn.functional as F


def _gen(seed: int, name: str) -> torch.Generator:
    h = int.from_bytes(hashlib.sha256(f"{seed}:{name}".encode()).digest()[:7], "little")
    g = torch.Generator(device="cpu")
    g.manual_seed(h)
    return g


def _calibration(seed: int):
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", f"calibration_seed{seed}.json")
    if not os.path.isfile(path):
        raise FileNotFoundError(f"no calibration gains for seed {seed} ({path}); "
                                "generate them with tests/golden/make_calibration.py or pass calibrated=False")
    with open(path) as f:
        return json.load(f)


def seeded_state_dict(seed: int = 0, ch_mv: int = 64, N: int = 64, M: int = 96,
                      calibrated: bool = True) -> Dict[str, torch.Tensor]:
    """A full DMC state dict (reference key names) with the reference's init distributions
    (Xavier-normal gain sqrt(2) conv weights) drawn from per-tensor seeded generators, plus
    mildly perturbed biases / factorized-prior parameters so that every parameter matters.

    calibrated=True additionally applies the stored per-layer gains (data/calibration_seed*.json)
    that keep activations O(1) like a trained codec's, so that PSNR / bpp are meaningful;
    calibrated=False is the raw reference-style init (activations grow to ~1e4)."""
    sd = _raw_seeded_state_dict(seed, ch_mv, N, M)
    if calibrated:
        sd = apply_calibration(sd, _calibration(seed))
    return sd


def apply_calibration(sd: Dict[str, torch.Tensor], cal) -> Dict[str, torch.Tensor]:
    """A copy of the raw seeded state dict with the calibration applied:
    conv weight *= gain (a scalar, or one value per output channel), bias = bias * gain + bias_shift;
    the first matrix of each factorised prior is sharpened per channel (softplus(m') = k softplus(m),
    i.e. the density gets k times narrower)."""
    sd = dict(sd)
    for name, g in cal["gain"].items():
        g = torch.as_tensor(g, dtype=torch.float32)
        sd[name + ".weight"] = sd[name + ".weight"] * (g.view(-1, 1, 1, 1) if g.dim() else g)
        sd[name + ".bias"] = sd[name + ".bias"] * g
    for name, shift in cal["bias_shift"].items():
        sd[name + ".bias"] = sd[name + ".bias"] + torch.as_tensor(shift, dtype=torch.float32)
    for key, k in cal.get("eb_sharpen", {}).items():
        name = f"{key}.entropy_bottleneck._matrix0"
        m = sd[name].double()
        target = torch.as_tensor(k, dtype=torch.float64).view(-1, 1, 1) * torch.nn.functional.softplus(m)
        # inverse softplus; for large arguments log(expm1(t)) == t to double precision
        sd[name] = torch.where(target > 30.0, target, torch.log(torch.expm1(target.clamp(max=30.0)))).float()
    return sd


def _raw_seeded_state_dict(seed: int, ch_mv: int, N: int, M: int) -> Dict[str, torch.Tensor]:
    from .models import DMC

    net = DMC(ch_mv, N, M)
    sd = {}
    for name, t in net.state_dict().items():
        g = _gen(seed, name)
        leaf = name.rsplit(".", 1)[-1]
        if t.numel() == 0:            # empty CDF buffers before update()
            sd[name] = t.clone()
        elif t.dim() == 4:            # conv weight
            fan_in = t.shape[1] * t.shape[2] * t.shape[3]
            fan_out = t.shape[0] * t.shape[2] * t.shape[3]
            std = math.sqrt(2.0) * math.sqrt(2.0 / (fan_in + fan_out))
            sd[name] = torch.randn(t.shape, generator=g) * std
        elif leaf == "bias":
            sd[name] = 0.01 + 0.02 * torch.randn(t.shape, generator=g)
        elif t.dim() == 2:            # squeeze-excite linear
            bound = 1.0 / math.sqrt(t.shape[1])
            sd[name] = (torch.rand(t.shape, generator=g) * 2 - 1) * bound
        elif leaf.startswith("_matrix"):
            sd[name] = t.clone() + 0.1 * torch.randn(t.shape, generator=g)
        elif leaf.startswith("_bias"):
            sd[name] = torch.rand(t.shape, generator=g) - 0.5
        elif leaf.startswith("_factor"):
            sd[name] = 0.3 * torch.randn(t.shape, generator=g)
        elif leaf == "quantiles":
            med = (torch.rand(t.shape[0], generator=g) - 0.5) * 0.6
            width = 6.0 + 6.0 * torch.rand(t.shape[0], generator=g)
            sd[name] = torch.stack([med - width, med, med + width], dim=1).view(t.shape)
        else:                         # `target` and other constant buffers
            sd[name] = t.clone()
    return sd


def synthetic_gop(num_frames: int, height: int, width: int, seed: int = 0, batch: int = 1,
                  sigma: float = 8.0, max_shift: float = 4.0, noise_std: float = 0.01) -> List[torch.Tensor]:
    """A moving smooth scene: Gaussian-blurred noise field translated by a per-frame global
    shift drawn from U(-max_shift, max_shift) px, plus N(0, noise_std^2) noise, in [0, 1]."""
    g = _gen(seed, f"gop{num_frames}x{height}x{width}x{batch}")
    margin = int(math.ceil(max_shift * num_frames)) + 2
    hh, ww = height + 2 * margin, width + 2 * margin
    # blur at 1/4 resolution then upsample: cheap and smooth enough
    sh, sw = (hh + 3) // 4, (ww + 3) // 4
    field = torch.randn(batch, 3, sh, sw, generator=g)
    s4 = max(sigma / 4.0, 0.5)
    rad = int(math.ceil(3 * s4))
    k = torch.exp(-0.5 * (torch.arange(-rad, rad + 1, dtype=torch.float32) / s4) ** 2)
    k = (k / k.sum()).view(1, 1, -1)
    field = F.pad(field, (rad, rad, rad, rad), mode="reflect")
    field = F.conv2d(field.reshape(-1, 1, sh + 2 * rad, sw + 2 * rad), k.unsqueeze(2))
    field = F.conv2d(field, k.unsqueeze(3)).reshape(batch, 3, sh, sw)
    field = F.interpolate(field, size=(hh, ww), mode="bilinear", align_corners=False)
    field = field - field.amin(dim=(1, 2, 3), keepdim=True)
    field = field / field.amax(dim=(1, 2, 3), keepdim=True).clamp_min(1e-6)
    frames = []
    pos_x = pos_y = 0.0
    for t in range(num_frames):
        if t > 0:
            pos_x += (torch.rand(1, generator=g).item() * 2 - 1) * max_shift
            pos_y += (torch.rand(1, generator=g).item() * 2 - 1) * max_shift
        # sub-pixel shift by bilinear sampling of the big field
        x0, y0 = margin + pos_x, margin + pos_y
        ix, iy = int(math.floor(x0)), int(math.floor(y0))
        fx, fy = x0 - ix, y0 - iy
        win = field[:, :, iy:iy + height + 1, ix:ix + width + 1]
        fr = ((1 - fy) * ((1 - fx) * win[:, :, :-1, :-1] + fx * win[:, :, :-1, 1:]) +
              fy * ((1 - fx) * win[:, :, 1:, :-1] + fx * win[:, :, 1:, 1:]))
        fr = fr + noise_std * torch.randn(fr.shape, generator=g)
        frames.append(fr.clamp(0.0, 1.0).contiguous())
    return frames


def random_gop(num_frames: int, height: int, width: int, seed: int = 0, batch: int = 1) -> List[torch.Tensor]:
    """Independent U[0,1) frames (worst case for motion estimation and entropy coding)."""
    g = _gen(seed, f"rand{num_frames}x{height}x{width}x{batch}")
    return [torch.rand(batch, 3, height, width, generator=g) for _ in range(num_frames)]


def pad_to_multiple(x: torch.Tensor, p: int = 64) -> torch.Tensor:
    """Centred zero padding to a multiple of p (dmc/test.py:75-88)."""
    h, w = x.size(2), x.size(3)
    H, W = (h + p - 1) // p * p, (w + p - 1) // p * p
    left = (W - w) // 2
    top = (H - h) // 2
    return F.pad(x, (left, W - w - left, top, H - h - top), mode="constant", value=0)


def crop_to(x: torch.Tensor, size) -> torch.Tensor:
    """Inverse of pad_to_multiple (dmc/test.py:91-103)."""
    H, W = x.size(2), x.size(3)
    h, w = size
    left = (W - w) // 2
    top = (H - h) // 2
    return x[:, :, top:top + h, left:left + w]
