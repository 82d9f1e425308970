#!/usr/bin/env python3
"""Data-dependent gains that make the seeded random-init DMC numerically well behaved.

Why: the reference's init (Xavier-normal, gain sqrt 2, video_model.py:508-513) on ~180
un-normalised conv layers blows activations up to ~1e4 and latents to ~1e3, so an
untrained model's PSNR / bpp say nothing (and round() parity is dominated by the blow-up).
A trained codec keeps activations O(1).  Since no trained weights exist offline, this
script emulates that conditioning: it runs the CPU oracle once on a synthetic GOP and
rescales every convolution (LSUV style) so its output has a chosen standard deviation.
Only the per-layer gains / bias shifts are stored (deepvideocodec_amd/data/calibration_seed0.json);
weights are still regenerated from the seed.

Run in the build container:  python tests/golden/make_calibration.py
"""
import json
import os
import sys

import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from deepvideocodec_amd.testing import seeded_state_dict, synthetic_gop  # noqa: E402
from oracle import dmc_oracle as O  # noqa: E402

# output std targets; anything not listed gets 1.0
TARGET_STD = {
    "recon_generation_net.recon_conv": 0.15,
    "motion_decoder.7.0": 1.0,                     # decoded flow, pixels
    "motion_encoder.6": 3.0,                       # y_mv: a few quantisation bins wide
    "contextual_encoder.conv4": 3.0,               # y
    "motion_context_model.hyper_encoder.8": 3.0,   # z
    "frame_context_model.hyper_encoder.8": 3.0,
}
for lvl in range(4):
    TARGET_STD[f"optic_flow.moduleBasic.{lvl}.conv5"] = 0.5   # per-level flow refinement, pixels
# (mean, scale) heads: shift the scale half so that most scales are positive
SCALE_HEADS = ["motion_context_model.y_prior_fusion.4", "motion_context_model.y_spatial_prior.4",
               "frame_context_model.y_prior_fusion.4", "frame_context_model.y_spatial_prior.4"]


def main():
    torch.manual_seed(0)
    seed = 0
    sd = seeded_state_dict(seed=seed, calibrated=False)
    gains, bias_shift = {}, {}

    def calibrated_conv(sd_, name, x, stride=1):
        w = sd_[name + ".weight"]
        b = sd_[name + ".bias"]
        pad = w.shape[-1] // 2
        if name not in gains:
            out = F.conv2d(x, w, b, stride=stride, padding=pad)
            std = out.std().item()
            g = TARGET_STD.get(name, 1.0) / max(std, 1e-12)
            gains[name] = g
            sd_[name + ".weight"] = w = w * g
            sd_[name + ".bias"] = b = b * g
            shift = torch.zeros_like(b)
            if name == "recon_generation_net.recon_conv":
                out = F.conv2d(x, w, b, stride=stride, padding=pad)
                shift = 0.5 - out.mean(dim=(0, 2, 3))
            elif name in SCALE_HEADS:
                c = b.numel()
                if "spatial" in name:      # chunks mu0, sigma0, mu1, sigma1
                    shift[c // 4:c // 2] = 1.5
                    shift[3 * c // 4:] = 1.5
                else:                      # chunks means, scales
                    shift[c // 2:] = 1.5
            bias_shift[name] = shift.tolist()
            sd_[name + ".bias"] = b + shift
        return F.conv2d(x, sd_[name + ".weight"], sd_[name + ".bias"], stride=stride, padding=pad)

    O.conv = calibrated_conv
    ent = O.EntropyState(sd)
    frames = synthetic_gop(3, 128, 192, seed=100)
    with torch.no_grad():
        res = O.forward(sd, ent, frames)
    for i, xh in enumerate(res["x_hat"]):
        print(f"x_hat{i}: mean {xh.mean():.3f} std {xh.std():.3f} min {xh.min():.3f} max {xh.max():.3f}",
              "psnr", O.psnr(frames[i + 1], xh))
    print("bpp per frame:", [O.likelihood_bpp([l], 128 * 192).item() for l in res["likelihoods"]])
    out_dir = os.path.join(ROOT, "deepvideocodec_amd", "data")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, f"calibration_seed{seed}.json"), "w") as f:
        json.dump({"seed": seed, "gain": gains, "bias_shift": {k: v for k, v in bias_shift.items() if any(v)}},
                  f, indent=0, sort_keys=True)
    print(len(gains), "layers calibrated")


if __name__ == "__main__":
    main()
