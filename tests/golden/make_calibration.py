#!/usr/bin/env python3
"""Data-dependent gains that put the seeded random-init DMC at a trained codec's operating point.

Why: the reference's init (Xavier-normal, gain sqrt 2, video_model.py:508-513) on ~180
un-normalised conv layers blows activations up to ~1e4, and an untrained entropy model prices
every latent at many bits (several bpp), with most latents a coin flip away from a rounding
boundary.  A trained codec keeps activations O(1), concentrates the latent energy in a few
channels (most channels quantise to zero everywhere) and predicts scales that match that
energy, so it codes ~0.05-0.5 bpp.  No trained weights exist offline, so this script emulates
that conditioning on the CPU oracle:

  * every convolution gets a scalar gain so that its output has standard deviation 1 (LSUV style);
  * the two latent heads (motion_encoder.6, contextual_encoder.conv4) get PER-CHANNEL gains: the
    standard deviation of channel c follows a geometric profile (a few channels span several
    quantisation bins, the rest are far inside the zero bin), zero mean;
  * the (mean, scale) heads of the prior networks get per-channel gains and bias shifts so that the
    predicted scale of channel c is the profile's value +- 15 % and the predicted mean a tenth of it;
  * the hyper-latent heads get a profile as well and the factorised prior's first matrix is
    sharpened per channel so that its density has the matching width.

Gains are fitted by fixed-point iteration on the statistics of the LAST coded frame of a short
GOP (so the layers fed by the decoded-picture buffer see their steady-state inputs), at the
headline size by default.  Only gains / shifts are stored
(deepvideocodec_amd/data/calibration_seed0.json); weights are still regenerated from the seed.

Run in the build container:  python tests/golden/make_calibration.py [H W [iterations]]
"""
import json
import math
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from deepvideocodec_amd import testing  # noqa: E402
from deepvideocodec_amd.testing import pad_to_multiple, synthetic_gop  # noqa: E402
from oracle import dmc_oracle as O  # noqa: E402

CH_MV, N, M = 64, 64, 96


def y_profile(channels, top=3.0, per_octave=2.0):
    """Target std of latent channel c; both channel halves (the two checkerboard groups) get the
    same geometric profile."""
    half = channels // 2
    return torch.tensor([top * 2.0 ** (-(c % half) / per_octave) for c in range(channels)])


def z_profile(channels, top=1.5, per_octave=8.0, floor=0.02):
    return torch.tensor([max(top * 2.0 ** (-c / per_octave), floor) for c in range(channels)])


# scalar output-std targets; anything not listed gets 1.0
TARGET_STD = {
    "recon_generation_net.recon_conv": 0.15,
    "motion_decoder.7.0": 1.0,                     # decoded flow, pixels
}
for lvl in range(4):
    TARGET_STD[f"optic_flow.moduleBasic.{lvl}.conv5"] = 0.5   # per-level flow refinement, pixels

# per-channel targets: name -> (std per channel, mean per channel or None to leave the mean alone)
Y_MV, Y_FR = y_profile(CH_MV), y_profile(M)
CHANNEL_TARGETS = {
    "motion_encoder.6": (Y_MV, torch.zeros(CH_MV)),
    "contextual_encoder.conv4": (Y_FR, torch.zeros(M)),
    "motion_context_model.hyper_encoder.8": (z_profile(CH_MV), torch.zeros(CH_MV)),
    "frame_context_model.hyper_encoder.8": (z_profile(N), torch.zeros(N)),
}
# the two feature adaptors (first P frame of a GOP: from the intra picture; later frames: from the
# previous frame's features) are normalised per channel, so that everything downstream sees the same
# first and second moments on both paths
for _name in ("feature_adaptor_I", "feature_adaptor_P"):
    CHANNEL_TARGETS[_name] = (torch.ones(N), torch.zeros(N))
# predicted mean = MEAN_FRAC x profile (noise), predicted scale = SCALE_MARGIN x profile +- SCALE_NOISE x profile.
# The margin keeps latents of frames whose statistics differ from the fitted ones (other content, the first P
# frame of a GOP) inside ~4 predicted standard deviations, where the 16-bit CDF tables still resolve them.
MEAN_FRAC, SCALE_NOISE, SCALE_MARGIN = 0.10, 0.15, float(os.environ.get("CAL_SCALE_MARGIN", "1.5"))


def _head(profile, spatial):
    """(mean, scale) head targets.  y_prior_fusion.4 emits [means(C), scales(C)]; y_spatial_prior.4
    emits [mu_0, sigma_0, mu_1, sigma_1] for the two channel halves (video_model.py:190)."""
    c = profile.numel()
    if spatial:
        p0, p1 = profile[:c // 2], profile[c // 2:]
        std = torch.cat([MEAN_FRAC * p0, SCALE_NOISE * p0, MEAN_FRAC * p1, SCALE_NOISE * p1])
        mean = torch.cat([torch.zeros_like(p0), SCALE_MARGIN * p0, torch.zeros_like(p1), SCALE_MARGIN * p1])
    else:
        std = torch.cat([MEAN_FRAC * profile, SCALE_NOISE * profile])
        mean = torch.cat([torch.zeros_like(profile), SCALE_MARGIN * profile])
    return std, mean


for _key, _prof in (("motion_context_model", Y_MV), ("frame_context_model", Y_FR)):
    CHANNEL_TARGETS[f"{_key}.y_prior_fusion.4"] = _head(_prof, spatial=False)
    CHANNEL_TARGETS[f"{_key}.y_spatial_prior.4"] = _head(_prof, spatial=True)

# logistic scale of the factorised prior per unit of target std (a logistic of scale s has std 1.81 s)
EB_SCALE_PER_STD = 0.55
EB_INIT_SCALE = 10.0     # width of compressai's initial density (init_scale)


def main():
    height = int(sys.argv[1]) if len(sys.argv) > 1 else 1080
    width = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
    iterations = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    torch.manual_seed(0)
    torch.set_num_threads(8)
    seed = 0
    raw = testing.seeded_state_dict(seed=seed, calibrated=False)
    conv_names = [k[:-len(".weight")] for k, v in raw.items() if k.endswith(".weight") and v.dim() == 4]
    gain = {n: torch.ones(raw[n + ".weight"].shape[0]) for n in conv_names}
    shift = {n: torch.zeros(raw[n + ".weight"].shape[0]) for n in conv_names}
    eb_sharpen = {"motion_context_model": (EB_INIT_SCALE / (EB_SCALE_PER_STD * z_profile(CH_MV))).tolist(),
                  "frame_context_model": (EB_INIT_SCALE / (EB_SCALE_PER_STD * z_profile(N))).tolist()}

    def build():
        cal = {"gain": {n: g.tolist() for n, g in gain.items()},
               "bias_shift": {n: s.tolist() for n, s in shift.items()}, "eb_sharpen": eb_sharpen}
        return testing.apply_calibration(raw, cal)

    frames = [pad_to_multiple(f, 64) for f in synthetic_gop(int(os.environ.get('CAL_FRAMES', '4')), height, width, seed=100)]
    n_pix = frames[0].shape[2] * frames[0].shape[3]
    plain_conv = O.conv
    fitting = [False]
    sd = build()

    def measure(name, out):
        if name in CHANNEL_TARGETS or name == "recon_generation_net.recon_conv":
            var_c, mean_c = torch.var_mean(out, dim=(0, 2, 3))
            return {"mean_c": mean_c, "std_c": var_c.sqrt().clamp_min(1e-12), "std": max(out.std().item(), 1e-12)}
        return {"std": max(out.std().item(), 1e-12)}

    def fit(name, m):
        """Update gain / shift of `name` from its output statistics; returns (ratio, old shift, new shift)
        per channel: the corrected output is (out - old) * ratio + new."""
        old = shift[name]
        if name in CHANNEL_TARGETS:
            t_std, t_mean = CHANNEL_TARGETS[name]
            r = t_std / m["std_c"]
            new = t_mean - r * (m["mean_c"] - old)
        else:
            r = torch.full_like(gain[name], TARGET_STD.get(name, 1.0) / m["std"])
            new = old * 0
            if name == "recon_generation_net.recon_conv":
                new = 0.5 - r * (m["mean_c"] - old)
        gain[name] = gain[name] * r
        shift[name] = new
        return r, old, new

    def hooked_conv(sd_, name, x, stride=1):
        out = plain_conv(sd_, name, x, stride)
        if fitting[0]:
            # sequential (LSUV-style) fit: each layer is corrected where it is called, so everything
            # downstream already sees conditioned inputs -- a simultaneous update of all layers from one
            # set of statistics compounds the corrections along the depth of the net and diverges
            r, old, new = fit(name, measure(name, out))
            out = (out - old.view(1, -1, 1, 1)) * r.view(1, -1, 1, 1) + new.view(1, -1, 1, 1)
            sd_[name + ".weight"] = raw[name + ".weight"] * gain[name].view(-1, 1, 1, 1)
            sd_[name + ".bias"] = raw[name + ".bias"] * gain[name] + shift[name]
        return out

    O.conv = hooked_conv

    def report(label, i, dt, frame, x, lik, trace):
        tot = ex = 0.0
        b = {}
        for a_, d in lik.items():
            for k, v in d.items():
                bits = -torch.log2(v)
                b[f"{a_}.{k}"] = round(bits.sum().item() / n_pix, 4)
                tot += bits.sum().item()
                ex += (bits - 16).clamp_min(0).sum().item()
        print(f"[{label}] frame {i}: {dt:.0f}s psnr {O.psnr(frame, x):.2f} bpp {tot / n_pix:.4f} {b} "
              f"bits beyond the 16-bit table resolution: {ex / tot:.4f}", flush=True)
        for key, tr in trace.items():
            yp = tr["y_pre"]
            print(f"    {key}: |y-mu| mean {yp.abs().mean():.3f} max {yp.abs().max():.1f}, nonzero symbols "
                  f"{(yp.round() != 0).float().mean():.4f}, z nonzero {(tr['z_pre'].round() != 0).float().mean():.4f}")

    def run_gop(label, gop, n=None):
        """Plain coding of a GOP with the current gains; returns the buffer after the first P frame."""
        ent = O.EntropyState(sd)
        dpb = {"x_ref": gop[0], "feature_ref": None, "y_ref": None, "y_mv_ref": None}
        first = None
        for i in range(1, n or len(gop)):
            trace = {}
            t0 = time.time()
            with torch.no_grad():
                x, lik, ctx = O.forward_inter(sd, ent, gop[i], dpb, trace=trace)
            report(label, i, time.time() - t0, gop[i], x, lik, trace)
            dpb = dict(ctx)
            first = first or dpb
        return first

    plain_mc = O.motion_compensation

    def mixed_motion_compensation(sd_, mv, dpb):
        """video_model.py:490-506 for a batch whose sample 0 is the first P frame of a GOP (features from
        the intra picture through feature_adaptor_I) and whose sample 1 is a later frame (features from the
        previous frame's feature_ref through feature_adaptor_P)."""
        warpframe = O.flow_warp(dpb["x_ref"], mv)
        mv2 = O.bilinear_down2(mv) / 2
        mv3 = O.bilinear_down2(mv2) / 2
        feature = torch.cat([O.conv(sd_, "feature_adaptor_I", dpb["x_ref"][0:1]),
                             O.conv(sd_, "feature_adaptor_P", dpb["feature_ref"][1:2])], 0)
        f1, f2, f3 = O.feature_extractor(sd_, feature)
        c1, c2, c3 = O.context_fusion(sd_, O.flow_warp(f1, mv), O.flow_warp(f2, mv2), O.flow_warp(f3, mv3))
        return c1, c2, c3, warpframe

    def fit_pass(label, warm):
        """One sequential fit over a batch of two: [first P frame with an empty buffer, second P frame with
        the buffer `warm` left by the first] -- every layer is fitted to the pooled statistics of both."""
        ent = O.EntropyState(sd)
        z = torch.zeros_like
        dpb = {"x_ref": torch.cat([frames[0], warm["x_ref"]], 0),
               "feature_ref": torch.cat([z(warm["feature_ref"]), warm["feature_ref"]], 0),
               "y_ref": torch.cat([z(warm["y_ref"]), warm["y_ref"]], 0),          # zeros == "no reference latents"
               "y_mv_ref": torch.cat([z(warm["y_mv_ref"]), warm["y_mv_ref"]], 0)}
        x_cur = torch.cat([frames[1], frames[2]], 0)
        O.motion_compensation = mixed_motion_compensation
        fitting[0] = True
        t0 = time.time()
        try:
            with torch.no_grad():
                O.forward_inter(sd, ent, x_cur, dpb)
        finally:
            fitting[0] = False
            O.motion_compensation = plain_mc
        print(f"[{label}] fitted in {time.time() - t0:.0f}s", flush=True)

    # cold start: fit on the first P frame alone, then refit on [first, second] P frame pooled
    fitting[0] = True
    run_gop("cold fit", frames, n=2)
    fitting[0] = False
    for it in range(iterations):
        warm = run_gop(f"pass {it + 1}: buffer", frames, n=2)
        fit_pass(f"pass {it + 1}", warm)
    run_gop("check, fitted GOP", frames)
    other = [pad_to_multiple(f, 64) for f in synthetic_gop(3, height, width, seed=1)]
    run_gop("check, other content", other)

    cal = {"seed": seed, "fitted_at": [height, width],
           "gain": {n: (g.tolist() if n in CHANNEL_TARGETS else float(g[0])) for n, g in gain.items()},
           "bias_shift": {n: s.tolist() for n, s in shift.items() if bool(s.abs().max() > 0)},
           "eb_sharpen": eb_sharpen}
    out_dir = os.path.join(ROOT, "deepvideocodec_amd", "data")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, f"calibration_seed{seed}.json"), "w") as f:
        json.dump(cal, f, sort_keys=True)
    print(len(gain), "layers calibrated")


if __name__ == "__main__":
    main()
