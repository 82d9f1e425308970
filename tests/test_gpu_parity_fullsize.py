"""Parity at the headline size: the first two P frames of a 1080p GOP (padded to 1088x1920) through
the product path on the GPU against the CPU oracle on the same inputs, frame by frame from the oracle's
decoded-picture buffer.

Bar for the exact fp32 path (the one that carries the benchmark value), absolute, as stated for the path:
|dPSNR| <= 1e-3 dB, |dbpp| <= 1e-4 (likelihood bpp, dmc/train.py:74-93), and at most 1e-4 of the 1.3 M
quantised latents on the other side of a rounding boundary (round() sits in the middle of the graph; two
fp32 implementations with different summation orders may disagree on a latent within ~1e-6 of x.5).
The fp16 hi+lo split arithmetic (never the benchmark value) is run beside it and held to 5x those bars.
The oracle needs ~10 s per frame on 16 cores."""
import math

import pytest
import torch

from oracle import dmc_oracle as O

pytestmark = pytest.mark.gpu

PSNR_TOL = 1e-3
BPP_TOL = 1e-4
N_LATENTS = 160 * 68 * 120
FLIP_TOL = 1e-4 * N_LATENTS


def psnr(x, rec):
    return -10.0 * math.log10(torch.mean((x.double() - rec.double()) ** 2).item())


def bpp(lik, n_pix):
    return sum(-torch.log2(t.double()).sum().item() for v in lik.values() for t in v.values()) / n_pix


@pytest.mark.timeout(900)
def test_first_p_frames_1080p_match_oracle(gpu_net, seeded_sd, oracle_entropy):
    from deepvideocodec_amd.testing import pad_to_multiple, synthetic_gop
    frames = [pad_to_multiple(f, 64) for f in synthetic_gop(3, 1080, 1920, seed=4)]
    n_pix = 1088 * 1920
    dpb = {"x_ref": frames[0], "feature_ref": None, "y_ref": None, "y_mv_ref": None}
    report, rate = [], {}
    try:
        with torch.no_grad():
            for i in (1, 2):
                x_o, lik_o, ctx_o = O.forward_inter(seeded_sd, oracle_entropy, frames[i], dpb)
                p_o, b_o = psnr(frames[i], x_o), bpp(lik_o, n_pix)
                rate[i] = (b_o, p_o)
                dpb_g = {k: (None if v is None else v.cuda()) for k, v in dpb.items()}
                for mode in ("fp32", "fp16x2"):
                    gpu_net.set_precision(mode)
                    x_g, lik_g, ctx_g = gpu_net.forward_inter(frames[i].cuda(), dpb_g)
                    lik_g = {a: {b: t.cpu() for b, t in v.items()} for a, v in lik_g.items()}
                    flips = sum(int(((ctx_g[k].cpu() - ctx_o[k]).abs() > 0.5).sum()) for k in ("y_ref", "y_mv_ref"))
                    report.append((i, mode, psnr(frames[i], x_g.cpu()) - p_o, bpp(lik_g, n_pix) - b_o, flips))
                dpb = dict(ctx_o)
    finally:
        gpu_net.set_precision("fp32")
    print("\n".join(f"frame {i} {m}: oracle {rate[i][0]:.4f} bpp / {rate[i][1]:.3f} dB; dPSNR {dp:+.2e} dB, dbpp {db:+.2e}, "
                    f"{fl} of {N_LATENTS} latents differ" for i, m, dp, db, fl in report))
    for i, (b_o, _) in rate.items():
        assert 0.02 <= b_o <= 1.0, f"frame {i}: {b_o} bpp is not a realistic operating point for the parity bar"
    for i, mode, d_psnr, d_bpp, flips in report:
        k = 1.0 if mode == "fp32" else 5.0
        assert abs(d_psnr) <= k * PSNR_TOL, (i, mode, d_psnr)
        assert abs(d_bpp) <= k * BPP_TOL, (i, mode, d_bpp)
        assert flips <= k * FLIP_TOL, (i, mode, flips)
