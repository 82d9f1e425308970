"""Free-running whole-GOP parity: the evaluation loop of dmc/test.py:162-196 (encode_inter ->
decode_inter -> clamp -> buffer update, the buffer carried from frame to frame) run independently on the
GPU product path and on the CPU oracle -- no teacher forcing, so any error feeds forward through the
decoded-picture buffer for 11 (GOP 12) or 31 (GOP 32, the reference's hard-coded intra period) frames.

Asserted on the GOP average: |dPSNR| <= 1e-3 dB and |dbpp| <= 1e-4, bpp from the bytes actually written
(dmc/test.py:213-217).  The per-frame drift is printed.  The exact fp32 path and the two split-operand modes
(fp16 and bf16 hi + lo parts on the 16-bit matrix instructions) are held to the same bar against one oracle run.
The last case is the benchmark's own configuration: 1080 x 1920 (padded to 1088 x 1920), GOP 12, 11 P frames,
the GOP bench.py codes with seed 0 (about 130 s of oracle time on 16 host threads); per-frame |dPSNR| is held to
the same 1e-3 dB there as well."""
import math

import pytest
import torch

from oracle import dmc_oracle as O

pytestmark = pytest.mark.gpu


def psnr(x, rec):
    return -10.0 * math.log10(torch.mean((x.double() - rec.double()) ** 2).item())


def _code_gop(enc_fn, dec_fn, frames, to_dev, clamp):
    dpb = {"x_ref": to_dev(frames[0]), "feature_ref": None, "y_ref": None, "y_mv_ref": None}
    per_frame = []
    for x in frames[1:]:
        enc = enc_fn(to_dev(x), dpb)
        rec, info = dec_fn(enc["strings"], enc["shape"], dpb)
        rec = clamp(rec)
        nbytes = sum(len(s[0]) for v in enc["strings"].values() for s in v)
        per_frame.append((nbytes, psnr(x, rec.cpu()), enc["strings"]))
        dpb = {"x_ref": rec, "feature_ref": info["feature_ref"], "y_ref": info["y_ref"], "y_mv_ref": info["y_mv_ref"]}
    return per_frame


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("gop,size,seed", [(12, 256, 31), (32, 128, 32), (4, (180, 420), 33), (12, (1080, 1920), 0)],
                         ids=["gop12-256", "gop32-128", "gop4-180x420", "gop12-1080p"])
def test_free_running_gop_matches_oracle(gpu_net, seeded_sd, oracle_entropy, gop, size, seed):
    from deepvideocodec_amd import ops
    from deepvideocodec_amd.testing import pad_to_multiple, synthetic_gop
    # (a pair: a frame size that is not a multiple of 64, zero-padded as dmc/test.py:75-88 does -- 192 x 448 here,
    # whose 1/4 .. 1/16 scales leave partial 32-pixel tiles in every convolution variant)
    h, w = (size, size) if isinstance(size, int) else size
    frames = [pad_to_multiple(f, 64) for f in synthetic_gop(gop, h, w, seed=seed)]
    h, w = frames[0].shape[-2:]
    n_pix = h * w
    with torch.no_grad():
        ref = _code_gop(lambda x, d: O.encode_inter(seeded_sd, oracle_entropy, x, d),
                        lambda s, sh, d: O.decode_inter(seeded_sd, oracle_entropy, s, sh, d), frames, lambda t: t,
                        lambda t: t.clamp(0, 1))
    n = len(ref)
    bpp_r = 8.0 * sum(b for b, _, _ in ref) / (n_pix * n)
    psnr_r = sum(p for _, p, _ in ref) / n
    results = {}
    import os
    try:
        for mode in ("fp32", "fp16x2", "bf16x2"):
            gpu_net.set_precision(mode)
            with torch.no_grad():
                got = _code_gop(gpu_net.encode_inter, gpu_net.decode_inter, frames, lambda t: t.cuda(),
                                lambda t: ops.clamp(t, 0.0, 1.0))
            lines, first_diff = [], None
            for i, ((bg, pg, sg), (br, pr, sr)) in enumerate(zip(got, ref), start=1):
                same = sg == sr
                if not same and first_diff is None:
                    first_diff = i
                lines.append(f"frame {i:2d}: bytes {bg:6d} vs {br:6d}, PSNR {pg:.5f} vs {pr:.5f} (d {pg - pr:+.1e}) "
                             f"{'identical streams' if same else 'streams differ'}")
            if mode == "fp32":
                print("\n".join(lines))
            bpp_g = 8.0 * sum(b for b, _, _ in got) / (n_pix * n)
            psnr_g = sum(p for _, p, _ in got) / n
            print(f"GOP {gop} at {h}x{w} [{mode}]: GPU {bpp_g:.5f} bpp {psnr_g:.4f} dB | oracle {bpp_r:.5f} bpp "
                  f"{psnr_r:.4f} dB | first differing stream: {first_diff}")
            results[mode] = (bpp_g, psnr_g, max(abs(pg - pr) for (_, pg, _), (_, pr, _) in zip(got, ref)))
    finally:
        gpu_net.set_precision(os.environ.get("DVC_TEST_PRECISION", "fp32"))
    assert 0.02 <= bpp_r <= 1.5, bpp_r
    for mode, (bpp_g, psnr_g, worst_frame) in results.items():
        assert abs(psnr_g - psnr_r) <= 1e-3, (mode, psnr_g, psnr_r)
        assert abs(bpp_g - bpp_r) <= 1e-4, (mode, bpp_g, bpp_r)
        if h * w >= 1088 * 1920:
            assert worst_frame <= 1e-3, (mode, worst_frame)
