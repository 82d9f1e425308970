"""The CPU oracle against the fixtures captured from the reference's own Python files
(tests/golden/make_golden.py): pins the oracle's graph wiring to the reference."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import dmc_oracle as O


def T(a):
    return torch.from_numpy(np.asarray(a))


def test_ops_match_reference_layers():
    g = load_golden("ops_layers.npz")
    out = O.flow_warp(T(g["warp_feat"]), T(g["warp_flow"]))
    assert torch.equal(out, T(g["warp_out"]))
    assert torch.equal(O.bilinear_up2(T(g["resample_in"])), T(g["up2_out"]))
    assert torch.equal(O.bilinear_down2(T(g["resample_in"])), T(g["down2_out"]))
    assert torch.equal(O.quantize_ste(T(g["ste_in"])), T(g["ste_out"]))
    # round half to even: .5 -> 0, 1.5 -> 2, 2.5 -> 2, -.5 -> -0
    assert O.quantize_ste(torch.tensor([0.5, 1.5, 2.5, -0.5])).tolist() == [0.0, 2.0, 2.0, -0.0]


def test_blocks_match_reference(seeded_sd):
    g = load_golden("blocks_64.npz")
    sd = seeded_sd
    with torch.no_grad():
        mv = O.me_spynet(sd, T(g["x1"]), T(g["x0"]))
        assert torch.allclose(mv, T(g["spynet_mv"]), atol=1e-5, rtol=1e-5)
        y_mv = O.motion_encoder(sd, T(g["spynet_mv"]))
        assert torch.allclose(y_mv, T(g["motion_enc"]), atol=1e-5, rtol=1e-5)
        assert torch.allclose(O.motion_decoder(sd, T(g["motion_enc"])), T(g["motion_dec"]), atol=1e-5, rtol=1e-5)
        u = O.unet(sd, "recon_generation_net.unet_1", T(g["unet_in"]))
        assert torch.allclose(u, T(g["unet_out"]), atol=1e-5, rtol=1e-5)
        fo = O.context_fusion(sd, T(g["fusion_in0"]), T(g["fusion_in1"]), T(g["fusion_in2"]))
        for i in range(3):
            assert torch.allclose(fo[i], T(g[f"fusion_out{i}"]), atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize("tag", ["synthetic", "rand"])
def test_forward_matches_reference(seeded_sd, oracle_entropy, tag):
    g = load_golden(f"forward_64_{tag}.npz")
    frames = [T(g[f"frame{i}"]) for i in range(4)]
    with torch.no_grad():
        res = O.forward(seeded_sd, oracle_entropy, frames)
    assert len(res["x_hat"]) == 3
    for i in range(3):
        assert torch.allclose(res["x_hat"][i], T(g[f"x_hat{i}"]), atol=2e-5, rtol=1e-5), i
        for a in ("motion", "frame"):
            for b in ("y", "z"):
                ref = T(g[f"lik{i}_{a}_{b}"])
                got = res["likelihoods"][i][a][b]
                assert got.shape == ref.shape
                # bits agree to well below the 1e-4 bpp budget
                bits_ref = -torch.log2(ref).sum().item()
                bits_got = -torch.log2(got).sum().item()
                assert abs(bits_ref - bits_got) <= 1e-5 * (64 * 64) + 1e-6 * abs(bits_ref), (i, a, b)


def test_forward_rejects_non_list(seeded_sd, oracle_entropy):
    with pytest.raises(RuntimeError):
        O.forward(seeded_sd, oracle_entropy, torch.zeros(1, 3, 64, 64))


def test_codec_loop_matches_reference(seeded_sd, oracle_entropy):
    """encode_inter -> decode_inter -> clamp -> dpb update, two P frames (dmc/test.py:186-195)."""
    g = load_golden("codec_64.npz")
    sd, ent = seeded_sd, oracle_entropy
    dpb = {"x_ref": T(g["frame0"]), "feature_ref": None, "y_ref": None, "y_mv_ref": None}
    with torch.no_grad():
        for i in (1, 2):
            enc = O.encode_inter(sd, ent, T(g[f"frame{i}"]), dpb)
            for key in ("motion", "frame"):
                assert tuple(enc["shape"][key]) == tuple(g[f"shape{i}_{key}"])
                for j in range(3):
                    assert enc["strings"][key][j][0] == g[f"bytes{i}_{key}_{j}"].tobytes(), (i, key, j)
            rec, info = O.decode_inter(sd, ent, enc["strings"], enc["shape"], dpb)
            rec = rec.clamp(0, 1)
            assert torch.allclose(rec, T(g[f"rec{i}"]), atol=2e-5)
            assert torch.allclose(info["y_ref"], T(g[f"y_ref{i}"]), atol=1e-5)
            assert torch.allclose(info["y_mv_ref"], T(g[f"y_mv_ref{i}"]), atol=1e-5)
            dpb = {"x_ref": rec, "feature_ref": info["feature_ref"], "y_ref": info["y_ref"],
                   "y_mv_ref": info["y_mv_ref"]}


def test_decode_reproduces_forward(seeded_sd, oracle_entropy):
    """decode_inter(encode_inter(x)) gives the x_hat of forward() in eval mode."""
    from deepvideocodec_amd.testing import synthetic_gop
    frames = synthetic_gop(2, 64, 64, seed=21)
    sd, ent = seeded_sd, oracle_entropy
    dpb = {"x_ref": frames[0], "feature_ref": None, "y_ref": None, "y_mv_ref": None}
    with torch.no_grad():
        fwd = O.forward(sd, ent, frames)
        enc = O.encode_inter(sd, ent, frames[1], dpb)
        rec, _ = O.decode_inter(sd, ent, enc["strings"], enc["shape"], dpb)
    assert torch.allclose(rec, fwd["x_hat"][0], atol=1e-5)


def test_remaining_blocks_match_reference(seeded_sd):
    """Feature extractor (intra and inter adaptor), motion compensation, contextual encoder / decoder,
    reconstruction net, hyper encoder / decoder, prior fusion, temporal prior and the reference's own
    process_with_mask / forward_dual_prior (both modes), block by block (tests/golden/blocks2_32.npz)."""
    g = {k: T(v) for k, v in load_golden("blocks2_32.npz").items()}
    sd = seeded_sd
    ok = lambda a, b: torch.allclose(a, b, atol=1e-5, rtol=1e-5)
    with torch.no_grad():
        for tag, dpb in (("I", {"x_ref": g["x_ref"], "feature_ref": None}),
                         ("P", {"x_ref": g["x_ref"], "feature_ref": g["feature_ref"]})):
            feat = O.conv(sd, "feature_adaptor_I", dpb["x_ref"]) if tag == "I" else O.conv(sd, "feature_adaptor_P", dpb["feature_ref"])
            for i, f in enumerate(O.feature_extractor(sd, feat)):
                assert ok(f, g[f"extractor_{tag}{i}"]), (tag, i)
            c1, c2, c3, warp = O.motion_compensation(sd, g["mv"], dpb)
            assert ok(c1, g[f"mc_{tag}_c1"]) and ok(c2, g[f"mc_{tag}_c2"]) and ok(c3, g[f"mc_{tag}_c3"])
            assert ok(warp, g[f"mc_{tag}_warp"])
        assert ok(O.contextual_encoder(sd, g["x_cur"], g["ctx1"], g["ctx2"], g["ctx3"]), g["ctx_enc_y"])
        assert ok(O.contextual_decoder(sd, g["ctx_dec_in"], g["ctx2"], g["ctx3"]), g["ctx_dec_out"])
        # the reference calls recon_generation_net(x_rec_feature, context1): video_model.py:575
        feature, recon = O.recon_generation(sd, g["ctx_dec_out"], g["ctx1"])
        assert ok(feature, g["recon_feature"]) and ok(recon, g["recon_x"])
        for name, cy in (("motion", 64), ("frame", 96)):
            key = name + "_context_model"
            assert ok(O.hyper_encoder(sd, key + ".hyper_encoder", g[f"{name}_y"]), g[f"{name}_hyper_enc"])
            assert ok(O.hyper_decoder(sd, key + ".hyper_decoder", g[f"{name}_z_hat"]), g[f"{name}_hyper_dec"])
            means, scales = O.context_model_params(sd, key, g[f"{name}_y"], g[f"{name}_y_ref"], g[f"{name}_z_hat"],
                                                   g.get("frame_temporal_in"))
            fused = g[f"{name}_prior_fusion"]
            assert ok(torch.cat((means, scales), 1), fused)
            means, scales = fused.chunk(2, 1)
            m0, m1 = O.checkerboard_masks(4, 4)
            assert torch.equal(m0, g[f"{name}_mask0"]) and torch.equal(m1, g[f"{name}_mask1"])
            y = g[f"{name}_y"]
            for j, t in enumerate(O.process_with_mask(y[:, :cy // 2], means[:, :cy // 2], scales[:, :cy // 2], m1)):
                assert torch.equal(t + 0.0, g[f"{name}_pwm{j}"] + 0.0), (name, j)
            for j, t in enumerate(O.dual_prior(sd, key, y, means, scales)):
                assert ok(t, g[f"{name}_fdp{j}"]), (name, j)
            for j, t in enumerate(O.dual_prior(sd, key, y, means, scales, compress=True)):
                assert ok(t, g[f"{name}_fdpc{j}"]), (name, j)
