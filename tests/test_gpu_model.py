"""Whole-path parity on the GPU: the product DMC (HIP kernels through the C ABI) against
(1) the fixtures captured from the reference's own graph and (2) the CPU oracle, with the
north-star tolerances: |dPSNR| <= 1e-3 dB and |dbpp| <= 1e-4 on the same inputs; plus
size-independent properties at the full 1088x1920 size."""
import math

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import dmc_oracle as O

pytestmark = pytest.mark.gpu

PSNR_TOL = 1e-3
BPP_TOL = 1e-4


@pytest.fixture(autouse=True)
def _inference_mode():
    """These are the evaluation-loop tests: like dmc/test.py they run under torch.no_grad()."""
    with torch.no_grad():
        yield


def T(a):
    return torch.from_numpy(np.asarray(a))


def psnr(x, rec):
    return -10.0 * math.log10(torch.mean((x.double() - rec.double()) ** 2).item())


def bits(lik):
    return -torch.log2(lik.double()).sum().item()


@pytest.mark.parametrize("tag", ["synthetic", "rand"])
def test_forward_matches_reference_fixture(gpu_net, tag):
    g = load_golden(f"forward_64_{tag}.npz")
    frames = [T(g[f"frame{i}"]) for i in range(4)]
    res = gpu_net([f.cuda() for f in frames])
    assert set(res) == {"x_hat", "likelihoods"} and len(res["x_hat"]) == 3
    n_pix = 64 * 64
    for i in range(3):
        ref = T(g[f"x_hat{i}"])
        got = res["x_hat"][i].cpu()
        assert abs(psnr(frames[i + 1], got) - psnr(frames[i + 1], ref)) <= PSNR_TOL, i
        assert (got - ref).abs().max().item() < 1e-4 * max(1.0, ref.abs().max().item()), i
        d_bits = 0.0
        for a in ("motion", "frame"):
            for b in ("y", "z"):
                lk = res["likelihoods"][i][a][b]
                assert tuple(lk.shape) == tuple(g[f"lik{i}_{a}_{b}"].shape)
                d_bits += bits(lk.cpu()) - bits(T(g[f"lik{i}_{a}_{b}"]))
        assert abs(d_bits) / n_pix <= BPP_TOL, (i, d_bits)


def test_codec_loop_matches_reference_fixture(gpu_net):
    """encode_inter -> decode_inter -> clamp -> dpb update (dmc/test.py:186-195)."""
    g = load_golden("codec_64.npz")
    dpb = {"x_ref": T(g["frame0"]).cuda(), "feature_ref": None, "y_ref": None, "y_mv_ref": None}
    total_ref = total_got = 0
    for i in (1, 2):
        x = T(g[f"frame{i}"])
        enc = gpu_net.encode_inter(x.cuda(), dpb)
        assert set(enc) == {"strings", "shape"}
        for key in ("motion", "frame"):
            assert tuple(enc["shape"][key]) == tuple(g[f"shape{i}_{key}"])
            assert len(enc["strings"][key]) == 3
            for j in range(3):
                s = enc["strings"][key][j]
                assert isinstance(s, list) and len(s) == 1 and isinstance(s[0], bytes)
                total_got += len(s[0])
                total_ref += g[f"bytes{i}_{key}_{j}"].size
                # identical symbols => identical bytes (integer work is bit-exact)
                assert s[0] == g[f"bytes{i}_{key}_{j}"].tobytes(), (i, key, j)
        rec, info = gpu_net.decode_inter(enc["strings"], enc["shape"], dpb)
        assert set(info) == {"x_ref", "feature_ref", "y_ref", "y_mv_ref"}
        rec = rec.clamp(0, 1)
        ref = T(g[f"rec{i}"])
        assert abs(psnr(x, rec.cpu()) - psnr(x, ref)) <= PSNR_TOL
        assert (info["y_ref"].cpu() - T(g[f"y_ref{i}"])).abs().max().item() < 1e-4
        assert (info["y_mv_ref"].cpu() - T(g[f"y_mv_ref{i}"])).abs().max().item() < 1e-4
        dpb = {"x_ref": rec, "feature_ref": info["feature_ref"], "y_ref": info["y_ref"],
               "y_mv_ref": info["y_mv_ref"]}
    assert total_got == total_ref


def _near_tie(pre, tol=2e-3):
    """Elements whose pre-rounding value sits within tol of a rounding boundary (x.5)."""
    frac = pre - torch.floor(pre)
    return (frac - 0.5).abs() <= tol * pre.abs().clamp_min(1.0)


def _explained_by_ties(got_hat, trace, got_z_lik=None, ref_z_lik=None):
    """A mismatch in quantised latents is legitimate only if it starts at a genuine fp32 rounding tie.

    Hyper-latents: the z symbols that differ (seen through their likelihoods: another symbol has another
    probability) must THEMSELVES be near-ties in the oracle's pre-rounding values; a differing z element
    moves every mean / scale in its neighbourhood, so it explains the y differences of the frame.
    Latents (no z difference): at least one differing y element is a near-tie, and every other differing
    element lies within the 7x7 receptive field of the spatial prior (three 3x3 layers) around such a
    tie -- the second checkerboard pass sees the flipped first-pass value.  Returns (ok, n_diff)."""
    # y_hat = round(y - mu) + mu: equal symbols give values equal to ~1e-5, a flipped symbol moves the
    # value by a whole step
    diff = (got_hat - trace["y_hat"]).abs() > 0.5
    n_diff = int(diff.sum())
    if got_z_lik is not None:
        z_diff = (got_z_lik - ref_z_lik).abs() > 1e-3 * ref_z_lik.abs().clamp_min(1e-6)
        if bool(z_diff.any()):
            ok = int(z_diff.sum()) <= 4 and bool(_near_tie(trace["z_pre"])[z_diff].all())
            return ok, max(n_diff, int(z_diff.sum()))
    if n_diff == 0:
        return True, 0
    ties = diff & _near_tie(trace["y_pre"])
    if not bool(ties.any()) or int(ties.sum()) > 8:
        return False, n_diff
    near = torch.nn.functional.max_pool2d(ties.any(dim=1, keepdim=True).float(), 7, stride=1, padding=3) > 0
    return bool((diff.any(dim=1, keepdim=True) <= near).all()), n_diff


def test_forward_vs_oracle_128_batch2(gpu_net, seeded_sd, oracle_entropy):
    """A different size and batch 2 against the oracle run on the CPU in the same test, frame by
    frame from the oracle's decoded-picture buffer.  round() sits in the middle of the graph, so two
    fp32 implementations with different summation orders may legitimately disagree on a latent that
    lies within ~1e-6 of x.5; such a frame is accepted only if every disagreement is traced to a
    genuine tie in the oracle's own pre-rounding values, and the strict PSNR / bpp tolerances are
    asserted on all other frames."""
    from deepvideocodec_amd.testing import synthetic_gop
    frames = synthetic_gop(4, 128, 192, seed=7, batch=2)
    n_pix = 128 * 192
    dpb = {"x_ref": frames[0], "feature_ref": None, "y_ref": None, "y_mv_ref": None}
    strict = flips = 0
    for i in range(1, 4):
        trace = {}
        with torch.no_grad():
            x_o, lik_o, ctx_o = O.forward_inter(seeded_sd, oracle_entropy, frames[i], dpb, trace=trace)
        dpb_g = {k: (None if v is None else v.cuda()) for k, v in dpb.items()}
        x_g, lik_g, ctx_g = gpu_net.forward_inter(frames[i].cuda(), dpb_g)
        for b in range(2):
            sl = slice(b, b + 1)
            tr_m = {k: v[sl] for k, v in trace["motion_context_model"].items()}
            tr_f = {k: v[sl] for k, v in trace["frame_context_model"].items()}
            ok_m, n_m = _explained_by_ties(ctx_g["y_mv_ref"][sl].cpu(), tr_m, lik_g["motion"]["z"][sl].cpu(),
                                           lik_o["motion"]["z"][sl])
            assert ok_m, f"frame {i} batch {b}: {n_m} motion latents differ without a rounding tie"
            if n_m:
                flips += 1
                continue
            ok_f, n_f = _explained_by_ties(ctx_g["y_ref"][sl].cpu(), tr_f, lik_g["frame"]["z"][sl].cpu(),
                                           lik_o["frame"]["z"][sl])
            assert ok_f, f"frame {i} batch {b}: {n_f} frame latents differ without a rounding tie"
            if n_f:
                flips += 1
                continue
            strict += 1
            assert abs(psnr(frames[i][b], x_o[b]) - psnr(frames[i][b], x_g[b].cpu())) <= PSNR_TOL, (i, b)
            bw = sum(bits(t[sl]) for v in lik_o.values() for t in v.values()) / n_pix
            bg = sum(bits(t[sl].cpu()) for v in lik_g.values() for t in v.values()) / n_pix
            assert abs(bw - bg) <= BPP_TOL, (i, b, bw, bg)
        dpb = dict(ctx_o)
    assert strict >= 4, f"only {strict} of 6 frame/batch pairs were free of rounding ties ({flips} ties)"


def test_codec_batch2_equals_per_sample_coding(gpu_net):
    """encode_inter / decode_inter on a batch of two sequences: each sample's strings and reconstruction
    equal, bit for bit, what coding that sequence alone gives (the strings are per sample, as compressai's
    compress returns them; nothing in the kernels mixes samples)."""
    from deepvideocodec_amd import ops
    from deepvideocodec_amd.testing import synthetic_gop
    frames = [f.cuda() for f in synthetic_gop(3, 128, 192, seed=7, batch=2)]

    def code(fr):
        dpb = {"x_ref": fr[0], "feature_ref": None, "y_ref": None, "y_mv_ref": None}
        out = []
        for x in fr[1:]:
            enc = gpu_net.encode_inter(x, dpb)
            rec, info = gpu_net.decode_inter(enc["strings"], enc["shape"], dpb)
            out.append((enc["strings"], rec))
            dpb = {"x_ref": ops.clamp(rec, 0.0, 1.0), "feature_ref": info["feature_ref"], "y_ref": info["y_ref"],
                   "y_mv_ref": info["y_mv_ref"]}
        return out

    both = code(frames)
    for b in range(2):
        alone = code([f[b:b + 1].contiguous() for f in frames])
        for i, ((s2, r2), (s1, r1)) in enumerate(zip(both, alone), start=1):
            for key in s2:
                assert len(s2[key]) == len(s1[key])
                for part2, part1 in zip(s2[key], s1[key]):
                    assert len(part2) == 2 and part2[b] == part1[0], f"sample {b} frame {i}: {key} strings differ"
            assert torch.equal(r2[b:b + 1], r1), f"sample {b} frame {i}: reconstructions differ"


def test_decode_reproduces_encoder_state(gpu_net):
    """The decoder rebuilds exactly the latents the encoder quantised (bit-exact), and
    decode(encode(x)) equals forward()'s reconstruction in eval mode."""
    from deepvideocodec_amd.testing import synthetic_gop
    frames = [f.cuda() for f in synthetic_gop(2, 128, 128, seed=9)]
    dpb = {"x_ref": frames[0], "feature_ref": None, "y_ref": None, "y_mv_ref": None}
    fwd = gpu_net(frames)
    enc = gpu_net.encode_inter(frames[1], dpb)
    rec, info = gpu_net.decode_inter(enc["strings"], enc["shape"], dpb)
    assert torch.equal(rec, fwd["x_hat"][0])
    x_rec, lik, ctx = gpu_net.forward_inter(frames[1], dpb)
    assert torch.equal(info["y_ref"], ctx["y_ref"]) and torch.equal(info["y_mv_ref"], ctx["y_mv_ref"])
    assert torch.equal(info["feature_ref"], ctx["feature_ref"])
    # actual bytes are close to the likelihood estimate
    est = sum(bits(t.cpu()) for v in lik.values() for t in v.values())
    act = 8 * sum(len(s[0]) for v in enc["strings"].values() for s in v)
    # six streams, each flushed with a 32-bit word granularity and an 8-byte final state: a fixed ~6 x 64 bits
    # on top of the 2 % the quantised CDF tables may cost at this small size (a few kbit in all)
    assert abs(act - est) <= 0.02 * est + 6 * 64, (act, est)


def test_motion_pretrain_and_training_mode(gpu_net):
    from deepvideocodec_amd.testing import synthetic_gop
    frames = [f.cuda() for f in synthetic_gop(3, 64, 64, seed=13)]
    out = gpu_net(frames, motion_pretrain=True)
    assert len(out["x_hat"]) == 2 and set(out["likelihoods"][0]) == {"motion"}
    gpu_net.train()
    try:
        torch.manual_seed(0)
        a = gpu_net(frames)
        torch.manual_seed(1)
        b = gpu_net(frames)
    finally:
        gpu_net.eval()
    # decoder path is noise-free (round-STE), the rate is computed on noisy latents
    assert torch.equal(a["x_hat"][0], b["x_hat"][0])
    assert not torch.equal(a["likelihoods"][0]["frame"]["y"], b["likelihoods"][0]["frame"]["y"])


class _TableCost:
    """Records the (symbols, indexes) a context model hands to the range coder and prices them with the
    integer CDF tables the coder uses: sum of -log2(frequency / 2^16), i.e. what an ideal arithmetic coder
    spends with exactly these tables.  The actual stream may exceed it only by the coder's inefficiency."""

    def __init__(self, model):
        self._parts = []      # (bits, escapes, symbols) per coder call; calls come from several host threads
        gc, eb = model.gaussian_conditional, model.entropy_bottleneck
        self._undo = []
        for mod, with_idx in ((gc, True), (eb, False)):
            orig = mod.compress_symbols
            if with_idx:
                def wrapped(sym, idx, _orig=orig, _mod=mod):
                    self._price(_mod, sym, idx)
                    return _orig(sym, idx)
            else:
                def wrapped(sym, _orig=orig, _mod=mod):
                    n, c, h, w = sym.shape
                    self._price(_mod, sym, np.broadcast_to(np.arange(c).reshape(1, c, 1, 1), sym.shape))
                    return _orig(sym)
            mod.compress_symbols = wrapped
            self._undo.append(mod)

    def _price(self, mod, sym, idx):
        cdf = mod._quantized_cdf.cpu().numpy().astype(np.int64)
        length = mod._cdf_length.cpu().numpy().astype(np.int64)
        offset = mod._offset.cpu().numpy().astype(np.int64)
        s = np.asarray(sym, dtype=np.int64).reshape(-1)
        k = np.asarray(idx, dtype=np.int64).reshape(-1)
        v = s - offset[k]
        last = length[k] - 2                       # table position of the escape entry
        esc = (v < 0) | (v >= last)
        v = np.where(esc, last, v)
        freq = cdf[k, v + 1] - cdf[k, v]
        assert (freq >= 1).all()
        # an escaped value is followed by its magnitude in 4-bit groups (a few groups for |v| < 4096)
        self._parts.append((float((16.0 - np.log2(freq)).sum()) + 16.0 * int(esc.sum()), int(esc.sum()), s.size))

    bits = property(lambda self: sum(p[0] for p in self._parts))
    escapes = property(lambda self: sum(p[1] for p in self._parts))
    symbols = property(lambda self: sum(p[2] for p in self._parts))

    def close(self):
        for mod in self._undo:
            del mod.compress_symbols


def test_full_size_round_trip_1080p(gpu_net):
    """BASELINE config 2 size (1080p padded to 1088x1920), two P frames: size-independent
    properties -- decode(encode(x)) == forward(x), the dpb recurrence stays consistent,
    reconstruction is finite, and the coded size is pinned from both sides:
      * against the integer CDF tables (the bits an ideal coder spends with the tables the range coder
        uses): the actual streams are within 0.2 % (measured +0.04 %) -- this pins the coder and the tables;
      * against the likelihood estimate (continuous scales, dmc/train.py:74-93): within 3 %.  The
        streams run 0.5-2.5 % above it because the coder replaces every scale by the next larger of the
        64 table entries (base_model.py:43-49, a 13 % grid) and most of the bits sit in latents whose
        cost is steep in the scale."""
    from deepvideocodec_amd.testing import pad_to_multiple, synthetic_gop
    frames = [pad_to_multiple(f, 64).cuda() for f in synthetic_gop(3, 1080, 1920, seed=1)]
    assert tuple(frames[0].shape) == (1, 3, 1088, 1920)
    dpb = {"x_ref": frames[0], "feature_ref": None, "y_ref": None, "y_mv_ref": None}
    fwd = gpu_net(frames)
    for i in (1, 2):
        costs = [_TableCost(gpu_net.motion_context_model), _TableCost(gpu_net.frame_context_model)]
        try:
            enc = gpu_net.encode_inter(frames[i], dpb)
        finally:
            for c in costs:
                c.close()
        assert tuple(enc["shape"]["motion"]) == (17, 30) and tuple(enc["shape"]["frame"]) == (17, 30)
        rec, info = gpu_net.decode_inter(enc["strings"], enc["shape"], dpb)
        assert torch.isfinite(rec).all()
        assert torch.equal(rec, fwd["x_hat"][i - 1])
        est = sum(bits(t.cpu()) for v in fwd["likelihoods"][i - 1].values() for t in v.values())
        act = 8 * sum(len(s[0]) for v in enc["strings"].values() for s in v)
        table = sum(c.bits for c in costs)
        n_sym, n_esc = sum(c.symbols for c in costs), sum(c.escapes for c in costs)
        print(f"frame {i}: {act} bits written, {table:.0f} from the CDF tables ({act / table - 1:+.3%}), "
              f"{est:.0f} from the likelihoods ({act / est - 1:+.3%}); {n_esc} of {n_sym} symbols escape-coded")
        assert n_sym == 160 * 68 * 120 + 2 * 64 * 17 * 30
        assert n_esc <= 1e-4 * n_sym, (n_esc, n_sym)
        assert abs(act - table) <= 0.002 * table + 6 * 64, (i, act, table)
        assert abs(act - est) <= 0.03 * est, (i, act, est, act / est)
        # forward() chains on the un-clamped reconstruction (reference: video_model.py:544-549)
        dpb = {"x_ref": rec, "feature_ref": info["feature_ref"], "y_ref": info["y_ref"],
               "y_mv_ref": info["y_mv_ref"]}


def test_evaluation_loop_bitstream_round_trip(gpu_net, seeded_sd, oracle_entropy):
    """dmc/test.py's per-sequence loop: 5 frames at 70x100 (padded to 128x128), GOP = 3.  The
    written bitstream decodes on its own to the encoder's reconstructions; bpp is the file size
    over the unpadded pixel count; the first GOP's bytes equal the CPU oracle's for the same loop
    (unless a rounding tie flips a symbol, in which case sizes must still agree to 2 %)."""
    import io
    from deepvideocodec_amd import bitstream, evaluate, ops, sharding
    from deepvideocodec_amd.testing import synthetic_gop
    from oracle import dmc_oracle as O
    frames_cpu = synthetic_gop(5, 70, 100, seed=21)
    frames = [f.cuda() for f in frames_cpu]
    stats, segs = evaluate.evaluate_sequence(gpu_net, frames, gop=3, name="clip", quality=6)
    assert [g for g, _ in segs] == [0, 1] and list(stats) == ["video", "quality", "bpp", "psnr"]
    total = sum(len(b) for _, b in segs)
    # frames 0 and 3 are the (uncoded) intra pictures of the two GOPs: three coded frames
    assert stats["bpp"] == pytest.approx(total * 8 / (70 * 100 * 3))
    assert math.isfinite(stats["psnr"]) and 0 < stats["psnr"] < 100
    clamp = lambda t: ops.clamp(t, 0.0, 1.0)
    # stand-alone decode of GOP 0 from its bytes reproduces the encoder-side loop
    fd = io.BytesIO(segs[0][1])
    recs = evaluate.decode_gop(gpu_net, fd, evaluate.pad(frames[0]), 3, clamp)
    assert fd.read() == b"" and len(recs) == 2
    fd2 = io.BytesIO()
    nbytes, sum_psnr, counted = evaluate.code_gop(gpu_net, frames[0:3], fd2, evaluate.RawIntra(), clamp)
    assert fd2.getvalue() == segs[0][1] and nbytes == len(segs[0][1])
    assert counted == 2 and math.isfinite(sum_psnr)   # the raw intra frame (zero error) is not counted
    for i, r in enumerate(recs):
        assert 0 < psnr(frames_cpu[i + 1], evaluate.crop(r, (70, 100)).cpu()) < 100
    # the oracle through the same loop
    class Net:
        def encode_inter(self, x, dpb):
            return O.encode_inter(seeded_sd, oracle_entropy, x, dpb)

        def decode_inter(self, s, sh, dpb):
            return O.decode_inter(seeded_sd, oracle_entropy, s, sh, dpb)
    fd3 = io.BytesIO()
    with torch.no_grad():
        evaluate.code_gop(Net(), frames_cpu[0:3], fd3, evaluate.RawIntra(), lambda t: t.clamp(0, 1))
    a, b = fd3.getvalue(), segs[0][1]
    assert a == b or abs(len(a) - len(b)) <= 0.02 * len(a)


def test_encoder_side_reuse_equals_separate_decode(gpu_net):
    """encode_inter_with_recon returns the same strings as encode_inter and exactly the
    reconstruction / buffer entries decode_inter rebuilds from those strings, over a 3-frame chain."""
    from deepvideocodec_amd.testing import synthetic_gop
    frames = [f.cuda() for f in synthetic_gop(4, 128, 192, seed=17)]
    dpb = {"x_ref": frames[0], "feature_ref": None, "y_ref": None, "y_mv_ref": None}
    for x in frames[1:]:
        enc = gpu_net.encode_inter(x, dpb)
        rec, info = gpu_net.decode_inter(enc["strings"], enc["shape"], dpb)
        enc2, rec2, info2 = gpu_net.encode_inter_with_recon(x, dpb)
        assert enc2["strings"] == enc["strings"] and enc2["shape"] == enc["shape"]
        assert torch.equal(rec2, rec)
        for k in ("feature_ref", "y_ref", "y_mv_ref", "x_ref"):
            assert torch.equal(info2[k], info[k]), k
        dpb = {"x_ref": rec.clamp(0, 1), "feature_ref": info["feature_ref"], "y_ref": info["y_ref"],
               "y_mv_ref": info["y_mv_ref"]}


@pytest.mark.timeout(600)
def test_config0_256x256_gop4_random_frames_vs_oracle(gpu_net, seeded_sd, oracle_entropy):
    """BASELINE configs[0]: one 256x256 GOP of 4 uniformly random frames, forward + likelihood bpp,
    HIP path against the CPU oracle frame by frame (teacher-forced from the oracle's buffer; a frame
    whose only differences are genuine rounding ties is counted, not compared strictly)."""
    from deepvideocodec_amd.testing import random_gop
    frames = random_gop(4, 256, 256, seed=0)
    n_pix = 256 * 256
    dpb = {"x_ref": frames[0], "feature_ref": None, "y_ref": None, "y_mv_ref": None}
    strict = 0
    torch.set_num_threads(8)
    for i in range(1, 4):
        trace = {}
        x_o, lik_o, ctx_o = O.forward_inter(seeded_sd, oracle_entropy, frames[i], dpb, trace=trace)
        dpb_g = {k: (None if v is None else v.cuda()) for k, v in dpb.items()}
        x_g, lik_g, ctx_g = gpu_net.forward_inter(frames[i].cuda(), dpb_g)
        ok_m, n_m = _explained_by_ties(ctx_g["y_mv_ref"].cpu(), trace["motion_context_model"],
                                       lik_g["motion"]["z"].cpu(), lik_o["motion"]["z"])
        assert ok_m, f"frame {i}: {n_m} motion latents differ without a rounding tie"
        ok_f, n_f = (True, 0) if n_m else _explained_by_ties(ctx_g["y_ref"].cpu(), trace["frame_context_model"],
                                                             lik_g["frame"]["z"].cpu(), lik_o["frame"]["z"])
        assert ok_f, f"frame {i}: {n_f} frame latents differ without a rounding tie"
        if n_m == 0 and n_f == 0:
            strict += 1
            assert abs(psnr(frames[i], x_o) - psnr(frames[i], x_g.cpu())) <= PSNR_TOL, i
            bw = sum(bits(t) for v in lik_o.values() for t in v.values()) / n_pix
            bg = sum(bits(t.cpu()) for v in lik_g.values() for t in v.values()) / n_pix
            assert abs(bw - bg) <= BPP_TOL, (i, bw, bg)
        dpb = dict(ctx_o)
    assert strict >= 2, f"only {strict} of 3 frames were free of rounding ties"


def test_bf16_precision_option(seeded_sd):
    """The reduced-precision option: encoder and decoder stay consistent (the stream decodes to the
    encoder's reconstruction), results stay close to the fp32 path, and switching back restores it."""
    from deepvideocodec_amd.models import DMC
    from deepvideocodec_amd.testing import synthetic_gop
    net = DMC()
    net.load_state_dict(seeded_sd)
    net = net.cuda().eval()
    net.update(force=True)
    frames = [f.cuda() for f in synthetic_gop(3, 128, 192, seed=23)]
    n_pix = 128 * 192

    def run():
        dpb = {"x_ref": frames[0], "feature_ref": None, "y_ref": None, "y_mv_ref": None}
        nbytes, recs = 0, []
        for x in frames[1:]:
            enc, rec_e, _ = net.encode_inter_with_recon(x, dpb)
            rec, info = net.decode_inter(enc["strings"], enc["shape"], dpb)
            assert torch.equal(rec, rec_e)                     # decoder == encoder-side reconstruction
            nbytes += sum(len(s[0]) for v in enc["strings"].values() for s in v)
            recs.append(rec)
            dpb = {"x_ref": rec.clamp(0, 1), "feature_ref": info["feature_ref"], "y_ref": info["y_ref"],
                   "y_mv_ref": info["y_mv_ref"]}
        return nbytes * 8 / (2 * n_pix), [psnr(x.cpu(), r.cpu()) for x, r in zip(frames[1:], recs)], recs

    bpp32, psnr32, rec32 = run()
    net.set_precision("bf16")
    # every convolution but the flow / RGB heads (<= 4 output channels) takes the mode
    assert net.contextual_encoder.conv1._precision() and net.optic_flow.moduleBasic[0].conv1._precision()
    assert not net.optic_flow.moduleBasic[0].conv5._precision() and not net.recon_generation_net.recon_conv._precision()
    bpp16, psnr16, rec16 = run()
    assert not torch.equal(rec16[0], rec32[0])                 # it really is a different arithmetic
    assert abs(bpp16 - bpp32) <= 0.03 * bpp32, (bpp16, bpp32)
    assert all(abs(a - b) <= 0.1 for a, b in zip(psnr16, psnr32)), (psnr16, psnr32)
    net.set_precision("bf16x2")                                 # split operands: ~2^-17 per product
    assert net.contextual_encoder.conv1._precision() == 2
    bpp2, psnr2, rec2 = run()
    assert not torch.equal(rec2[0], rec32[0])
    assert abs(bpp2 - bpp32) <= 2e-3 * bpp32, (bpp2, bpp32)
    assert all(abs(a - b) <= 2e-3 for a, b in zip(psnr2, psnr32)), (psnr2, psnr32)
    net.set_precision("fp16x2")                                 # fp16 parts: ~2^-21 per product
    assert net.contextual_encoder.conv1._precision() == 3
    bpp3, psnr3, rec3 = run()
    assert abs(bpp3 - bpp32) <= 2e-4 * bpp32, (bpp3, bpp32)
    assert all(abs(a - b) <= 2e-4 for a, b in zip(psnr3, psnr32)), (psnr3, psnr32)
    net.set_precision("fp32")
    bpp_again, _, rec_again = run()
    assert bpp_again == bpp32 and torch.equal(rec_again[1], rec32[1])
    with pytest.raises(ValueError):
        net.set_precision("fp8")
