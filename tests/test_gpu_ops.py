"""Per-kernel parity: every C-ABI device entry point (through deepvideocodec_amd.ops, which
only marshals pointers) against the CPU oracle / plain fp32 PyTorch on the same seeded
inputs.  Floating-point tolerance: 1e-5 relative to the output's magnitude (both sides are
fp32 with different summation orders); integer outputs (symbols, indexes) must be exact."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import dmc_oracle as O
from oracle import entropy_oracle as E

pytestmark = pytest.mark.gpu


def dev(t):
    return t.to("cuda:0")


def close(got, want, rel=2e-5, abs_=1e-6):
    got = got.detach().cpu()
    tol = rel * max(1.0, float(want.abs().max())) + abs_
    err = float((got - want).abs().max())
    assert got.shape == want.shape, (got.shape, want.shape)
    assert err <= tol, f"max err {err:.3e} > tol {tol:.3e}"


CONV_CASES = [
    # k, stride, cins (concat parts), cout, H, W, batch
    (3, 1, (64,), 64, 37, 45, 1),
    (3, 1, (3, 64), 64, 32, 64, 2),
    (3, 2, (67,), 64, 48, 80, 1),
    (3, 2, (2,), 64, 64, 64, 1),
    (3, 1, (96, 96, 96), 288, 17, 30, 1),
    (3, 1, (144,), 192, 8, 12, 2),
    (3, 1, (64,), 3, 40, 40, 1),
    (3, 1, (32,), 32, 70, 33, 1),
    (7, 1, (8,), 32, 30, 50, 1),
    (7, 1, (32,), 64, 24, 40, 2),
    (7, 1, (16,), 2, 45, 37, 1),
    (1, 1, (64,), 64, 33, 47, 1),
    (1, 2, (2,), 64, 32, 48, 1),
    (1, 2, (64,), 64, 34, 50, 2),
    (1, 1, (32, 32), 64, 64, 64, 1),
    (1, 1, (64,), 64, 32, 47, 1),       # H*W = 47 rows of 32: the 1x1 kernel's rows of 32 cross the image's rows
    (1, 1, (16, 48), 32, 24, 20, 2),    # 15 rows of 32: a partial tile
    (3, 1, (5, 7, 9, 11), 40, 21, 19, 1),
]


@pytest.mark.parametrize("k,stride,cins,cout,H,W,batch", CONV_CASES)
def test_conv2d_plain(k, stride, cins, cout, H, W, batch):
    from deepvideocodec_amd import ops
    g = torch.Generator().manual_seed(k * 1000 + cout + H)
    parts = [torch.randn(batch, c, H, W, generator=g) for c in cins]
    cin = sum(cins)
    w = torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)
    b = torch.randn(cout, generator=g)
    want = F.conv2d(torch.cat(parts, 1), w, b, stride=stride, padding=k // 2)
    got = ops.conv2d([dev(p) for p in parts], dev(w), dev(b), stride=stride)
    close(got, want)


def test_conv2d_big_tiles_and_slices():
    """Large enough for the 16x32-pixel tile variant; sources / residual / destination are
    channel-slice views of larger buffers."""
    from deepvideocodec_amd import ops
    g = torch.Generator().manual_seed(5)
    big = torch.randn(1, 96, 272, 480, generator=g)
    x = big[:, 16:80]                       # 64-channel view
    w = torch.randn(64, 64, 3, 3, generator=g) / 24
    b = torch.randn(64, generator=g)
    res_big = torch.randn(1, 128, 272, 480, generator=g)
    res = res_big[:, 64:]
    want = F.leaky_relu(F.conv2d(F.leaky_relu(x, 0.01), w, b, padding=1), 0.1) + res
    dbig = dev(big)
    out_big = torch.zeros(1, 128, 272, 480, device="cuda:0")
    got = ops.conv2d(dbig[:, 16:80], dev(w), dev(b), in_slope=0.01, out_slope=0.1, residual=dev(res_big)[:, 64:],
                     out=out_big[:, :64])
    close(got, want)
    assert float(out_big[:, 64:].abs().max()) == 0.0   # nothing written outside the slice


def test_conv2d_epilogues():
    from deepvideocodec_amd import ops
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 32, 20, 28, generator=g)
    w = torch.randn(64, 32, 3, 3, generator=g) / 17
    b = torch.randn(64, generator=g)
    r1 = torch.randn(2, 64, 20, 28, generator=g)
    r2 = torch.randn(2, 64, 20, 28, generator=g)
    sc = torch.rand(2, 64, generator=g)
    base = F.conv2d(x, w, b, padding=1)
    # ReLU + scaled residual + second residual
    want = (F.relu(base) + r1 * sc[:, :, None, None]) + r2
    got = ops.conv2d(dev(x), dev(w), dev(b), out_slope=0.0, residual=dev(r1), res_scale=dev(sc).reshape(-1),
                     residual2=dev(r2))
    close(got, want)
    # PixelShuffle(2) store with activation
    want = F.pixel_shuffle(F.leaky_relu(base, 0.01), 2)
    got = ops.conv2d(dev(x), dev(w), dev(b), out_slope=0.01, pixel_shuffle=True)
    close(got, want)
    # PixelShuffle + residual in the shuffled layout, 1x1 kernel
    w1 = torch.randn(8, 32, 1, 1, generator=g) / 6
    b1 = torch.randn(8, generator=g)
    rs = torch.randn(2, 2, 40, 56, generator=g)
    want = F.pixel_shuffle(F.conv2d(x, w1, b1), 2) + rs
    got = ops.conv2d(dev(x), dev(w1), dev(b1), pixel_shuffle=True, residual=dev(rs))
    close(got, want)


@pytest.mark.parametrize("H,W", [(24, 44), (16, 50), (15, 20), (272, 480)])
def test_conv2d_1x1_on_the_flattened_plane(H, W):
    """1x1 stride 1 runs on the plane as rows of 32 pixels when H*W is a multiple of 32 (15 x 20 is not): input and
    output activations, scaled residual and second residual, sources / residuals / destination as channel slices of
    larger buffers, and nothing written around the destination."""
    from deepvideocodec_amd import ops
    g = torch.Generator().manual_seed(H * 7 + W)
    big = torch.randn(2, 96, H, W, generator=g)
    x1, x2 = big[:, 4:52], big[:, 60:76]        # 48 + 16 channels
    w = torch.randn(64, 64, 1, 1, generator=g) / 8
    b = torch.randn(64, generator=g)
    rbig = torch.randn(2, 160, H, W, generator=g)
    r1, r2 = rbig[:, 8:72], rbig[:, 80:144]
    sc = torch.rand(2, 64, generator=g)
    want = (F.leaky_relu(F.conv2d(F.leaky_relu(torch.cat([x1, x2], 1), 0.01), w, b), 0.1) + r1 * sc[:, :, None, None]) + r2
    dbig, drbig = dev(big), dev(rbig)
    buf = torch.full((2, 80, H, W), 7.5, device="cuda:0")
    got = ops.conv2d([dbig[:, 4:52], dbig[:, 60:76]], dev(w), dev(b), in_slope=0.01, out_slope=0.1,
                     residual=drbig[:, 8:72], res_scale=dev(sc).reshape(-1), residual2=drbig[:, 80:144], out=buf[:, 8:72])
    close(got, want)
    assert float((buf[:, :8] - 7.5).abs().max()) == 0.0 and float((buf[:, 72:] - 7.5).abs().max()) == 0.0


@pytest.mark.parametrize("H,W", [(1, 1), (3, 5), (6, 10), (7, 7), (8, 12), (17, 30), (5, 64)])
@pytest.mark.parametrize("shuffle", [False, True])
def test_conv2d_epilogue_paths_stay_inside_output(H, W, shuffle):
    """Odd and even widths pick the 16-byte or the 4-byte store path; the destination is a slice
    in the middle of a sentinel-filled buffer that must come back untouched around it."""
    from deepvideocodec_amd import ops
    g = torch.Generator().manual_seed(100 * H + W + int(shuffle))
    cout = 8 if shuffle else 5
    x = torch.randn(2, 16, H, W, generator=g)
    w = torch.randn(cout, 16, 3, 3, generator=g) / 12
    b = torch.randn(cout, generator=g)
    cf, Ho, Wo = (cout // 4, 2 * H, 2 * W) if shuffle else (cout, H, W)
    res = torch.randn(2, cf, Ho, Wo, generator=g)
    want = F.conv2d(x, w, b, padding=1)
    want = (F.pixel_shuffle(want, 2) if shuffle else want) + res
    pad = 4   # channels of guard band either side of the slice (keeps it 16-byte aligned when H*W % 4 == 0)
    buf = torch.full((2, cf + 2 * pad, Ho, Wo), 7.5, device="cuda:0")
    got = ops.conv2d(dev(x), dev(w), dev(b), pixel_shuffle=shuffle, residual=dev(res), out=buf[:, pad:pad + cf])
    close(got, want)
    assert bool((buf[:, :pad] == 7.5).all()) and bool((buf[:, pad + cf:] == 7.5).all())


def test_flow_warp_and_resampling_match_reference_fixture():
    from conftest import load_golden
    from deepvideocodec_amd import ops
    g = load_golden("ops_layers.npz")
    got = ops.flow_warp(dev(torch.from_numpy(g["warp_feat"])), dev(torch.from_numpy(g["warp_flow"])))
    close(got, torch.from_numpy(g["warp_out"]), rel=2e-5)
    x = dev(torch.from_numpy(g["resample_in"]))
    close(ops.bilinear_up2(x), torch.from_numpy(g["up2_out"]), rel=1e-6)
    close(ops.avg_pool2(x), torch.from_numpy(g["down2_out"]), rel=1e-6)


def test_flow_warp_large_and_views():
    from deepvideocodec_amd import ops
    g = torch.Generator().manual_seed(2)
    feat = torch.randn(1, 64, 136, 240, generator=g)
    flow = torch.randn(1, 2, 136, 240, generator=g) * 5
    want = O.flow_warp(feat, flow)
    got = ops.flow_warp(dev(feat), dev(flow))
    close(got, want, rel=5e-5)
    # zero flow: the identity up to the normalised-grid rounding the reference has too
    z = ops.flow_warp(dev(feat), torch.zeros(1, 2, 136, 240, device="cuda:0"))
    close(z, O.flow_warp(feat, torch.zeros(1, 2, 136, 240)), rel=5e-5)
    close(z, feat, rel=1e-3)


@pytest.mark.parametrize("h,w,batch", [(68, 120, 1), (100, 132, 16), (272, 480, 2), (544, 960, 1), (1088, 1920, 1)])
def test_flow_warp_window_kernel_equals_direct_gather(h, w, batch):
    """Tensors with >= 16 channels and >= 256 tiles of 64 x 16 pixels (all cases here but the first) go through the
    LDS-window kernel (loading and interpolating waves), thinner ones through the direct gather:
    the same planes warped 8 channels at a time must come out bit-identical, for motion that fits the window,
    motion that does not (tiles fall back inside the kernel), and positions clipped at every border."""
    from deepvideocodec_amd import ops
    g = torch.Generator().manual_seed(h + w)
    feat = dev(torch.randn(batch, 40, h, w, generator=g))
    smooth = F.interpolate(torch.randn(batch, 2, max(h // 16, 2), max(w // 16, 2), generator=g) * 2.5, size=(h, w),
                           mode="bilinear", align_corners=False)
    flows = {
        "smooth": smooth,
        "noise": torch.randn(batch, 2, h, w, generator=g) * 1.5,
        "large": torch.randn(batch, 2, h, w, generator=g) * 40,
        "shift": smooth * 0.1 + torch.tensor([37.25, -21.5]).view(1, 2, 1, 1),
        "clip": torch.cat([torch.full((batch, 1, h, w), float(w)), torch.full((batch, 1, h, w), -float(h))], 1),
    }
    for name, flow in flows.items():
        flow = dev(flow.contiguous())
        got = ops.flow_warp(feat, flow)
        want = torch.cat([ops.flow_warp(feat[:, c:c + 8].contiguous(), flow) for c in range(0, 40, 8)], 1)
        assert torch.equal(got, want), (name, float((got - want).abs().max()))
        close(got, O.flow_warp(feat.cpu(), flow.cpu()), rel=5e-5)
        # channel counts that end the three-buffer ring at each of its slots, a source that is a channel slice of a
        # wider tensor and a destination inside a wider tensor (channel strides larger than the plane)
        for nch in (16, 17, 18):
            wide = torch.full((batch, nch + 5, h, w), -7.0, device="cuda:0")
            ops.flow_warp(feat[:, 3:3 + nch], flow, out=wide[:, 2:2 + nch])
            assert torch.equal(wide[:, 2:2 + nch], want[:, 3:3 + nch]), (name, nch)
            assert bool((wide[:, :2] == -7.0).all()) and bool((wide[:, 2 + nch:] == -7.0).all()), (name, nch)


def test_pools_and_scaled_resampling():
    from deepvideocodec_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 7, 36, 52, generator=g)
    close(ops.max_pool2(dev(x)), F.max_pool2d(x, 2, 2), rel=0, abs_=0)
    close(ops.avg_pool2(dev(x)), F.avg_pool2d(x, 2, 2), rel=1e-6)
    close(ops.avg_pool2(dev(x[:, :2]), 0.5), O.bilinear_down2(x[:, :2]) / 2, rel=1e-6)
    close(ops.bilinear_up2(dev(x[:, :2]), 2.0), O.bilinear_up2(x[:, :2]) * 2.0, rel=1e-6)
    close(ops.clamp(dev(x), 0.0, 1.0), x.clamp(0, 1), rel=0, abs_=0)
    dst = torch.zeros(2, 10, 36, 52, device="cuda:0")
    ops.copy_into(dev(x), dst[:, 3:])
    close(dst[:, 3:], x, rel=0, abs_=0)
    assert float(dst[:, :3].abs().max()) == 0.0


def test_spynet_level_input():
    from deepvideocodec_amd import ops
    g = torch.Generator().manual_seed(4)
    im1 = torch.rand(2, 3, 32, 48, generator=g)
    im2 = torch.rand(2, 3, 32, 48, generator=g)
    fl = torch.randn(2, 2, 16, 24, generator=g) * 2
    flow_up = O.bilinear_up2(fl) * 2.0
    want = torch.cat([im1, O.flow_warp(im2, flow_up), flow_up], 1)
    close(ops.spynet_level_input(dev(im1), dev(im2), dev(fl)), want, rel=2e-5)
    zero = torch.zeros(2, 2, 32, 48)
    want0 = torch.cat([im1, O.flow_warp(im2, zero), zero], 1)
    close(ops.spynet_level_input(dev(im1), dev(im2), None), want0, rel=2e-5)


def test_se_gate():
    from deepvideocodec_amd import ops
    g = torch.Generator().manual_seed(6)
    for c, h, w in ((32, 64, 96), (128, 17, 30), (64, 5, 3)):
        x = torch.randn(2, c, h, w, generator=g) + 0.3
        fc1 = torch.randn(c // 16, c, generator=g) / math.sqrt(c)
        fc2 = torch.randn(c, c // 16, generator=g)
        want = torch.sigmoid(F.linear(F.relu(F.linear(x.mean(dim=(-1, -2)), fc1)), fc2))
        got = ops.se_gate(dev(x), dev(fc1), dev(fc2)).view(2, c)
        close(got, want, rel=2e-5)


def _eb(channels, seed):
    torch.manual_seed(seed)
    eb = E.EntropyBottleneck(channels).eval()
    with torch.no_grad():
        for i in range(4):
            getattr(eb, f"_factor{i}").normal_(0, 0.3)
        for i in range(5):
            getattr(eb, f"_matrix{i}").add_(0.1 * torch.randn_like(getattr(eb, f"_matrix{i}")))
        eb.quantiles[:, 0, 1] = torch.rand(channels) * 0.6 - 0.3
    return eb


def test_factorized_prior_kernel():
    from deepvideocodec_amd import ops
    from deepvideocodec_amd.models.entropy_models import EntropyBottleneck
    ref = _eb(64, 0)
    ours = EntropyBottleneck(64)
    ours.load_state_dict(ref.state_dict())
    ours = ours.to("cuda:0").eval()
    z = torch.randn(2, 64, 17, 30) * 4
    want_out, want_lik = ref(z)
    got_out, got_lik = ours(dev(z))
    assert torch.equal(got_out.cpu(), want_out)
    rel = ((got_lik.cpu() - want_lik).abs() / want_lik).max().item()
    assert rel < 5e-4, rel
    bits_w, bits_g = -torch.log2(want_lik).sum().item(), -torch.log2(got_lik.cpu()).sum().item()
    assert abs(bits_w - bits_g) < 1e-5 * z.numel() / 64
    # symbols (exact) and de-quantisation
    z_hat, sym = ours.quantize_symbols(dev(z))
    med = ref._get_medians().view(1, 64, 1, 1)
    assert torch.equal(sym.cpu(), torch.round(z - med).int())
    assert torch.equal(ops.eb_dequantize(sym, ours._medians_flat()).cpu(), want_out)
    # training mode: additive noise, likelihood at the noisy value
    ours.train()
    torch.manual_seed(3)
    out_t, lik_t = ours(dev(z))
    d = (out_t.cpu() - z)
    assert d.abs().max() <= 0.5 and d.std() > 0.2
    want = torch.clamp(ref._likelihood(out_t.cpu().permute(1, 0, 2, 3).reshape(64, 1, -1)), min=1e-9)
    want = want.reshape(64, 2, 17, 30).permute(1, 0, 2, 3)
    assert ((lik_t.cpu() - want).abs() / want).max().item() < 5e-4


@pytest.mark.parametrize("C,h,w", [(64, 68, 120), (96, 17, 31), (8, 4, 4)])
def test_dual_prior_kernels(C, h, w):
    """pass1 / pass2 / decoder passes against the oracle's forward_dual_prior restatement."""
    from deepvideocodec_amd import ops
    g = torch.Generator().manual_seed(C + h)
    y = torch.randn(2, C, h, w, generator=g) * 5
    means = torch.randn(2, C, h, w, generator=g)
    scales = torch.rand(2, C, h, w, generator=g) * 4 - 0.5
    spatial = torch.randn(2, 2 * C, h, w, generator=g)
    spatial[:, C // 2:C] = spatial[:, C // 2:C].abs() * 2 - 0.2
    spatial[:, 3 * C // 2:] = spatial[:, 3 * C // 2:].abs() * 2 - 0.2
    table = torch.exp(torch.linspace(math.log(0.11), math.log(256), 64))
    gc = E.GaussianConditional(None).eval()
    gc.update_scale_table(table, force=True)

    m0, m1 = O.checkerboard_masks(h, w)
    y_0, y_1 = y.chunk(2, 1)
    mu_0, mu_1 = means.chunk(2, 1)
    sc_0, sc_1 = scales.chunk(2, 1)
    q00, yh00, mh00, sh00 = O.process_with_mask(y_0, mu_0, sc_0, m0)
    q11, yh11, mh11, sh11 = O.process_with_mask(y_1, mu_1, sc_1, m1)
    got1 = ops.dual_prior_pass1(dev(y), dev(means), dev(scales))
    assert torch.equal(got1.cpu() + 0.0, torch.cat((yh00, yh11), 1) + 0.0)

    smu_0, ssc_0, smu_1, ssc_1 = spatial.chunk(4, 1)
    q01, yh01, mh01, sh01 = O.process_with_mask(y_0, smu_0, ssc_0, m1)
    q10, yh10, mh10, sh10 = O.process_with_mask(y_1, smu_1, ssc_1, m0)
    y_hat = torch.cat((yh00 + yh01, yh11 + yh10), 1)
    means_hat = torch.cat((mh00 + mh01, mh11 + mh10), 1)
    scales_hat = torch.cat((sh00 + sh01, sh11 + sh10), 1)
    _, lik = gc(y, scales_hat, means_hat)
    r = ops.dual_prior_pass2(dev(y), dev(means), dev(scales), dev(spatial),
                             want=("y_hat", "means_hat", "scales_hat", "likelihood", "symbols"),
                             scale_table=dev(table))
    assert torch.equal(r["y_hat"].cpu(), y_hat)
    assert torch.equal(r["means_hat"].cpu() + 0.0, means_hat + 0.0)
    assert torch.equal(r["scales_hat"].cpu() + 0.0, scales_hat + 0.0)
    gl = r["likelihood"].cpu()
    assert (gl - lik).abs().max().item() < 3e-7
    big = lik > 1e-4
    assert ((gl[big] - lik[big]).abs() / lik[big]).max().item() < 1e-3
    # compress-mode planes: exact integers
    qw0, qw1 = (q00 + q11), (q01 + q10)
    sw0, sw1 = (sh00 + sh11), (sh01 + sh10)
    assert torch.equal(r["symbols"][0].cpu(), qw0.int())
    assert torch.equal(r["symbols"][1].cpu(), qw1.int())
    assert torch.equal(r["indexes"][0].cpu(), gc.build_indexes(sw0))
    assert torch.equal(r["indexes"][1].cpu(), gc.build_indexes(sw1))
    # decoder side
    i0 = ops.dual_prior_dec_indexes(dev(scales), 0, C, dev(table))
    assert torch.equal(i0.cpu(), gc.build_indexes(sc_0 * m0 + sc_1 * m1))
    d1 = ops.dual_prior_dec_apply(dev(qw0.int().contiguous()), dev(means), 0, C)
    assert torch.equal(d1.cpu() + 0.0, torch.cat(((qw0 + mu_0) * m0, (qw0 + mu_1) * m1), 1) + 0.0)
    i1 = ops.dual_prior_dec_indexes(dev(spatial), 1, C, dev(table))
    assert torch.equal(i1.cpu(), gc.build_indexes(ssc_0 * m1 + ssc_1 * m0))
    d2 = ops.dual_prior_dec_apply(dev(qw1.int().contiguous()), dev(spatial), 1, C, yhat1=d1)
    assert torch.equal(d2.cpu(), y_hat)
    # training-mode rate: likelihood at y + noise
    noise = torch.rand(2, C, h, w, generator=g) - 0.5
    _, lik_n = gc(y, scales_hat, means_hat, training=False)  # placeholder shape check
    vals = (y + noise - means_hat).abs()
    s = scales_hat.clamp_min(0.11)
    want_n = (E.standardized_cumulative((0.5 - vals) / s) - E.standardized_cumulative((-0.5 - vals) / s)).clamp_min(1e-9)
    rn = ops.dual_prior_pass2(dev(y), dev(means), dev(scales), dev(spatial), noise=dev(noise), want=("likelihood",))
    assert (rn["likelihood"].cpu() - want_n).abs().max().item() < 3e-7


def test_gaussian_conditional_module_and_indexes():
    from deepvideocodec_amd.models.entropy_models import GaussianConditional
    g = torch.Generator().manual_seed(8)
    table = torch.exp(torch.linspace(math.log(0.11), math.log(256), 64))
    ref = E.GaussianConditional(None).eval()
    ref.update_scale_table(table, force=True)
    ours = GaussianConditional(None).to("cuda:0").eval()
    ours.update_scale_table(table, force=True)
    assert torch.equal(ours._quantized_cdf.cpu(), ref._quantized_cdf)
    y = torch.randn(2, 12, 9, 11, generator=g) * 3
    mu = torch.randn(2, 12, 9, 11, generator=g)
    sc = torch.cat([table, table * 1.00001, torch.rand(2 * 12 * 9 * 11 - 128, generator=g) * 300 - 1]).view(2, 12, 9, 11)
    o_w, l_w = ref(y, sc, mu)
    o_g, l_g = ours(dev(y), dev(sc), dev(mu))
    assert torch.equal(o_g.cpu(), o_w)
    assert (l_g.cpu() - l_w).abs().max().item() < 3e-7
    assert torch.equal(ours.build_indexes(dev(sc)).cpu(), ref.build_indexes(sc))
    # strings: same bytes as the oracle coder, and they decode back
    idx = ref.build_indexes(sc)
    q = torch.round(y - mu)
    s_ref = ref.compress(q, idx)
    s_ours = ours.compress(dev(q), dev(idx))
    assert s_ref == s_ours
    assert torch.equal(ours.decompress(s_ours, dev(idx)).cpu(), q)


def test_bits_and_sse_reductions():
    from deepvideocodec_amd import ops
    g = torch.Generator().manual_seed(10)
    lik = torch.rand(3, 96, 68, 120, generator=g).clamp_min(1e-9)
    want = -torch.log2(lik.double()).sum(dim=(1, 2, 3))
    got = ops.likelihood_bits(dev(lik)).cpu().double()
    assert ((got - want).abs() / want).max().item() < 2e-6
    a = torch.rand(2, 3, 100, 150, generator=g)
    b = torch.rand(2, 3, 100, 150, generator=g)
    want = ((a.double() - b.double()) ** 2).sum(dim=(1, 2, 3))
    got = ops.sq_err_sum(dev(a), dev(b)).cpu().double()
    assert ((got - want).abs() / want).max().item() < 2e-6
    # deterministic: two runs are bit-identical
    assert torch.equal(ops.likelihood_bits(dev(lik)), ops.likelihood_bits(dev(lik)))


BF16_CASES = [
    # k, stride, cins, cout, H, W, batch, in_slope, out_slope, shuffle
    (3, 1, (64,), 64, 36, 52, 1, None, None, False),
    (3, 1, (16, 8), 40, 17, 23, 2, 0.1, 0.01, False),
    (3, 1, (5, 7, 9, 11), 96, 21, 19, 1, None, 0.0, False),
    (3, 2, (67,), 64, 32, 48, 1, None, 0.1, False),
    (1, 1, (64,), 32, 33, 47, 1, None, None, False),
    (1, 2, (12,), 64, 34, 50, 2, None, None, False),
    (3, 1, (24,), 32, 10, 14, 2, None, 0.01, True),
    (3, 1, (288,), 288, 17, 30, 1, None, 0.2, False),
    (3, 1, (64,), 64, 272, 480, 1, 0.01, 0.01, False),
    (3, 1, (64, 64), 64, 544, 960, 1, None, 0.1, False),    # 8-row tiles of 64 channels, 8 chunks: the production variant
    (3, 1, (64,), 64, 544, 962, 1, None, None, False),      # width not a multiple of 4: 16-row form not available
    (3, 1, (48, 3, 64), 128, 544, 960, 1, 0.1, None, False), # 16-row form: chunks straddling sources, ragged last chunk
    (3, 1, (128,), 256, 272, 480, 1, None, None, True),     # 16-row form with the PixelShuffle store
    (3, 1, (64,), 64, 1088, 1920, 1, None, 0.01, False),    # the benchmark's dominant layer
    (3, 1, (16,), 32, 40, 70, 1, None, None, False),        # one chunk: the ring never fills
    (3, 1, (40,), 64, 9, 33, 3, None, None, False),         # ragged last chunk (8 live channels), ragged tiles
    (7, 1, (8,), 32, 30, 50, 1, None, 0.0, False),          # 7x7: half a chunk of channels
    (7, 1, (32,), 64, 24, 40, 2, None, 0.0, False),
    (7, 1, (64,), 32, 136, 240, 1, None, 0.0, False),
    (7, 1, (32,), 16, 45, 37, 1, None, 0.0, False),
    (7, 1, (32,), 64, 272, 480, 1, None, 0.0, False),       # 16-row tiles, one workgroup per CU
    (7, 1, (64,), 32, 272, 480, 1, None, 0.0, False),
]


@pytest.mark.parametrize("k,stride,cins,cout,H,W,batch,in_slope,out_slope,shuffle", BF16_CASES)
def test_conv2d_bf16_mode(k, stride, cins, cout, H, W, batch, in_slope, out_slope, shuffle):
    """The reduced-precision mode: operands rounded to bf16 (after the input activation), products and
    sums in fp32 -- so it must agree with an fp32 convolution of the rounded operands to fp32 noise."""
    from deepvideocodec_amd import ops
    g = torch.Generator().manual_seed(k * 31 + cout + W)
    parts = [torch.randn(batch, c, H, W, generator=g) for c in cins]
    cin = sum(cins)
    w = torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)
    b = torch.randn(cout, generator=g)
    res = torch.randn(batch, cout // 4 if shuffle else cout, *( (2 * ((H + 2 * (k // 2) - k) // stride + 1), 2 * ((W + 2 * (k // 2) - k) // stride + 1)) if shuffle
                      else ((H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1)), generator=g)
    x = torch.cat(parts, 1)
    if in_slope is not None:
        x = F.leaky_relu(x, in_slope)
    want = F.conv2d(x.bfloat16().float(), w.bfloat16().float(), b, stride=stride, padding=k // 2)
    if out_slope is not None:
        want = F.leaky_relu(want, out_slope)
    want = (F.pixel_shuffle(want, 2) if shuffle else want) + res
    got = ops.conv2d([dev(p) for p in parts], dev(w), dev(b), stride=stride, in_slope=in_slope, out_slope=out_slope,
                     pixel_shuffle=shuffle, residual=dev(res), precision=1)
    close(got, want, rel=2e-5)


@pytest.mark.parametrize("mode", ["bf16x2", "fp16x2"])
@pytest.mark.parametrize("k,stride,cins,cout,H,W,batch,in_slope,out_slope,shuffle", BF16_CASES)
def test_conv2d_split_modes(k, stride, cins, cout, H, W, batch, in_slope, out_slope, shuffle, mode):
    """The split modes: every operand is hi + lo with hi = half(v), lo = half(v - hi), half = bf16 or
    (saturating) fp16; a product is hi*hi + hi*lo + lo*hi, summed in fp32.  The result must agree with that
    expression evaluated in fp64 to fp32 noise, and with the exact convolution to ~2^-16 (bf16 parts) or
    ~2^-21 (fp16 parts) per product."""
    from deepvideocodec_amd import ops
    half = torch.bfloat16 if mode == "bf16x2" else torch.float16
    g = torch.Generator().manual_seed(k * 37 + cout + H)
    parts = [torch.randn(batch, c, H, W, generator=g) for c in cins]
    cin = sum(cins)
    w = torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)
    b = torch.randn(cout, generator=g)
    x = torch.cat(parts, 1)
    if in_slope is not None:
        x = F.leaky_relu(x, in_slope)

    def split(t):
        hi = t.to(half).float()
        return hi.double(), (t - hi).to(half).double()

    xh, xl = split(x)
    wh, wl = split(w)
    conv = lambda u, v: F.conv2d(u, v, None, stride=stride, padding=k // 2)
    want = conv(xh, wh) + conv(xh, wl) + conv(xl, wh) + b.double().view(1, -1, 1, 1)
    exact = conv(x.double(), w.double()) + b.double().view(1, -1, 1, 1)
    if out_slope is not None:
        want, exact = F.leaky_relu(want, out_slope), F.leaky_relu(exact, out_slope)
    if shuffle:
        want, exact = F.pixel_shuffle(want, 2), F.pixel_shuffle(exact, 2)
    res = torch.randn(want.shape, generator=g)
    got = ops.conv2d([dev(p) for p in parts], dev(w), dev(b), stride=stride, in_slope=in_slope, out_slope=out_slope,
                     pixel_shuffle=shuffle, residual=dev(res), precision=ops.PRECISIONS[mode])
    close(got, (want + res).float(), rel=2e-5)
    err = (got.cpu().double() - (exact + res)).abs().max().item()
    assert err < (2e-4 if mode == "bf16x2" else 1.5e-5), err


# cins, cout, H, W, batch, n_res, shuffle -- sizes that select the 16-row, one-workgroup-per-CU form of the split modes
PERSISTENT_CASES = [
    ((64,), 64, 544, 960, 1, 2, False),      # two residuals
    ((64,), 128, 544, 960, 2, 1, False),     # the walk crosses channel blocks and samples
    ((128,), 256, 272, 480, 1, 1, True),     # PixelShuffle store with a residual, four channel blocks
    ((64,), 64, 530, 962, 1, 1, False),      # ragged tiles on both edges, rows that are not whole 16-byte pieces
    ((32, 40), 64, 536, 964, 1, 2, False),   # ragged tiles with 16-byte rows; ragged last chunk, chunk across two sources
    ((16,), 64, 544, 960, 1, 0, False),      # a single chunk per tile
]


@pytest.mark.parametrize("mode", ["bf16x2", "fp16x2"])
@pytest.mark.parametrize("cins,cout,H,W,batch,n_res,shuffle", PERSISTENT_CASES)
def test_conv2d_split_modes_persistent_form(cins, cout, H, W, batch, n_res, shuffle, mode):
    """The 16-row form of the split modes keeps one workgroup per CU that walks many tiles, with staging waves that run
    ahead of the multiplying waves across tile boundaries: every fused epilogue and edge case it can meet, against the
    exact convolution (fp64) to the modes' accuracy -- ~2^-16 (bf16 parts) / ~2^-21 (fp16 parts) per product."""
    from deepvideocodec_amd import ops
    torch.set_num_threads(16)
    g = torch.Generator().manual_seed(cout + H + n_res)
    parts = [torch.randn(batch, c, H, W, generator=g) for c in cins]
    cin = sum(cins)
    w = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)
    b = torch.randn(cout, generator=g)
    want = F.leaky_relu(F.conv2d(torch.cat(parts, 1).double(), w.double(), b.double(), padding=1), 0.1)
    if shuffle:
        want = F.pixel_shuffle(want, 2)
    res = [torch.randn(want.shape, generator=g) for _ in range(n_res)]
    for r in res:
        want = want + r
    got = ops.conv2d([dev(p) for p in parts], dev(w), dev(b), out_slope=0.1, pixel_shuffle=shuffle,
                     residual=dev(res[0]) if n_res >= 1 else None, residual2=dev(res[1]) if n_res == 2 else None,
                     precision=ops.PRECISIONS[mode])
    err = (got.cpu().double() - want).abs().max().item()
    assert err < (2e-4 if mode == "bf16x2" else 1.5e-5), err
    # twice in a row into the same buffer: the walk leaves nothing behind
    again = ops.conv2d([dev(p) for p in parts], dev(w), dev(b), out_slope=0.1, pixel_shuffle=shuffle,
                       residual=dev(res[0]) if n_res >= 1 else None, residual2=dev(res[1]) if n_res == 2 else None,
                       precision=ops.PRECISIONS[mode], out=torch.full_like(got, float("nan")))
    assert torch.equal(again, got)


def test_conv2d_fp16_split_range():
    """fp16 parts: operands far below the fp16 normal range keep an absolute accuracy of ~2^-25 (subnormal
    lo parts are not flushed), and operands beyond +-65504 saturate instead of becoming infinite."""
    from deepvideocodec_amd import ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 32, 20, 36, generator=g)
    w = torch.randn(32, 32, 3, 3, generator=g) / 17
    for scale in (1e-3, 1.0, 300.0):
        xs = x * scale
        exact = F.conv2d(xs.double(), w.double(), None, padding=1)
        got = ops.conv2d(dev(xs), dev(w), None, precision=ops.PRECISIONS["fp16x2"]).cpu().double()
        assert (got - exact).abs().max().item() < 3e-6 * max(scale, 1e-2) * 4, scale
    big = x.clone()
    big[0, 0, 3, 3] = 5e5
    got = ops.conv2d(dev(big), dev(w), None, precision=ops.PRECISIONS["fp16x2"])
    assert torch.isfinite(got).all()


THIN_CASES = [
    # k, cin, cout, H, W, batch, in_slope, out_slope, residuals
    (7, 16, 2, 131, 517, 1, None, None, 1),      # SpyNet level output: + up-sampled flow (a channel slice)
    (7, 16, 2, 64, 96, 6, 0.0, None, 0),         # batch supplies the tiles
    (3, 64, 3, 136, 500, 1, None, None, 0),      # reconstruction layer
    (3, 21, 4, 130, 515, 1, 0.1, 0.01, 2),
    (3, 9, 1, 128, 512, 1, None, 0.2, 1),
    (7, 5, 3, 129, 513, 1, 0.01, None, 0),       # fewer channels than one LDS chunk
]


@pytest.mark.parametrize("k,cin,cout,H,W,batch,in_slope,out_slope,nres", THIN_CASES)
def test_conv2d_few_output_channels(k, cin, cout, H, W, batch, in_slope, out_slope, nres):
    """Layers with at most 4 output channels and enough pixels run on the direct (vector-ALU) kernel:
    same results as the convolution of the torch reference, fused activations and residuals included,
    destination and residual given as channel slices of larger buffers."""
    from deepvideocodec_amd import ops
    g = torch.Generator().manual_seed(k * 131 + cin * 7 + cout)
    x = torch.randn(batch, cin, H, W, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)
    b = torch.randn(cout, generator=g)
    rbig = torch.randn(batch, cout + 6, H, W, generator=g)
    r2 = torch.randn(batch, cout, H, W, generator=g)
    xin = F.leaky_relu(x, in_slope) if in_slope is not None else x
    want = F.conv2d(xin, w, b, padding=k // 2)
    if out_slope is not None:
        want = F.leaky_relu(want, out_slope)
    if nres >= 1:
        want = want + rbig[:, 3:3 + cout]
    if nres == 2:
        want = want + r2
    out_big = torch.zeros(batch, cout + 5, H, W, device="cuda:0")
    rb = dev(rbig)
    ops.conv2d(dev(x), dev(w), dev(b), in_slope=in_slope, out_slope=out_slope,
               residual=rb[:, 3:3 + cout] if nres >= 1 else None, residual2=dev(r2) if nres == 2 else None,
               out=out_big[:, 2:2 + cout])
    close(out_big[:, 2:2 + cout], want)
    assert float(out_big[:, :2].abs().max()) == 0.0 and float(out_big[:, 2 + cout:].abs().max()) == 0.0


@pytest.mark.parametrize("cin,H,W,batch,in_slope,out_slope", [
    (32, 64, 64, 2, None, 0.0),      # 4-row tiles
    (32, 560, 960, 1, None, 0.0),    # 8-row tiles (35 x 30 workgroups)
    (32, 544, 960, 1, 0.1, 0.0),     # 16-row tiles (17 x 30 workgroups), activation on the input
    (2, 64, 96, 1, None, None),      # a single ragged chunk (the data gradient of a 16 -> 2 layer)
    (32, 72, 64, 1, None, 0.0),      # height 72 = 9 x 8: only the 4-row tile divides it
    (32, 68, 64, 1, None, 0.0),      # height the row-pair tiles do not divide: ordinary form
])
def test_conv2d_7x7_sixteen_channels_row_pairs(cin, H, W, batch, in_slope, out_slope):
    """7x7 layers with exactly 16 output channels run in the row-pair form (two output rows per accumulator
    tile) where its tiles divide the plane.  Same sums as the ordinary form: the first 16 channels of the same
    layer widened to 17 output channels (ordinary form) must be bit-identical, and both match the torch conv.
    The destination is a channel slice of a larger buffer."""
    from deepvideocodec_amd import ops
    g = torch.Generator().manual_seed(cin * 1000 + H)
    x = torch.randn(batch, cin, H, W, generator=g)
    w = torch.randn(17, cin, 7, 7, generator=g) / math.sqrt(cin * 49)
    b = torch.randn(17, generator=g)
    xin = F.leaky_relu(x, in_slope) if in_slope is not None else x
    want = F.conv2d(xin, w[:16], b[:16], padding=3)
    if out_slope is not None:
        want = F.leaky_relu(want, out_slope)
    out_big = torch.zeros(batch, 21, H, W, device="cuda:0")
    ops.conv2d(dev(x), dev(w[:16].contiguous()), dev(b[:16].contiguous()), in_slope=in_slope, out_slope=out_slope,
               out=out_big[:, 2:18])
    wide = ops.conv2d(dev(x), dev(w), dev(b), in_slope=in_slope, out_slope=out_slope)
    close(out_big[:, 2:18], want)
    assert torch.equal(out_big[:, 2:18], wide[:, :16]), float((out_big[:, 2:18] - wide[:, :16]).abs().max())
    assert float(out_big[:, :2].abs().max()) == 0.0 and float(out_big[:, 18:].abs().max()) == 0.0


@pytest.mark.parametrize("cin,cout,H,W,batch,out_slope", [
    (32, 32, 64, 96, 2, None),      # 32-channel blocks
    (64, 64, 256, 256, 4, None),    # 64-channel blocks, batch
    (32, 128, 272, 480, 1, 0.01),   # two channel blocks per tile, activation before the sums
    (64, 64, 544, 960, 1, None),    # 16-row tiles
])
def test_conv2d_plane_sums_feed_the_se_gate(cin, cout, H, W, batch, out_slope):
    """A 3x3 convolution made of whole tiles can emit per-tile sums of its output planes with its stores; the
    squeeze-excite gate computed from them equals the gate computed by a pass over the stored tensor (to rounding:
    the two sum in different orders), and the stored tensor itself is unchanged."""
    from deepvideocodec_amd import ops
    g = torch.Generator().manual_seed(cin + cout + H)
    x = dev(torch.randn(batch, cin, H, W, generator=g))
    w = dev(torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9))
    b = dev(torch.randn(cout, generator=g))
    fc1 = dev(torch.randn(cout // 16, cout, generator=g) * 0.3)
    fc2 = dev(torch.randn(cout, cout // 16, generator=g) * 0.3)
    plain = ops.conv2d(x, w, b, out_slope=out_slope)
    out, sums = ops.conv2d(x, w, b, out_slope=out_slope, plane_sums=True)
    assert sums is not None and sums.shape[:2] == (batch, cout)
    assert torch.equal(out, plain)
    want = F.conv2d(x.cpu(), w.cpu(), b.cpu(), padding=1)       # an implementation that shares nothing with the kernel
    want = want if out_slope is None else F.leaky_relu(want, out_slope)
    close(out, want)
    close(sums.sum(dim=2), want.double().sum(dim=(2, 3)).float(), rel=2e-6)
    close(sums.sum(dim=2), plain.double().sum(dim=(2, 3)).float().cpu(), rel=2e-6)
    gate = ops.se_gate_from_sums(sums, H, W, fc1, fc2)
    close(gate, ops.se_gate(plain, fc1, fc2).cpu(), rel=1e-6)
    # not available: ragged tiles, a residual, a reduced-operand mode -> None, and the plain result
    y = dev(torch.randn(1, cin, 50, 70, generator=g))
    o2, s2 = ops.conv2d(y, w, b, plane_sums=True)
    assert s2 is None and torch.equal(o2, ops.conv2d(y, w, b))
    o3, s3 = ops.conv2d(x, w, b, residual=plain, plane_sums=True)
    assert s3 is None
