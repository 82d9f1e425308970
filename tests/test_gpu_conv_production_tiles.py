"""Element-wise parity of the convolution kernel variants that carry the 1080p benchmark, at sizes that
select the production tiles (16 x 32-pixel tiles of 64 output channels, the 8-row stride-2 tiles, the
flattened 1x1 form, the 7x7 forms including row pairs, the direct kernel for <= 4 output channels, the
68 x 120 latent layers), each with the fused prologue / epilogue it runs with in the coded frame:
input activation, channel concat of 2-3 sources, output activation, residual, second residual, gated
residual, PixelShuffle(2), plane sums.  Reference: torch.nn.functional.conv2d (+ activation / residual /
pixel_shuffle) in fp32 on the CPU -- an implementation that shares nothing with the kernels.  Tolerance
2e-5 of the output's magnitude (both sides are fp32 sums of up to 4 320 products in different orders).

`expect` names the (MT, NT) template arguments the library is expected to pick (bench.conv_variant states the
same rule); the test fails if the size no longer selects the production tile.
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def dev(t):
    return t.to("cuda:0")


def close(got, want, rel=2e-5, abs_=1e-6):
    got = got.detach().cpu()
    assert got.shape == want.shape, (got.shape, want.shape)
    tol = rel * max(1.0, float(want.abs().max())) + abs_
    err = float((got - want).abs().max())
    assert err <= tol, f"max err {err:.3e} > tol {tol:.3e}"


def lrelu(x, slope):
    return x if slope is None else F.leaky_relu(x, slope)


# name, k, stride, cins, cout, H, W, in_slope, out_slope, n_res (0/1/2), gate, shuffle, expect (MT, NT)
CASES = [
    ("k3s1 64->64 plain, 1088x1920",      3, 1, (64,),        64, 1088, 1920, 0.01, 0.01, 0, False, False, (2, 4)),
    ("k3s1 64->64 residual",              3, 1, (64,),        64,  544,  960, None, None, 1, False, False, (2, 4)),
    ("k3s1 64->64 two residuals",         3, 1, (64,),        64,  544,  960, None, None, 2, False, False, (2, 4)),
    ("k3s1 64->64 residual, 1088x1920",   3, 1, (64,),        64, 1088, 1920, None, 0.1,  1, False, False, (2, 4)),
    ("k3s1 cat(64,64)->64",               3, 1, (64, 64),     64,  544,  960, None, None, 0, False, False, (2, 4)),
    ("k3s1 cat(32,64)->64",               3, 1, (32, 64),     64,  544,  960, None, None, 0, False, False, (2, 4)),
    ("k3s1 cat(64,64)->256 shuffle",      3, 1, (64, 64),    256,  272,  480, None, None, 0, False, True,  (2, 4)),
    ("k3s1 128->128 shuffle",             3, 1, (128,),      128,  272,  480, None, None, 0, False, True,  (2, 4)),
    ("k3s1 64->128 relu-in residual",     3, 1, (64,),       128,  544,  960, 0.1,  0.1,  1, False, False, (2, 4)),
    ("k3s1 64->32",                       3, 1, (64,),        32,  544,  960, None, 0.01, 0, False, False, (1, 4)),
    ("k3s1 32->32",                       3, 1, (32,),        32,  544,  960, None, None, 0, False, False, (1, 4)),
    ("k3s1 64->64 @272x480 (8-row tile)", 3, 1, (64,),        64,  272,  480, 0.01, 0.01, 1, False, False, (2, 2)),
    ("k3s2 cat(3,64)->64",                3, 2, (3, 64),      64,  544,  960, None, None, 0, False, False, (2, 2)),
    ("k3s2 64->64",                       3, 2, (64,),        64,  544,  960, None, None, 0, False, False, (2, 2)),
    ("k3s2 128->64",                      3, 2, (128,),       64,  544,  960, None, None, 0, False, False, (2, 2)),
    ("k1 cat(64,64)->64 gated residual",  1, 1, (64, 64),     64,  544,  960, None, None, 1, True,  False, (2, 4)),
    ("k1 64->32 gated residual",          1, 1, (64,),        32,  544,  960, None, None, 1, True,  False, (1, 4)),
    ("k1 64->64 plain",                   1, 1, (64,),        64,  544,  960, None, None, 0, False, False, (2, 4)),
    ("k1 64->128 shuffle",                1, 1, (64,),       128,  272,  480, None, 0.01, 0, False, True,  (2, 4)),
    ("k1 128->256 shuffle",               1, 1, (128,),      256,  272,  480, None, None, 0, False, True,  (2, 4)),
    ("k7 64->32",                         7, 1, (64,),        32,  544,  960, None, 0.0,  0, False, False, (1, 4)),
    ("k7 32->64",                         7, 1, (32,),        64,  544,  960, None, 0.0,  0, False, False, (2, 4)),
    ("k7 8->32",                          7, 1, (8,),         32,  544,  960, None, 0.0,  0, False, False, (1, 4)),
    ("k7 32->16 (row pairs)",             7, 1, (32,),        16,  544,  960, None, 0.0,  0, False, False, (1, 4)),
    ("k7 16->2 residual (direct kernel)", 7, 1, (16,),         2,  544,  960, None, None, 1, False, False, (0, 0)),
    ("k3 64->3 (direct kernel)",          3, 1, (64,),         3,  544,  960, None, None, 0, False, False, (0, 0)),
    ("latent cat(192,192,96)->384",       3, 1, (192, 192, 96), 384, 68, 120, None, 0.2,  0, False, False, (2, 1)),
    ("latent 384->288",                   3, 1, (384,),      288,   68,  120, None, 0.2,  0, False, False, (1, 1)),
    ("latent cat(96,96,96)->288",         3, 1, (96, 96, 96), 288,  68,  120, None, 0.2,  0, False, False, (1, 1)),
    ("latent 288->192",                   3, 1, (288,),      192,   68,  120, None, None, 0, False, False, (1, 1)),
    ("latent cat(128,64)->192",           3, 1, (128, 64),   192,   68,  120, None, 0.2,  0, False, False, (1, 1)),
    ("latent 192->128",                   3, 1, (192,),      128,   68,  120, None, None, 0, False, False, (1, 1)),
]


@pytest.mark.parametrize("name,k,stride,cins,cout,H,W,in_slope,out_slope,n_res,gate,shuffle,expect", CASES,
                         ids=[c[0] for c in CASES])
def test_production_tile_variant_matches_torch(name, k, stride, cins, cout, H, W, in_slope, out_slope, n_res, gate,
                                               shuffle, expect):
    import bench
    from deepvideocodec_amd import ops
    torch.set_num_threads(16)
    g = torch.Generator().manual_seed(len(name) * 131 + cout + H)
    parts = [torch.randn(1, c, H, W, generator=g) for c in cins]
    cin = sum(cins)
    w = torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)
    b = torch.randn(cout, generator=g)
    ho, wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
    thin_ok = len(cins) == 1 and not shuffle and not gate
    assert bench.conv_variant(1, cout, ho, wo, k, stride, thin_ok, shuffle) == expect, "size no longer selects this tile"
    oshape = (1, cout // 4, 2 * ho, 2 * wo) if shuffle else (1, cout, ho, wo)
    res = [torch.randn(oshape, generator=g) for _ in range(n_res)]
    sc = torch.rand(1, oshape[1], generator=g) if gate else None
    with torch.no_grad():
        want = lrelu(F.conv2d(lrelu(torch.cat(parts, 1), in_slope), w, b, stride=stride, padding=k // 2), out_slope)
        if shuffle:
            want = F.pixel_shuffle(want, 2)
        if n_res:
            want = want + (res[0] * sc[:, :, None, None] if gate else res[0])
        if n_res == 2:
            want = want + res[1]
    got = ops.conv2d([dev(p) for p in parts], dev(w), dev(b), stride=stride, in_slope=in_slope, out_slope=out_slope,
                     residual=dev(res[0]) if n_res else None, res_scale=dev(sc).reshape(-1) if gate else None,
                     residual2=dev(res[1]) if n_res == 2 else None, pixel_shuffle=shuffle)
    close(got, want)


@pytest.mark.parametrize("c,H,W", [(64, 544, 960), (32, 1088, 1920), (128, 272, 480)])
def test_plane_sums_of_production_tiles_match_torch(c, H, W):
    """The squeeze-excite gate from the plane sums the convolution's epilogue emits, against conv2d -> mean ->
    Linear -> ReLU -> Linear -> Sigmoid in plain PyTorch (dmc/models/layers.py:267-280)."""
    from deepvideocodec_amd import ops
    torch.set_num_threads(16)
    g = torch.Generator().manual_seed(c + H)
    x = torch.randn(1, c, H, W, generator=g)
    w = torch.randn(c, c, 3, 3, generator=g) / math.sqrt(9 * c)
    b = torch.randn(c, generator=g)
    fc1 = torch.randn(c // 16, c, generator=g) / math.sqrt(c)
    fc2 = torch.randn(c, c // 16, generator=g)
    with torch.no_grad():
        want = F.conv2d(x, w, b, padding=1)
        mean = want.double().mean((2, 3))
        want_gate = torch.sigmoid(F.relu(mean @ fc1.double().t()) @ fc2.double().t()).float().reshape(-1)
    out, sums = ops.conv2d(dev(x), dev(w), dev(b), plane_sums=True)
    assert sums is not None, "this size is expected to emit plane sums with its stores"
    close(out, want)
    gate = ops.se_gate_from_sums(sums, H, W, dev(fc1), dev(fc2))
    close(gate, want_gate, rel=1e-5)
    close(ops.se_gate(out, dev(fc1), dev(fc2)), want_gate, rel=1e-5)


def test_plane_sum_path_does_not_depend_on_the_batch():
    """64 channels at 272 x 480: one sample alone yields 8-row tiles, two samples together would yield 16-row tiles if
    the tile were chosen from the whole launch.  The gate (and the stored planes) of a sample must be the same bits
    either way: a stream coded as a batch is decodable sample by sample."""
    from deepvideocodec_amd import ops
    from deepvideocodec_amd.models.layers import ConvBlockResidual
    g = torch.Generator().manual_seed(77)
    for c_in, c_out, H, W in ((64, 64, 272, 480), (32, 64, 384, 640)):
        x = dev(torch.randn(2, c_in, H, W, generator=g))
        blk = ConvBlockResidual(c_in, c_out).to("cuda:0").eval()
        with torch.no_grad():
            both = blk(x)
            one = torch.cat([blk(x[i:i + 1].contiguous()) for i in range(2)], 0)
            t = blk.conv[0](x, out_slope=0.01)
            _, s2 = blk.conv[2](t, plane_sums=True)
            _, s1 = blk.conv[2](t[:1].contiguous(), plane_sums=True)
        assert (s1 is None) == (s2 is None)
        if s1 is not None:
            assert s1.shape[1:] == s2.shape[1:] and torch.equal(s1[0], s2[0])
        assert torch.equal(both, one), float((both - one).abs().max())
