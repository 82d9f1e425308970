"""Gradient kernels against torch.autograd on the CPU (fp32, same op graph as the reference's
nn.Conv2d / LeakyReLU / PixelShuffle / grid_sample / pooling / interpolate layers)."""
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import dmc_oracle as O

pytestmark = pytest.mark.gpu


def dev(t):
    return t.to("cuda:0")


def close(got, want, rel=2e-4, what=""):
    got = got.detach().cpu()
    scale = max(1e-6, float(want.abs().max()))
    err = float((got - want).abs().max())
    assert tuple(got.shape) == tuple(want.shape), (what, got.shape, want.shape)
    assert err <= rel * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e}"


GRAD_CONV_CASES = [
    # k, stride, cins, cout, H, W, batch, in_slope, out_slope, shuffle
    (3, 1, (64,), 64, 36, 52, 2, None, None, False),
    (3, 1, (16, 8), 40, 17, 23, 1, 0.1, 0.01, False),
    (3, 1, (5, 7, 9, 11), 32, 21, 19, 2, None, 0.0, False),
    (3, 1, (96,), 128, 12, 20, 1, None, 0.2, False),
    (3, 2, (67,), 64, 32, 48, 1, None, None, False),
    (3, 2, (16,), 24, 25, 31, 2, 0.01, 0.1, False),
    (7, 1, (8,), 32, 30, 50, 1, None, 0.0, False),
    (7, 1, (16,), 2, 20, 37, 2, None, None, False),
    (7, 1, (32,), 64, 19, 70, 2, None, 0.0, False),      # two output-channel blocks, ragged rows and columns
    (7, 1, (40, 24), 16, 9, 33, 1, 0.1, None, False),    # two input-channel blocks over two sources
    (1, 1, (64,), 32, 33, 47, 1, None, None, False),
    (1, 2, (12,), 64, 34, 50, 2, None, None, False),
    (3, 1, (24,), 32, 10, 14, 2, None, 0.01, True),
    (1, 1, (32,), 8, 9, 11, 1, None, None, True),
    (3, 1, (130,), 70, 8, 40, 1, 0.2, None, False),
]


@pytest.mark.parametrize("k,stride,cins,cout,H,W,batch,in_slope,out_slope,shuffle", GRAD_CONV_CASES)
def test_conv2d_gradients(k, stride, cins, cout, H, W, batch, in_slope, out_slope, shuffle):
    from deepvideocodec_amd import autograd as A
    g = torch.Generator().manual_seed(k * 977 + cout + H + stride)
    cin = sum(cins)
    for _ in range(8):
        parts = [torch.randn(batch, c, H, W, generator=g) for c in cins]
        w = torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)
        b = torch.randn(cout, generator=g)
        # a pre-activation within rounding noise of the ReLU kink would get either slope, legitimately
        # different between two fp32 implementations: draw again
        xin = torch.cat(parts, 1)
        pre = F.conv2d(F.leaky_relu(xin, in_slope) if in_slope is not None else xin, w, b, stride=stride, padding=k // 2)
        if out_slope is None or float(pre.abs().min()) > 1e-5:
            break

    def ref(parts, w, b):
        x = torch.cat(parts, 1)
        if in_slope is not None:
            x = F.leaky_relu(x, in_slope)
        y = F.conv2d(x, w, b, stride=stride, padding=k // 2)
        if out_slope is not None:
            y = F.leaky_relu(y, out_slope)
        return F.pixel_shuffle(y, 2) if shuffle else y

    cp = [p.clone().requires_grad_() for p in parts]
    cw, cb = w.clone().requires_grad_(), b.clone().requires_grad_()
    y_ref = ref(cp, cw, cb)
    R = torch.randn(y_ref.shape, generator=g)
    (y_ref * R).sum().backward()

    gp = [dev(p).requires_grad_() for p in parts]
    gw, gb = dev(w).requires_grad_(), dev(b).requires_grad_()
    y = A.conv2d(gp, gw, gb, stride=stride, in_slope=in_slope, out_slope=out_slope, pixel_shuffle=shuffle)
    close(y, y_ref.detach(), what="forward")
    (y * dev(R)).sum().backward()
    close(gw.grad, cw.grad, what="grad weight")
    close(gb.grad, cb.grad, what="grad bias")
    for i, (a, c) in enumerate(zip(gp, cp)):
        close(a.grad, c.grad, what=f"grad src[{i}]")


@pytest.mark.parametrize("mode", ["fp16x2", "bf16x2"])
@pytest.mark.parametrize("k,stride,cins,cout,H,W,batch,in_slope,out_slope,shuffle", [
    (3, 1, (64,), 64, 64, 96, 8, None, 0.01, False),      # 16-row persistent form in both directions
    (3, 1, (16, 8), 40, 17, 23, 1, 0.1, 0.01, False),
    (3, 2, (67,), 64, 32, 48, 1, None, None, False),      # data gradient through the zero-upsampled stride-1 convolution
    (7, 1, (32,), 64, 19, 70, 2, None, 0.0, False),
    (1, 1, (64,), 32, 33, 47, 1, None, None, False),
    (3, 1, (24,), 32, 10, 14, 2, None, 0.01, True),
    (3, 1, (3,), 64, 20, 28, 1, None, None, False),       # three input channels: the data gradient stays exact fp32
])
def test_conv2d_gradients_split_operand_modes(k, stride, cins, cout, H, W, batch, in_slope, out_slope, shuffle, mode):
    """Training with split 16-bit operands in the forward and data-gradient convolutions (DMC.set_precision(mode,
    train=True)): outputs and gradients agree with torch.autograd in fp64 to the modes' accuracy; the weight and bias
    gradients come from the fp32 kernels and see the mode only through the upstream gradient."""
    from deepvideocodec_amd import autograd as A, ops
    p = ops.PRECISIONS[mode]
    g = torch.Generator().manual_seed(k * 31 + cout + H + stride)
    cin = sum(cins)
    parts = [torch.randn(batch, c, H, W, generator=g) for c in cins]
    w = torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)
    b = torch.randn(cout, generator=g)
    cp = [q.double().requires_grad_() for q in parts]
    cw, cb = w.double().requires_grad_(), b.double().requires_grad_()
    x = torch.cat(cp, 1)
    pre = F.conv2d(F.leaky_relu(x, in_slope) if in_slope is not None else x, cw, cb, stride=stride, padding=k // 2)
    y_ref = F.leaky_relu(pre, out_slope) if out_slope is not None else pre
    # an output within the mode's rounding noise of the activation's kink gets either slope, legitimately: such
    # elements (a few per million) receive no upstream gradient here
    keep = (pre.detach().abs() > 1e-3).float()
    if shuffle:
        y_ref, keep = F.pixel_shuffle(y_ref, 2), F.pixel_shuffle(keep, 2)
    R = torch.randn(y_ref.shape, generator=g) * keep
    (y_ref * R.double()).sum().backward()

    gp = [dev(q).requires_grad_() for q in parts]
    gw, gb = dev(w).requires_grad_(), dev(b).requires_grad_()
    pf, pd = (p if cout > 4 else 0), (p if cin > 4 else 0)
    y = A.conv2d(gp, gw, gb, stride=stride, in_slope=in_slope, out_slope=out_slope, pixel_shuffle=shuffle,
                 packed=ops.pack_conv_weight(dev(w), stride, precision=pf), dgrad_packed=A.dgrad_weight_image(dev(w), pd),
                 precision=pf, dgrad_precision=pd)
    rel = 3e-4 if mode == "bf16x2" else 2e-5
    close(y, y_ref.detach().float(), rel=rel, what="forward")
    (y * dev(R)).sum().backward()
    close(gw.grad, cw.grad.float(), rel=rel, what="grad weight")
    close(gb.grad, cb.grad.float(), rel=rel, what="grad bias")
    for i, (a, c) in enumerate(zip(gp, cp)):
        close(a.grad, c.grad.float(), rel=rel, what=f"grad src[{i}]")


def test_training_step_in_split_operand_mode_tracks_fp32():
    """One optimisation step of the whole model with fp16x2 operands in the forward / data-gradient convolutions: loss
    terms and the updated weights stay within 1e-4 (relative) of the fp32 step."""
    from deepvideocodec_amd import train as TR
    from deepvideocodec_amd.models import DMC
    from deepvideocodec_amd.testing import seeded_state_dict, synthetic_gop
    results = {}
    for mode in ("fp32", "fp16x2"):
        torch.manual_seed(1234)   # the quantisation noise of the training forward pass: the same draw in both runs
        net = DMC()
        net.load_state_dict(seeded_state_dict(seed=0))
        net = net.to("cuda:0").train().set_precision(mode, train=True)
        crit = TR.RateDistortionLoss(lmbda=2048.0)
        opt, aux_opt = TR.configure_optimizers(net, 1e-4, 1e-3)
        frames = [f.to("cuda:0") for f in synthetic_gop(3, 128, 128, seed=5, batch=2)]
        stats = TR.train_step(net, crit, frames, opt, aux_opt, 1.0, 45)
        results[mode] = (stats, {k: v.detach().clone() for k, v in net.state_dict().items() if v.dtype == torch.float32})
    a, b = results["fp16x2"], results["fp32"]
    for key in ("loss", "mse_loss", "bpp_loss"):
        assert abs(a[0][key] - b[0][key]) <= 1e-4 * abs(b[0][key]), (key, a[0][key], b[0][key])
    # Adam's first step moves every weight by lr * sign(grad) (up to epsilon): compare where the gradient is not noise
    worst = 0.0
    for k in b[1]:
        if "quantiles" in k or b[1][k].numel() == 0:   # (quantiles: stepped by the auxiliary optimiser at ten times the rate)
            continue
        d = (a[1][k] - b[1][k]).abs().max().item()
        worst = max(worst, d)
    assert worst <= 2.1e-4, worst   # at most one sign flip of a 1e-4 step on a gradient within rounding of zero


def test_conv2d_wgrad_large_reduction():
    """A reduction long enough for many pixel slabs per block (256x256 crops, batch 4)."""
    from deepvideocodec_amd import autograd as A
    g = torch.Generator().manual_seed(3)
    x = torch.randn(4, 64, 256, 256, generator=g)
    w = torch.randn(64, 64, 3, 3, generator=g) / 24
    b = torch.randn(64, generator=g)
    cx, cw, cb = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    y_ref = F.conv2d(cx, cw, cb, padding=1)
    R = torch.randn(y_ref.shape, generator=g) / 64 + 0.01
    (y_ref * R).sum().backward()
    gx, gw, gb = dev(x).requires_grad_(), dev(w).requires_grad_(), dev(b).requires_grad_()
    y = A.conv2d(gx, gw, gb)
    (y * dev(R)).sum().backward()
    close(gw.grad, cw.grad, rel=5e-4, what="grad weight")
    close(gb.grad, cb.grad, rel=5e-4, what="grad bias")      # summed in the weight-gradient pass
    close(gx.grad, cx.grad, what="grad x")


@pytest.mark.parametrize("k,cin,cout,n,h,w,slope", [
    (7, 16, 2, 3, 64, 96, None),      # the last layer of a motion-estimation level: two thread groups per tile
    (3, 40, 3, 2, 45, 83, 0.01),      # ragged tiles, activation on the input
    (3, 8, 4, 2, 33, 40, None),       # four thread groups per tile
    (7, 18, 1, 1, 20, 70, 0.0),       # the widest 7x7 input the form takes
    (3, 64, 3, 2, 45, 83, 0.01),      # the reconstruction layer: past the form's limit, on the matrix cores
    (7, 8, 32, 2, 64, 96, None),      # the first layer of a motion-estimation level: 8 channels x 4 taps per block
    (7, 5, 40, 1, 37, 50, 0.0),       # the same form with fewer channels, two output blocks, ragged tiles
])
def test_conv2d_wgrad_thin_and_narrow_layers(k, cin, cout, n, h, w, slope):
    """Layers with <= 4 output channels and Cin x k <= 128 take the vector-ALU weight-gradient kernel, 7x7 layers with
    <= 8 input channels the narrow-input form of the matrix-core kernel: against torch autograd, with the input split
    over two concatenated sources."""
    from deepvideocodec_amd import ops
    g = torch.Generator().manual_seed(k * 100 + cin)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = (torch.randn(cout, cin, k, k, generator=g) / (k * cin ** 0.5)).requires_grad_()
    b = torch.zeros(cout, requires_grad=True)
    xa = x if slope is None else F.leaky_relu(x, slope)
    y = F.conv2d(xa, wt, b, padding=k // 2)
    R = torch.randn(y.shape, generator=g)
    (y * R).sum().backward()
    cut = cin // 3 + 1
    srcs = (dev(x[:, :cut].contiguous()), dev(x[:, cut:].contiguous()))
    gw, gb = ops.conv2d_wgrad(srcs, dev(R), wt.shape, 1, in_slope=slope, with_bias=True)
    close(gw, wt.grad, rel=5e-5, what="grad weight")
    close(gb, b.grad, rel=5e-5, what="grad bias")
    gw1, _ = ops.conv2d_wgrad(dev(x), dev(R), wt.shape, 1, in_slope=slope, with_bias=True)
    assert torch.equal(gw1, gw)       # one source or two: the same sums in the same order


@pytest.mark.parametrize("cin,cout,n,h,w", [(40, 72, 3, 13, 16), (32, 32, 4, 8, 8), (96, 24, 2, 16, 11), (192, 128, 8, 16, 16)])
def test_conv2d_wgrad_narrow_planes(cin, cout, n, h, w):
    """3x3 stride-1 layers on planes at most 16 pixels wide (the latent layers of a training crop) use tiles of
    8 x 16 instead of 4 x 32 pixels: all four block shapes, ragged rows and columns, against torch autograd."""
    from deepvideocodec_amd import ops
    g = torch.Generator().manual_seed(cin + cout)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)).requires_grad_()
    b = torch.zeros(cout, requires_grad=True)
    y = F.conv2d(F.leaky_relu(x, 0.2), wt, b, padding=1)
    R = torch.randn(y.shape, generator=g)
    (y * R).sum().backward()
    gw, gb = ops.conv2d_wgrad(dev(x), dev(R), wt.shape, 1, in_slope=0.2, with_bias=True)
    close(gw, wt.grad, rel=5e-5, what="grad weight")
    close(gb, b.grad, rel=5e-5, what="grad bias")


@pytest.mark.parametrize("start_relu,end_relu,bottleneck,slope,with_plus", [
    (True, False, False, 0.01, False),    # the default block (layers.py:59-81)
    (True, True, True, 0.1, False),       # bottleneck block of the contextual coder: activation in and out
    (False, False, False, 0.0, True),     # no input activation (the sum rides in the data-gradient epilogue), ReLU, extra addend
    (True, False, False, 0.01, True),
])
def test_res_block_as_one_tape_node(start_relu, end_relu, bottleneck, slope, with_plus):
    """Under autograd a ResBlock is one node whose backward forms x's gradient (output gradient + masked data gradient)
    inside a kernel: forward, input / addend gradients and the four parameter gradients against plain PyTorch."""
    from deepvideocodec_amd.models.layers import ResBlock
    torch.manual_seed(17)
    blk = ResBlock(32, slope=slope, start_from_relu=start_relu, end_with_relu=end_relu, bottleneck=bottleneck)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 32, 24, 40, generator=g)
    p = torch.randn(2, 32, 24, 40, generator=g) if with_plus else None
    act = (lambda t: F.relu(t)) if blk.slope == 0.0 else (lambda t: F.leaky_relu(t, blk.slope))
    cx = x.clone().requires_grad_()
    cp = p.clone().requires_grad_() if with_plus else None
    ref_params = [t.detach().clone().requires_grad_() for t in (blk.conv1.weight, blk.conv1.bias, blk.conv2.weight, blk.conv2.bias)]
    t = act(F.conv2d(act(cx) if start_relu else cx, ref_params[0], ref_params[1], padding=1))
    t = F.conv2d(t, ref_params[2], ref_params[3], padding=1)
    y_ref = cx + (act(t) if end_relu else t)
    if with_plus:
        y_ref = y_ref + cp
    R = torch.randn(y_ref.shape, generator=g)
    (y_ref * R).sum().backward()
    blk = blk.to("cuda:0")
    dx = dev(x).requires_grad_()
    dp = dev(p).requires_grad_() if with_plus else None
    y = blk(dx, plus=dp)
    assert type(y.grad_fn).__name__ == "_ResBlockBackward"
    close(y, y_ref.detach(), what="forward")
    (y * dev(R)).sum().backward()
    close(dx.grad, cx.grad, what="grad x")
    if with_plus:
        close(dp.grad, cp.grad, what="grad plus")
    for name, a, b in zip(("w1", "b1", "w2", "b2"), (blk.conv1.weight, blk.conv1.bias, blk.conv2.weight, blk.conv2.bias), ref_params):
        close(a.grad, b.grad, rel=5e-5, what="grad " + name)


def test_residual_and_gate_gradients():
    from deepvideocodec_amd import autograd as A
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 32, 20, 28, generator=g)
    w = torch.randn(64, 32, 3, 3, generator=g) / 17
    b = torch.randn(64, generator=g)
    r1 = torch.randn(2, 64, 20, 28, generator=g)
    r2 = torch.randn(2, 64, 20, 28, generator=g)
    sc = torch.rand(2, 64, generator=g)
    c = [t.clone().requires_grad_() for t in (x, w, b, r1, r2, sc)]
    y_ref = F.relu(F.conv2d(c[0], c[1], c[2], padding=1)) + c[3] * c[5][:, :, None, None] + c[4]
    R = torch.randn(y_ref.shape, generator=g)
    (y_ref * R).sum().backward()
    d = [dev(t).requires_grad_() for t in (x, w, b, r1, r2, sc)]
    y = A.conv2d(d[0], d[1], d[2], out_slope=0.0, residual=d[3], res_scale=d[5].reshape(-1), residual2=d[4])
    close(y, y_ref.detach(), what="forward")
    (y * dev(R)).sum().backward()
    for name, a, cc in zip(("x", "w", "b", "r1", "r2", "scale"), d, c):
        close(a.grad, cc.grad, what=f"grad {name}")


def test_warp_gradients_match_grid_sample():
    from deepvideocodec_amd import autograd as A
    g = torch.Generator().manual_seed(2)
    feat = torch.randn(2, 6, 33, 47, generator=g)
    flow = torch.randn(2, 2, 33, 47, generator=g) * 4
    flow[0, :, :3] = 60.0     # pushed far outside: clipped coordinates carry no flow gradient
    cf, cl = feat.clone().requires_grad_(), flow.clone().requires_grad_()
    y_ref = O.flow_warp(cf, cl)
    R = torch.randn(y_ref.shape, generator=g)
    (y_ref * R).sum().backward()
    df, dl = dev(feat).requires_grad_(), dev(flow).requires_grad_()
    y = A.flow_warp(df, dl)
    (y * dev(R)).sum().backward()
    close(df.grad, cf.grad, what="grad src")
    close(dl.grad, cl.grad, rel=1e-3, what="grad flow")


@pytest.mark.parametrize("channels,shift", [(6, (0.0, 0.0)), (3, (9.3, -5.6)), (9, (-70.0, 2.5))])
def test_warp_gradients_smooth_flow(channels, shift):
    """Smooth flows (the codec's case): the taps of an 8 x 32 tile fall into a small source window, which
    takes the LDS path of the adjoint kernel; ragged sizes, channel counts off the group of four, a uniform
    shift, and a shift that pushes everything against the border."""
    from deepvideocodec_amd import autograd as A
    g = torch.Generator().manual_seed(12 + channels)
    H, W = 45, 83
    feat = torch.randn(2, channels, H, W, generator=g)
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    flow = torch.stack([2.0 * torch.sin(xx / 9.0 + yy / 14.0) + shift[0], 1.5 * torch.cos(xx / 11.0 - yy / 7.0) + shift[1]])
    flow = flow.unsqueeze(0).repeat(2, 1, 1, 1) + 0.05 * torch.randn(2, 2, H, W, generator=g)
    cf, cl = feat.clone().requires_grad_(), flow.clone().requires_grad_()
    y_ref = O.flow_warp(cf, cl)
    R = torch.randn(y_ref.shape, generator=g)
    (y_ref * R).sum().backward()
    df, dl = dev(feat).requires_grad_(), dev(flow).requires_grad_()
    y = A.flow_warp(df, dl)
    close(y, y_ref.detach(), what="forward")
    (y * dev(R)).sum().backward()
    close(df.grad, cf.grad, what="grad src")
    close(dl.grad, cl.grad, rel=1e-3, what="grad flow")


@pytest.mark.parametrize("sigma", [3.0, 6.0, 12.0, 30.0])
def test_warp_gradients_noisy_flow(sigma):
    """Noisy motion (the decoded motion of an untrained codec): the source window of an 8 x 32 tile grows with the
    noise and the adjoint kernel takes fewer channels per pass through its LDS -- two (sigma 3), one (sigma 6), one
    with the source values gathered from memory (sigma 12) -- or scatters directly (sigma 30); seven channels leave a
    ragged last pass."""
    from deepvideocodec_amd import autograd as A
    g = torch.Generator().manual_seed(int(sigma))
    feat = torch.randn(2, 7, 64, 96, generator=g)
    flow = torch.randn(2, 2, 64, 96, generator=g) * sigma
    cf, cl = feat.clone().requires_grad_(), flow.clone().requires_grad_()
    y_ref = O.flow_warp(cf, cl)
    R = torch.randn(y_ref.shape, generator=g)
    (y_ref * R).sum().backward()
    df, dl = dev(feat).requires_grad_(), dev(flow).requires_grad_()
    y = A.flow_warp(df, dl)
    (y * dev(R)).sum().backward()
    close(df.grad, cf.grad, what="grad src")
    close(dl.grad, cl.grad, rel=1e-3, what="grad flow")


def test_warp_gradient_keeps_non_finite_values():
    """The adjoint's LDS sums are fixed point, scaled by the tile's largest |gradient|; a tile whose gradients are not
    all finite must fall back to float sums for that channel, so that a NaN or Inf still reaches the source gradient
    (and only the elements that pixel touches)."""
    from deepvideocodec_amd import ops
    g = torch.Generator().manual_seed(21)
    src = torch.randn(1, 5, 40, 64, generator=g)
    flow = torch.randn(1, 2, 40, 64, generator=g) * 1.5
    go = torch.randn(1, 5, 40, 64, generator=g)
    go[0, 1, 17, 33] = float("nan")
    go[0, 3, 5, 9] = float("inf")
    cs = src.clone().requires_grad_()
    (O.flow_warp(cs, flow) * go).sum().backward()     # (nan * 1 -> nan, inf * weight -> inf in the reference's adjoint)
    gs, _ = ops.flow_warp_backward(dev(src), dev(flow), dev(go))
    gs = gs.cpu()
    assert torch.equal(torch.isnan(gs), torch.isnan(cs.grad))
    assert torch.equal(torch.isinf(gs), torch.isinf(cs.grad))
    fin = torch.isfinite(cs.grad)
    assert float((gs[fin] - cs.grad[fin]).abs().max()) <= 2e-5 * float(cs.grad[fin].abs().max())


def test_pool_and_upsample_gradients():
    from deepvideocodec_amd import autograd as A
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 5, 18, 26, generator=g)
    for name, ours, ref in (
        ("avg", lambda t: A.avg_pool2(t, 0.5), lambda t: O.bilinear_down2(t) * 0.5 if hasattr(O, "bilinear_down2") else F.avg_pool2d(t, 2) * 0.5),
        ("max", A.max_pool2, lambda t: F.max_pool2d(t, 2)),
        ("up2", lambda t: A.bilinear_up2(t, 2.0), lambda t: F.interpolate(t, scale_factor=2, mode="bilinear", align_corners=False) * 2.0),
    ):
        cx = x.clone().requires_grad_()
        y_ref = ref(cx)
        R = torch.randn(y_ref.shape, generator=g)
        (y_ref * R).sum().backward()
        dx = dev(x).requires_grad_()
        y = ours(dx)
        close(y, y_ref.detach(), what=name + " forward")
        (y * dev(R)).sum().backward()
        close(dx.grad, cx.grad, what=name + " grad")


def test_factorized_prior_gradients():
    """Likelihood of z + noise w.r.t. z and all 58 parameters per channel, against the oracle's
    torch restatement of compressai's EntropyBottleneck under CPU autograd."""
    from deepvideocodec_amd.models.entropy_models import EntropyBottleneck
    from oracle import entropy_oracle as EO
    torch.manual_seed(0)
    C_ = 6
    ref = EO.EntropyBottleneck(C_)
    with torch.no_grad():   # move the parameters off their symmetric init
        for p in ref.parameters():
            p.add_(torch.randn_like(p) * 0.2)
    g = torch.Generator().manual_seed(1)
    z = torch.randn(2, C_, 5, 7, generator=g) * 3
    noise = torch.rand(z.shape, generator=g) - 0.5
    R = torch.randn(z.shape, generator=g)

    cz = z.clone().requires_grad_()
    vals = (cz + noise).permute(1, 0, 2, 3).reshape(C_, 1, -1)
    lik = EO.lower_bound(ref._likelihood(vals), 1e-9).reshape(C_, 2, 5, 7).permute(1, 0, 2, 3)
    (-(torch.log2(lik)) * R).sum().backward()

    net = EntropyBottleneck(C_)
    net.load_state_dict(ref.state_dict())
    net = net.cuda().train()
    dz = dev(z).requires_grad_()
    from deepvideocodec_amd import autograd as A
    out, dlik = A.eb_likelihood(dz, net.packed_parameters(differentiable=True), net._medians_flat(), dev(noise))
    close(dlik, lik.detach(), what="likelihood")
    close(out, (z + noise), what="outputs")
    (-(torch.log2(dlik)) * dev(R)).sum().backward()
    close(dz.grad, cz.grad, what="grad z")
    for (name, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        if name == "quantiles":
            continue
        close(p.grad, q.grad, rel=5e-4, what=f"grad {name}")


def _dual_prior_reference(y, means, scales, spatial_fn, noise):
    """reference video_model.py:160-232 with explicit noise (torch ops, STE round)."""
    from oracle import entropy_oracle as EO
    n, c, h, w = y.shape
    ii = torch.arange(h).view(h, 1)
    jj = torch.arange(w).view(1, w)
    m0 = (((ii + jj) % 2) == 0).float().view(1, 1, h, w)
    m1 = 1.0 - m0

    def ste(x):
        return (torch.round(x) - x).detach() + x

    def proc(y_, mu, sg, m):
        mh, sh = mu * m, sg * m
        return ste((y_ - mh) * m) + mh, mh, sh

    y0, y1 = y.chunk(2, 1)
    mu0, mu1 = means.chunk(2, 1)
    s0, s1 = scales.chunk(2, 1)
    yh00, mh00, sh00 = proc(y0, mu0, s0, m0)
    yh11, mh11, sh11 = proc(y1, mu1, s1, m1)
    spatial = spatial_fn(torch.cat((yh00, yh11, means, scales), 1))
    a0, b0, a1, b1 = spatial.chunk(4, 1)
    yh01, mh01, sh01 = proc(y0, a0, b0, m1)
    yh10, mh10, sh10 = proc(y1, a1, b1, m0)
    y_hat = torch.cat((yh00 + yh01, yh11 + yh10), 1)
    means_hat = torch.cat((mh00 + mh01, mh11 + mh10), 1)
    scales_hat = torch.cat((sh00 + sh01, sh11 + sh10), 1)
    gc = EO.GaussianConditional(None)
    lik = EO.lower_bound(gc._likelihood(y + noise, scales_hat, means_hat), 1e-9)
    return y_hat, lik


def test_dual_prior_gradients():
    from deepvideocodec_amd import autograd as A
    g = torch.Generator().manual_seed(7)
    n, c, h, w = 2, 8, 6, 10
    y = torch.randn(n, c, h, w, generator=g) * 4
    means = torch.randn(n, c, h, w, generator=g)
    scales = torch.randn(n, c, h, w, generator=g) * 0.8 + 0.5      # some below the .11 bound
    wsp = torch.randn(2 * c, 3 * c, 3, 3, generator=g) / 10
    bsp = torch.randn(2 * c, generator=g) * 0.3 + 0.4
    noise = torch.rand(n, c, h, w, generator=g) - 0.5
    R1 = torch.randn(n, c, h, w, generator=g)
    R2 = torch.randn(n, c, h, w, generator=g)

    cpu = [t.clone().requires_grad_() for t in (y, means, scales, wsp, bsp)]
    yh, lik = _dual_prior_reference(cpu[0], cpu[1], cpu[2], lambda t: F.conv2d(t, cpu[3], cpu[4], padding=1), noise)
    ((yh * R1).sum() + (-(torch.log2(lik)) * R2).sum()).backward()

    d = [dev(t).requires_grad_() for t in (y, means, scales, wsp, bsp)]
    yhat1 = A.dual_prior_pass1(d[0], d[1], d[2])
    spatial = A.conv2d((yhat1, d[1], d[2]), d[3], d[4])
    dyh, dlik = A.dual_prior_pass2(d[0], d[1], d[2], spatial, dev(noise))
    close(dyh, yh.detach(), what="y_hat")
    close(dlik, lik.detach(), rel=1e-4, what="likelihood")
    ((dyh * dev(R1)).sum() + (-(torch.log2(dlik)) * dev(R2)).sum()).backward()
    for name, a, b in zip(("y", "means", "scales", "w_spatial", "b_spatial"), d, cpu):
        close(a.grad, b.grad, rel=1e-3, what=f"grad {name}")


def test_se_gate_gradients():
    from deepvideocodec_amd import autograd as A
    g = torch.Generator().manual_seed(8)
    x = torch.randn(2, 32, 9, 13, generator=g)
    fc1 = torch.randn(2, 32, generator=g) / 4
    fc2 = torch.randn(32, 2, generator=g)
    R = torch.randn(2, 32, generator=g)
    c = [t.clone().requires_grad_() for t in (x, fc1, fc2)]
    gate = torch.sigmoid(torch.relu(c[0].mean(dim=(-1, -2)) @ c[1].t()) @ c[2].t())
    (gate * R).sum().backward()
    d = [dev(t).requires_grad_() for t in (x, fc1, fc2)]
    dg = A.se_gate(d[0], d[1], d[2])
    close(dg.view(2, 32), gate.detach(), what="gate")
    (dg.view(2, 32) * dev(R)).sum().backward()
    for name, a, b in zip(("x", "fc1", "fc2"), d, c):
        close(a.grad, b.grad, what=f"grad {name}")
