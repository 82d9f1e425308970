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


def test_conv2d_wgrad_large_reduction():
    """A reduction long enough for many pixel slabs per block (256x256 crops, batch 4)."""
    from deepvideocodec_amd import autograd as A
    g = torch.Generator().manual_seed(3)
    x = torch.randn(4, 64, 256, 256, generator=g)
    w = torch.randn(64, 64, 3, 3, generator=g) / 24
    cx, cw = x.clone().requires_grad_(), w.clone().requires_grad_()
    y_ref = F.conv2d(cx, cw, None, padding=1)
    R = torch.randn(y_ref.shape, generator=g) / 64
    (y_ref * R).sum().backward()
    gx, gw = dev(x).requires_grad_(), dev(w).requires_grad_()
    y = A.conv2d(gx, gw, None)
    (y * dev(R)).sum().backward()
    close(gw.grad, cw.grad, rel=5e-4, what="grad weight")
    close(gx.grad, cx.grad, what="grad x")


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
