"""torch.autograd bindings of the HIP kernels: every Function's forward and backward launch only
entry points of libdvc_hip.so (through ops.py); torch supplies the tape, the memory and the
accumulation of gradients that reach one tensor from several consumers.

The reference gets these gradients from torch.autograd over ATen ops (loss.backward() at
dmc/train.py:329-336); the functions here are the adjoints of the fused forward kernels, so a
training step runs the same HIP forward as inference plus the kernels in csrc/backward.hip and
csrc/conv_wgrad.hip.
"""
from typing import Optional, Sequence

import torch

from . import ops

Tensor = torch.Tensor


def _rows_ok(t: Tensor) -> Tensor:
    """Gradients arrive with arbitrary strides (expanded, transposed ...): make rows contiguous."""
    if t.dim() == 4 and (t.size(3) == 1 or t.stride(3) == 1) and (t.size(2) == 1 or t.stride(2) == t.size(3)) \
            and t.stride(1) >= t.size(2) * t.size(3) and (t.size(0) == 1 or t.stride(0) >= t.size(1) * t.stride(1)):
        return t
    return t.contiguous()


class _Conv2d(torch.autograd.Function):
    """y = act_out(conv(act_in(cat(srcs))) + bias), optionally PixelShuffle(2)-ed (nn.Conv2d +
    LeakyReLU / ReLU / PixelShuffle / torch.cat of the reference graph)."""

    @staticmethod
    def forward(ctx, weight, bias, stride, in_slope, out_slope, pixel_shuffle, packed, dgrad_packed, precision,
                dgrad_precision, residual, residual2, *srcs):
        # (residual operands only for a layer without output activation: the sum's adjoint is then the identity on
        # every operand and needs nothing saved; see conv2d() below)
        out = ops.conv2d(srcs, weight, bias, stride=stride, in_slope=in_slope, out_slope=out_slope,
                         residual=residual, residual2=residual2,
                         pixel_shuffle=pixel_shuffle, packed=packed, precision=precision)
        ctx.cfg = (stride, in_slope, out_slope, pixel_shuffle, bias is not None)
        ctx.dgrad_packed = dgrad_packed     # the layer's cached data-gradient weight image (or None: packed on demand)
        ctx.dgrad_precision = dgrad_precision
        ctx.save_for_backward(weight, out if out_slope is not None else None, *srcs)
        return out

    @staticmethod
    def backward(ctx, g):
        stride, in_slope, out_slope, ps, has_bias = ctx.cfg
        weight, out, *srcs = ctx.saved_tensors
        cout, cin, k, _ = weight.shape
        n, _, h, w = srcs[0].shape
        pad = k // 2
        ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
        g = _rows_ok(g)
        if out_slope is None and not ps and g.is_contiguous():
            g_pre = g
        else:
            g_pre = ops.act_grad(g, out, 1.0 if out_slope is None else out_slope, ps, (n, cout, ho, wo))
        need = ctx.needs_input_grad
        gw = gb = None
        if need[0] and has_bias and need[1]:     # the bias gradient rides along with the weight gradient
            gw, gb = ops.conv2d_wgrad(srcs, g_pre, weight.shape, stride, in_slope, with_bias=True)
        elif need[0]:
            gw = ops.conv2d_wgrad(srcs, g_pre, weight.shape, stride, in_slope)
        elif has_bias and need[1]:
            gb = ops.plane_dot(g_pre, None, sum_batch=True)
        gsrcs = [None] * len(srcs)
        if any(need[12:]):
            # data gradient = stride-1 convolution of (zero-upsampled) g_pre with the flipped,
            # transposed kernel -- the same MFMA kernel as the forward pass (and in the same operand mode)
            gin = ops.zero_upsample2(g_pre, h, w) if stride == 2 else g_pre
            image = ctx.dgrad_packed if ctx.dgrad_packed is not None else dgrad_weight_image(weight, ctx.dgrad_precision)
            gx = ops.conv2d(gin, (cin, cout, k, k), None, stride=1, packed=image, precision=ctx.dgrad_precision)
            c0 = 0
            for i, s in enumerate(srcs):
                c1 = c0 + s.size(1)
                if need[12 + i]:
                    gs = gx[:, c0:c1]
                    gsrcs[i] = ops.mul_act_grad(gs, s, in_slope) if in_slope is not None else gs
                c0 = c1
        return (gw, gb, None, None, None, None, None, None, None, None, g if need[10] else None,
                g if need[11] else None, *gsrcs)


def dgrad_weight_image(weight: Tensor, precision: int = 0) -> Tensor:
    """The kernel-layout image of the weight of `weight`'s data-gradient convolution (taps flipped, channel axes
    swapped) in operand mode `precision`."""
    if precision == 0:
        return ops.pack_conv_weight_dgrad(weight.detach().contiguous())
    return ops.pack_conv_weight(weight.detach().flip(2, 3).transpose(0, 1).contiguous(), 1, precision=precision)


def conv2d(srcs, weight, bias, stride=1, in_slope=None, out_slope=None, residual=None, res_scale=None,
           residual2=None, pixel_shuffle=False, packed=None, dgrad_packed=None, precision=0, dgrad_precision=0):
    """Differentiable ops.conv2d.  Behind an output activation, or with a gate, the residual adds are taken out of the
    convolution epilogue (their adjoint needs the pre-residual activation) and run as add_scaled.  `precision` / `dgrad_precision`: operand
    mode (ops.PRECISIONS) of the forward and of the data-gradient convolution -- `packed` / `dgrad_packed` must be images
    in those modes; the weight gradient is always formed in fp32."""
    if isinstance(srcs, torch.Tensor):
        srcs = (srcs,)
    if residual is not None and out_slope is None and res_scale is None and not pixel_shuffle:
        # no activation between the convolution and the sum (the second convolution of a ResBlock,
        # dmc/models/layers.py:75-81): the sum stays in the convolution's epilogue, as on the coding path
        return _Conv2d.apply(weight, bias, stride, in_slope, out_slope, pixel_shuffle, packed, dgrad_packed, precision,
                             dgrad_precision, residual, residual2, *srcs)
    y = _Conv2d.apply(weight, bias, stride, in_slope, out_slope, pixel_shuffle, packed, dgrad_packed, precision,
                      dgrad_precision, None, None, *srcs)
    if residual is not None:
        y = add_scaled(y, residual, res_scale, residual2)
    return y


class _ResBlock(torch.autograd.Function):
    """x + last(conv2(act(conv1(first(x))))) (+ plus) as ONE node of the tape (dmc/models/layers.py:59-81).  Node by node,
    x collects two gradients -- the block's output gradient along the sum and the first convolution's data gradient --
    which autograd adds in a pass of its own; here the sum is formed where the data gradient is finished: inside
    mul_act_grad when the block starts with an activation, in the data-gradient convolution's epilogue when it does
    not.  The saved tensors and every other kernel are those of the node-by-node form."""

    @staticmethod
    def forward(ctx, x, plus, w1, b1, w2, b2, slope_in, slope_mid, slope_out, packed1, packed2, dpacked1, dpacked2,
                pf1, pd1, pf2, pd2):
        t = ops.conv2d(x, w1, b1, in_slope=slope_in, out_slope=slope_mid, packed=packed1, precision=pf1)
        if slope_out is None:
            y = ops.conv2d(t, w2, b2, residual=x, residual2=plus, packed=packed2, precision=pf2)
            u = None
        else:
            u = ops.conv2d(t, w2, b2, out_slope=slope_out, packed=packed2, precision=pf2)
            y = ops.add_scaled(u, x, None, plus)
        ctx.cfg = (slope_in, slope_mid, slope_out, pd1, pd2)
        ctx.images = (dpacked1, dpacked2)
        ctx.save_for_backward(x, t, u, w1, w2)
        return y

    @staticmethod
    def backward(ctx, g):
        slope_in, slope_mid, slope_out, pd1, pd2 = ctx.cfg
        dpacked1, dpacked2 = ctx.images
        x, t, u, w1, w2 = ctx.saved_tensors
        need = ctx.needs_input_grad
        n, ch, h, w = x.shape
        mid = w1.shape[0]
        g = _rows_ok(g)
        # second convolution
        if slope_out is None and g.is_contiguous():
            g2 = g
        else:
            g2 = ops.act_grad(g, u, 1.0 if slope_out is None else slope_out, False, (n, ch, h, w))
        gw2 = gb2 = gw1 = gb1 = None
        if need[4] or need[5]:
            gw2, gb2 = ops.conv2d_wgrad((t,), g2, w2.shape, 1, None, with_bias=True)
        image2 = dpacked2 if dpacked2 is not None else dgrad_weight_image(w2, pd2)
        gt = ops.conv2d(g2, (mid, ch, 3, 3), None, stride=1, packed=image2, precision=pd2)
        # first convolution (its output activation is the block's middle one)
        g1 = ops.act_grad(gt, t, slope_mid, False, (n, mid, h, w))
        if need[2] or need[3]:
            gw1, gb1 = ops.conv2d_wgrad((x,), g1, w1.shape, 1, slope_in, with_bias=True)
        gx = None
        if need[0]:
            image1 = dpacked1 if dpacked1 is not None else dgrad_weight_image(w1, pd1)
            if slope_in is None:
                gx = ops.conv2d(g1, (ch, mid, 3, 3), None, stride=1, packed=image1, precision=pd1, residual=g)
            else:
                gx = ops.conv2d(g1, (ch, mid, 3, 3), None, stride=1, packed=image1, precision=pd1)
                gx = ops.mul_act_grad(gx, x, slope_in, out=gx, plus=g)
        return (gx, g if need[1] else None, gw1 if need[2] else None, gb1 if need[3] else None,
                gw2 if need[4] else None, gb2 if need[5] else None,
                None, None, None, None, None, None, None, None, None, None, None)


def res_block(x, plus, conv1, conv2, slope_in, slope_mid, slope_out):
    """Differentiable ResBlock over two Conv2d layers of the mirror (`plus`: an extra tensor added to the result)."""
    pf1, pd1 = conv1._train_precisions()
    pf2, pd2 = conv2._train_precisions()
    return _ResBlock.apply(x, plus, conv1.weight, conv1.bias, conv2.weight, conv2.bias, slope_in, slope_mid, slope_out,
                           conv1.packed_weight(pf1), conv2.packed_weight(pf2), conv1.dgrad_weight(pd1),
                           conv2.dgrad_weight(pd2), pf1, pd1, pf2, pd2)


class _AddScaled(torch.autograd.Function):
    """a + b * scale[n, c] (+ c2)."""

    @staticmethod
    def forward(ctx, a, b, scale, c2):
        ctx.save_for_backward(b if scale is not None else None, scale)
        ctx.has_c2 = c2 is not None
        return ops.add_scaled(a, b, scale, c2)

    @staticmethod
    def backward(ctx, g):
        b, scale = ctx.saved_tensors
        g = _rows_ok(g)
        need = ctx.needs_input_grad
        gb = gs = None
        if scale is None:
            gb = g if need[1] else None
        else:
            if need[1]:
                gb = ops.scale_planes(g, scale)
            if need[2]:
                gs = ops.plane_dot(g, b).view_as(scale)
        return (g if need[0] else None, gb, gs, g if (ctx.has_c2 and need[3]) else None)


def add_scaled(a, b, scale=None, c2=None):
    return _AddScaled.apply(a, b, scale, c2)


class _FlowWarp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, flow):
        ctx.save_for_backward(src, flow)
        return ops.flow_warp(src, flow)

    @staticmethod
    def backward(ctx, g):
        src, flow = ctx.saved_tensors
        gs, gf = ops.flow_warp_backward(src, flow, _rows_ok(g), ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return gs, gf


def flow_warp(src, flow):
    return _FlowWarp.apply(src, flow)


class _AvgPool2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale):
        ctx.cfg = (x.size(2), x.size(3), scale)
        return ops.avg_pool2(x, scale)

    @staticmethod
    def backward(ctx, g):
        h, w, scale = ctx.cfg
        return ops.avg_pool2_backward(_rows_ok(g), h, w, scale), None


def avg_pool2(x, scale=1.0):
    return _AvgPool2.apply(x, scale)


class _MaxPool2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return ops.max_pool2(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return ops.max_pool2_backward(x, _rows_ok(g))


def max_pool2(x):
    return _MaxPool2.apply(x)


class _BilinearUp2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale):
        ctx.cfg = (x.size(2), x.size(3), scale)
        return ops.bilinear_up2(x, scale)

    @staticmethod
    def backward(ctx, g):
        h, w, scale = ctx.cfg
        return ops.bilinear_up2_backward(_rows_ok(g), h, w, scale), None


def bilinear_up2(x, scale=1.0):
    return _BilinearUp2.apply(x, scale)


class _SEGate(torch.autograd.Function):
    """gate[n, c] = sigmoid(fc2 relu(fc1 mean_hw(x)))  (reference layers.py:267-280).  The plane
    means run on the HIP reduction kernels; the adjoint of the two tiny FC layers (C x C/16
    matrices on an (N, C) vector) is evaluated with torch on those few numbers."""

    @staticmethod
    def forward(ctx, x, fc1, fc2):
        ctx.save_for_backward(x, fc1, fc2)
        return ops.se_gate(x, fc1, fc2)

    @staticmethod
    def backward(ctx, g):
        x, fc1, fc2 = ctx.saved_tensors
        n, c, h, w = x.shape
        mean = (ops.plane_dot(x) / float(h * w)).view(n, c)
        with torch.enable_grad():
            m = mean.detach().requires_grad_()
            a, b = fc1.detach().requires_grad_(), fc2.detach().requires_grad_()
            gate = torch.sigmoid(torch.relu(m @ a.t()) @ b.t())
            gm, ga, gb = torch.autograd.grad(gate, (m, a, b), g.reshape(n, c))
        gx = (gm / float(h * w)).view(n, c, 1, 1).expand(n, c, h, w) if ctx.needs_input_grad[0] else None
        return gx, ga, gb


def se_gate(x, fc1, fc2):
    return _SEGate.apply(x, fc1, fc2)


class _RoundSTE(torch.autograd.Function):
    """round(z - median) + median with the identity as gradient (reference utils.py:149-152 used at
    video_model.py:222-224)."""

    @staticmethod
    def forward(ctx, z, packed, medians):
        return ops.eb_forward(z, packed, medians, want_likelihood=False)[0]

    @staticmethod
    def backward(ctx, g):
        return g, None, None


def round_ste(z, packed, medians):
    return _RoundSTE.apply(z, packed, medians)


class _EBLikelihood(torch.autograd.Function):
    """Factorized-prior likelihood of z + noise (compressai EntropyBottleneck.forward, training):
    returns (z + noise, likelihood); differentiable w.r.t. z and the packed (C, 58) parameters."""

    @staticmethod
    def forward(ctx, z, packed, medians, noise):
        out, lik, _ = ops.eb_forward(z, packed, medians, noise=noise)
        ctx.save_for_backward(out, packed)
        return out, lik

    @staticmethod
    def backward(ctx, g_out, g_lik):
        out, packed = ctx.saved_tensors
        gx, gp = ops.eb_backward(out, packed, _rows_ok(g_lik), want_gx=ctx.needs_input_grad[0])
        if gx is not None and g_out is not None:
            gx = gx + g_out
        return gx, gp, None, None


def eb_likelihood(z, packed, medians, noise):
    return _EBLikelihood.apply(z, packed, medians, noise)


class _DualPriorPass1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, means, scales):
        return ops.dual_prior_pass1(y, means, scales)

    @staticmethod
    def backward(ctx, g):
        return ops.dual_prior_pass1_backward(_rows_ok(g)), None, None


def dual_prior_pass1(y, means, scales):
    return _DualPriorPass1.apply(y, means, scales)


class _DualPriorPass2(torch.autograd.Function):
    """-> (y_hat, likelihood) of the second checkerboard pass (video_model.py:192-232)."""

    @staticmethod
    def forward(ctx, y, means, scales, spatial, noise):
        r = ops.dual_prior_pass2(y, means, scales, spatial, noise=noise, want=("y_hat", "likelihood"))
        ctx.save_for_backward(y, means, scales, spatial, noise)
        return r["y_hat"], r["likelihood"]

    @staticmethod
    def backward(ctx, g_yhat, g_lik):
        y, means, scales, spatial, noise = ctx.saved_tensors
        gy, gm, gs, gsp = ops.dual_prior_pass2_backward(
            y, means, scales, spatial, noise, _rows_ok(g_yhat) if g_yhat is not None else None,
            _rows_ok(g_lik) if g_lik is not None else None)
        return gy, gm, gs, gsp, None


def dual_prior_pass2(y, means, scales, spatial, noise):
    return _DualPriorPass2.apply(y, means, scales, spatial, noise)


class _LikelihoodBits(torch.autograd.Function):
    """(N,) sum over (C, H, W) of -log2(likelihood): the rate term of dmc/train.py:83."""

    @staticmethod
    def forward(ctx, lik):
        ctx.save_for_backward(lik)
        return ops.likelihood_bits(lik)

    @staticmethod
    def backward(ctx, g):
        (lik,) = ctx.saved_tensors
        return ops.likelihood_bits_backward(lik, g.contiguous().float())


def likelihood_bits(lik):
    return _LikelihoodBits.apply(lik)


class _SqErrSum(torch.autograd.Function):
    """(N,) sum over (C, H, W) of (a - b)^2: the distortion term of dmc/train.py:114-143."""

    @staticmethod
    def forward(ctx, a, b):
        ctx.save_for_backward(a, b)
        return ops.sq_err_sum(a, b)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        need = ctx.needs_input_grad
        ga = ops.sq_err_backward(a, b, g.contiguous().float()) if (need[0] or need[1]) else None
        return (ga if need[0] else None), (-ga if need[1] else None)


def sq_err_sum(a, b):
    return _SqErrSum.apply(a, b)
