"""Building blocks of the DMC graph, executed by the HIP kernels of this package.

Same block inventory, child names and parameter shapes as the reference's
dmc/models/layers.py (so state dicts interchange), but each block is a launch
plan rather than a stack of ATen calls: activations, residual adds, channel
concats, PixelShuffle and the squeeze-excite scaling are folded into the
convolution launches that produce / consume them (see DvcConv2d in
include/dvc_hip.h).  A "tensor or tuple of tensors" input means an implicit
channel concat that is never materialised.
"""
import math
from typing import Optional, Sequence, Tuple, Union

import torch
import torch.nn as nn

from .. import autograd as A
from .. import ops

Tensor = torch.Tensor
Srcs = Union[Tensor, Sequence[Tensor]]


class Conv2d(nn.Module):
    """Parameter holder + launcher for one convolution (weight OIHW, bias)."""

    def __init__(self, in_ch: int, out_ch: int, kernel_size: int, stride: int = 1):
        super().__init__()
        self.in_channels, self.out_channels = in_ch, out_ch
        self.kernel_size, self.stride = kernel_size, stride
        self.weight = nn.Parameter(torch.empty(out_ch, in_ch, kernel_size, kernel_size))
        self.bias = nn.Parameter(torch.empty(out_ch))
        # same distribution family as the reference's final init (video_model.py:508-513)
        nn.init.xavier_normal_(self.weight, math.sqrt(2))
        nn.init.constant_(self.bias, 0.01)

        self._packed = {}         # compute mode -> (key, kernel-layout image of `weight`)
        # inference-only option: 16-bit operand arithmetic, fp32 accumulation.  See DMC.set_precision().
        self.precision = 0      # ops.PRECISIONS: 0 exact fp32, 1 bf16, 2 bf16 hi + lo split, 3 fp16 hi + lo split
        # the same for the forward and data-gradient convolutions under autograd (DMC.set_precision(..., train=True))
        self.train_precision = 0

    def _precision(self) -> int:
        # layers with a handful of output channels (the flow and RGB heads) run on the vector ALUs in exact fp32
        return int(self.precision) if self.out_channels > 4 else 0

    def _train_precisions(self):
        """(forward, data-gradient) operand modes under autograd: the data gradient is a convolution with
        `in_channels` outputs, so the rule about a handful of output channels applies to that count."""
        p = int(self.train_precision)
        return (p if self.out_channels > 4 else 0, p if self.in_channels > 4 else 0)

    def packed_weight(self, precision: int = 0) -> Tensor:
        """The slab-layout copy of the weight the kernel reads (one per compute mode), rebuilt when
        the parameter is modified in place (optimizer step, load_state_dict) or moved (`.to(device)`)."""
        w = self.weight
        key = (w.data_ptr(), w._version, w.device)
        hit = self._packed.get(precision)
        if hit is None or hit[0] != key:
            hit = (key, ops.pack_conv_weight(w.detach().contiguous(), self.stride, precision=precision))
            self._packed[precision] = hit
        return hit[1]

    def dgrad_weight(self, precision: int = 0) -> Tensor:
        """The kernel-layout image of the weight of this layer's data-gradient convolution (training): built once
        per weight version and operand mode, so the P frames of one step share it."""
        w = self.weight
        key = (w.data_ptr(), w._version, w.device)
        hit = self._packed.get(("dgrad", precision))
        if hit is None or hit[0] != key:
            hit = (key, A.dgrad_weight_image(w, precision))
            self._packed[("dgrad", precision)] = hit
        return hit[1]

    def forward(self, x: Srcs, in_slope=None, out_slope=None, residual=None, res_scale=None, residual2=None,
                pixel_shuffle=False, out=None, plane_sums=False):
        """`plane_sums=True` (inference only): returns (out, sums), see ops.conv2d."""
        if torch.is_grad_enabled():
            # training path: same forward kernel, on the autograd tape (residual adds un-fused where an output
            # activation or a gate sits in front of them)
            if out is not None:
                raise ValueError("Conv2d: `out=` is an inference-only option (call under torch.no_grad())")
            pf, pd = self._train_precisions()
            return A.conv2d(x, self.weight, self.bias, stride=self.stride, in_slope=in_slope, out_slope=out_slope,
                            residual=residual, res_scale=res_scale, residual2=residual2,
                            pixel_shuffle=pixel_shuffle, packed=self.packed_weight(pf), dgrad_packed=self.dgrad_weight(pd),
                            precision=pf, dgrad_precision=pd)
        return ops.conv2d(x, self.weight.detach(), self.bias.detach(), stride=self.stride, in_slope=in_slope,
                          out_slope=out_slope, residual=residual, res_scale=res_scale, residual2=residual2,
                          pixel_shuffle=pixel_shuffle, out=out, packed=self.packed_weight(self._precision()),
                          precision=self._precision(), plane_sums=plane_sums)


def conv1x1(in_ch, out_ch, stride=1):
    return Conv2d(in_ch, out_ch, 1, stride)


def conv3x3(in_ch, out_ch, stride=1):
    return Conv2d(in_ch, out_ch, 3, stride)


class Indexed(nn.Module):
    """Children registered under integer names, like nn.Sequential with gaps
    (the reference's Sequentials interleave parameter-free activations)."""

    def __init__(self, **children):
        super().__init__()
        for k, m in children.items():
            self.add_module(k.lstrip("_"), m)

    def __getitem__(self, i) -> nn.Module:
        return self._modules[str(i)]


class SubpelConv(Indexed):
    """conv (child `0`) -> PixelShuffle(2) (reference: layers.py:45-56); the shuffle is the
    conv kernel's store pattern."""

    def __init__(self, in_ch, out_ch, kernel_size, r=2):
        assert r == 2
        super().__init__(_0=Conv2d(in_ch, out_ch * r * r, kernel_size))

    def forward(self, x: Srcs, out_slope=None, out=None) -> Tensor:
        return self[0](x, out_slope=out_slope, pixel_shuffle=True, out=out)


def subpel_conv1x1(in_ch, out_ch, r=2):
    return SubpelConv(in_ch, out_ch, 1, r)


def subpel_conv3x3(in_ch, out_ch, r=2):
    return SubpelConv(in_ch, out_ch, 3, r)


class ResBlock(nn.Module):
    """x + last(conv2(act(conv1(first(x))))) (reference: layers.py:59-81): two launches."""

    def __init__(self, channel, slope=0.01, start_from_relu=True, end_with_relu=False, bottleneck=False):
        super().__init__()
        self.slope = 0.0 if slope < 0.0001 else slope
        mid = channel // 2 if bottleneck else channel
        self.conv1 = Conv2d(channel, mid, 3)
        self.conv2 = Conv2d(mid, channel, 3)
        self.start_from_relu, self.end_with_relu = start_from_relu, end_with_relu

    def forward(self, x: Tensor, out=None, plus: Optional[Tensor] = None) -> Tensor:
        """`plus`: an extra tensor added to the block's result, (x + f(x)) + plus, in the same launch."""
        if torch.is_grad_enabled() and out is None and self.conv1.kernel_size == 3 and self.conv1.stride == 1:
            # training: the block as one node of the tape (the sum of x's two gradients is formed inside a kernel)
            return A.res_block(x, plus, self.conv1, self.conv2, self.slope if self.start_from_relu else None,
                               self.slope, self.slope if self.end_with_relu else None)
        t = self.conv1(x, in_slope=self.slope if self.start_from_relu else None, out_slope=self.slope)
        return self.conv2(t, out_slope=self.slope if self.end_with_relu else None, residual=x, residual2=plus,
                          out=out)


class ResidualBlock(nn.Module):
    """act(conv2(act(conv1 x))) + x (reference: layers.py:146-169)."""

    def __init__(self, in_ch, out_ch, leaky_relu_slope=0.01):
        super().__init__()
        self.conv1 = conv3x3(in_ch, out_ch)
        self.conv2 = conv3x3(out_ch, out_ch)
        self.slope = leaky_relu_slope

    def forward(self, x: Tensor) -> Tensor:
        t = self.conv1(x, out_slope=self.slope)
        return self.conv2(t, out_slope=self.slope, residual=x)


class ResidualBlockWithStride(nn.Module):
    """reference: layers.py:84-115 (stride-2 3x3, slopes .01/.1, 1x1 stride-2 shortcut)."""

    def __init__(self, in_ch, out_ch, stride=2):
        super().__init__()
        self.conv1 = conv3x3(in_ch, out_ch, stride=stride)
        self.conv2 = conv3x3(out_ch, out_ch)
        self.downsample = conv1x1(in_ch, out_ch, stride=stride) if stride != 1 else None

    def forward(self, x: Tensor) -> Tensor:
        t = self.conv1(x, out_slope=0.01)
        identity = self.downsample(x) if self.downsample is not None else x
        return self.conv2(t, out_slope=0.1, residual=identity)


class ResidualBlockUpsample(nn.Module):
    """reference: layers.py:118-143 (sub-pixel 1x1 main branch and shortcut)."""

    def __init__(self, in_ch, out_ch, upsample=2):
        super().__init__()
        self.subpel_conv = subpel_conv1x1(in_ch, out_ch, upsample)
        self.conv = conv3x3(out_ch, out_ch)
        self.upsample = subpel_conv1x1(in_ch, out_ch, upsample)

    def forward(self, x: Tensor) -> Tensor:
        t = self.subpel_conv(x, out_slope=0.01)
        identity = self.upsample(x)
        return self.conv(t, out_slope=0.1, residual=identity)


class SELayer(nn.Module):
    """Squeeze-excite gate (reference: layers.py:267-280); returns the per-(n,c) gate, the
    multiply is folded into the consumer's residual add."""

    def __init__(self, channel, reduction=16):
        super().__init__()
        lin1 = nn.Linear(channel, channel // reduction, bias=False)
        lin2 = nn.Linear(channel // reduction, channel, bias=False)
        self.fc = Indexed(_0=lin1, _2=lin2)  # parameter names fc.0.weight / fc.2.weight

    def forward(self, x: Tensor) -> Tensor:
        if torch.is_grad_enabled():
            return A.se_gate(x, self.fc[0].weight, self.fc[2].weight)
        return ops.se_gate(x, self.fc[0].weight.detach(), self.fc[2].weight.detach())

    def from_sums(self, sums: Tensor, h: int, w: int) -> Tensor:
        """The gate from the plane sums the producing convolution emitted with its stores (inference)."""
        return ops.se_gate_from_sums(sums, h, w, self.fc[0].weight.detach(), self.fc[2].weight.detach())


class ConvBlockResidual(nn.Module):
    """up_dim(x) + SE(conv(lrelu(conv(x)))) (reference: layers.py:283-297): three conv launches
    plus the gate; the SE scaling and the sum ride on the 1x1 up_dim launch."""

    def __init__(self, ch_in, ch_out, se_layer=True):
        super().__init__()
        assert se_layer
        self.conv = Indexed(_0=Conv2d(ch_in, ch_out, 3), _2=Conv2d(ch_out, ch_out, 3), _3=SELayer(ch_out))
        self.up_dim = Conv2d(ch_in, ch_out, 1)

    def forward(self, x: Srcs) -> Tensor:
        t = self.conv[0](x, out_slope=0.01)
        if torch.is_grad_enabled():
            t = self.conv[2](t)
            gate = self.conv[3](t)
        else:
            # the plane means the gate needs come out of the convolution's epilogue where the launch allows
            t, sums = self.conv[2](t, plane_sums=True)
            gate = self.conv[3](t) if sums is None else self.conv[3].from_sums(sums, t.size(2), t.size(3))
        return self.up_dim(x, residual=t, res_scale=gate)


class UNet(nn.Module):
    """reference: layers.py:300-341."""

    def __init__(self, in_ch=64, out_ch=64):
        super().__init__()
        self.conv1 = ConvBlockResidual(ch_in=in_ch, ch_out=32)
        self.conv2 = ConvBlockResidual(ch_in=32, ch_out=64)
        self.conv3 = ConvBlockResidual(ch_in=64, ch_out=128)
        self.context_refine = Indexed(_0=ResBlock(128, 0), _1=ResBlock(128, 0), _2=ResBlock(128, 0),
                                      _3=ResBlock(128, 0))
        self.up3 = subpel_conv1x1(128, 64, 2)
        self.up_conv3 = ConvBlockResidual(ch_in=128, ch_out=64)
        self.up2 = subpel_conv1x1(64, 32, 2)
        self.up_conv2 = ConvBlockResidual(ch_in=64, ch_out=out_ch)

    def forward(self, x: Tensor) -> Tensor:
        pool = A.max_pool2 if torch.is_grad_enabled() else ops.max_pool2
        x1 = self.conv1(x)
        x2 = self.conv2(pool(x1))
        x3 = self.conv3(pool(x2))
        for i in range(4):
            x3 = self.context_refine[i](x3)
        d3 = self.up_conv3((x2, self.up3(x3)))
        return self.up_conv2((x1, self.up2(d3)))


class MEBasic(nn.Module):
    """Five 7x7 convs, ReLU between (reference: layers.py:217-233); the last launch adds the
    up-sampled flow (the level's residual connection, layers.py:259)."""

    def __init__(self):
        super().__init__()
        self.conv1 = Conv2d(8, 32, 7)
        self.conv2 = Conv2d(32, 64, 7)
        self.conv3 = Conv2d(64, 32, 7)
        self.conv4 = Conv2d(32, 16, 7)
        self.conv5 = Conv2d(16, 2, 7)

    def forward(self, x: Srcs, residual: Optional[Tensor] = None) -> Tensor:
        x = self.conv1(x, out_slope=0.0)
        x = self.conv2(x, out_slope=0.0)
        x = self.conv3(x, out_slope=0.0)
        x = self.conv4(x, out_slope=0.0)
        return self.conv5(x, residual=residual)


class ME_Spynet(nn.Module):
    """4-level coarse-to-fine flow (reference: layers.py:236-264).  Per level one fused kernel
    builds [im1 | warp(im2, 2*up2(flow)) | 2*up2(flow)] and five conv launches follow."""

    def __init__(self):
        super().__init__()
        self.L = 4
        self.moduleBasic = nn.ModuleList([MEBasic() for _ in range(self.L)])

    def forward(self, im1: Tensor, im2: Tensor) -> Tensor:
        if im1.size(2) % 16 or im1.size(3) % 16:
            raise ValueError("ME_Spynet needs H and W to be multiples of 16")
        grad = torch.is_grad_enabled()
        pool = A.avg_pool2 if grad else ops.avg_pool2
        im1_list, im2_list = [im1], [im2]
        for level in range(self.L - 1):
            im1_list.append(pool(im1_list[level]))
            im2_list.append(pool(im2_list[level]))
        flow = None
        for level in range(self.L):
            k = self.L - 1 - level
            if grad:
                # the fused level-input kernel, taken apart into its differentiable pieces
                n, _, h, w = im1_list[k].shape
                flow_up = (A.bilinear_up2(flow, 2.0) if flow is not None else
                           torch.zeros((n, 2, h, w), dtype=torch.float32, device=im1.device))
                warped = A.flow_warp(im2_list[k], flow_up)
                flow = self.moduleBasic[level]((im1_list[k], warped, flow_up), residual=flow_up)
            else:
                inp = ops.spynet_level_input(im1_list[k], im2_list[k], flow)
                flow = self.moduleBasic[level](inp, residual=inp[:, 6:8])
        return flow


def flow_warp(im: Tensor, flow: Tensor) -> Tensor:
    """reference: layers.py:175-198."""
    return A.flow_warp(im, flow) if torch.is_grad_enabled() else ops.flow_warp(im, flow)


def bilineardownsacling(x: Tensor, scale: float = 1.0) -> Tensor:
    """reference: layers.py:201-206 (x0.5 bilinear == 2x2 mean for even sizes); `scale` folds the
    caller's division (video_model.py:499-500)."""
    return A.avg_pool2(x, scale) if torch.is_grad_enabled() else ops.avg_pool2(x, scale)


def bilinearupsacling(x: Tensor, scale: float = 1.0) -> Tensor:
    """reference: layers.py:209-214."""
    return A.bilinear_up2(x, scale) if torch.is_grad_enabled() else ops.bilinear_up2(x, scale)


class MotionEncoder(Indexed):
    """reference: layers.py:345-353."""

    def __init__(self, input_channel, channel):
        super().__init__(
            _0=ResidualBlockWithStride(input_channel, channel, stride=2), _1=ResidualBlock(channel, channel),
            _2=ResidualBlockWithStride(channel, channel, stride=2), _3=ResidualBlock(channel, channel),
            _4=ResidualBlockWithStride(channel, channel, stride=2), _5=ResidualBlock(channel, channel),
            _6=conv3x3(channel, channel, stride=2))

    def forward(self, x: Tensor) -> Tensor:
        for i in range(7):
            x = self[i](x)
        return x


class MotionDecoder(Indexed):
    """reference: layers.py:355-364."""

    def __init__(self, output_channel, channel):
        super().__init__(
            _0=ResidualBlock(channel, channel), _1=ResidualBlockUpsample(channel, channel, 2),
            _2=ResidualBlock(channel, channel), _3=ResidualBlockUpsample(channel, channel, 2),
            _4=ResidualBlock(channel, channel), _5=ResidualBlockUpsample(channel, channel, 2),
            _6=ResidualBlock(channel, channel), _7=subpel_conv1x1(channel, output_channel, 2))

    def forward(self, x: Tensor) -> Tensor:
        for i in range(8):
            x = self[i](x)
        return x


def get_enc_dec_models(input_channel, output_channel, channel):
    return MotionEncoder(input_channel, channel), MotionDecoder(output_channel, channel)


class HyperEncoder(Indexed):
    """reference: layers.py:370-380 (convs at 0,2,4,6,8; LeakyReLU(.01) between)."""

    def __init__(self, y_channel, z_channel):
        super().__init__(_0=conv3x3(y_channel, z_channel), _2=conv3x3(z_channel, z_channel),
                         _4=conv3x3(z_channel, z_channel, stride=2), _6=conv3x3(z_channel, z_channel),
                         _8=conv3x3(z_channel, z_channel, stride=2))

    def forward(self, y: Tensor) -> Tensor:
        x = self[0](y, out_slope=0.01)
        x = self[2](x, out_slope=0.01)
        x = self[4](x, out_slope=0.01)
        x = self[6](x, out_slope=0.01)
        return self[8](x)


class HyperDecoder(Indexed):
    """reference: layers.py:382-392."""

    def __init__(self, y_channel, z_channel):
        super().__init__(_0=conv3x3(z_channel, y_channel), _2=subpel_conv1x1(y_channel, y_channel, 2),
                         _4=conv3x3(y_channel, y_channel * 3 // 2),
                         _6=subpel_conv1x1(y_channel * 3 // 2, y_channel * 3 // 2, 2),
                         _8=conv3x3(y_channel * 3 // 2, y_channel * 2))

    def forward(self, z: Tensor) -> Tensor:
        x = self[0](z, out_slope=0.01)
        x = self[2](x, out_slope=0.01)
        x = self[4](x, out_slope=0.01)
        x = self[6](x, out_slope=0.01)
        return self[8](x)


def get_hyper_enc_dec_models(y_channel, z_channel):
    return HyperEncoder(y_channel, z_channel), HyperDecoder(y_channel, z_channel)


class ConvStack3(Indexed):
    """conv, LeakyReLU(slope), conv, LeakyReLU(slope), conv -- the prior fusion / spatial prior
    stacks (reference: video_model.py:134-148,306-320); convs at 0, 2, 4."""

    def __init__(self, c0, c1, c2, c3, slope=0.2):
        super().__init__(_0=conv3x3(c0, c1), _2=conv3x3(c1, c2), _4=conv3x3(c2, c3))
        self.slope = slope

    def forward(self, x: Srcs) -> Tensor:
        x = self[0](x, out_slope=self.slope)
        x = self[2](x, out_slope=self.slope)
        return self[4](x)
