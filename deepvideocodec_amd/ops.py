"""Tensor-level wrappers over the libdvc_hip.so C ABI.

torch is used for device memory and stream plumbing only: every arithmetic op
here is a HIP kernel of this package launched on torch's current stream.
All tensors must be fp32 (int32 for symbols / indexes) CUDA(HIP) tensors in
NCHW with contiguous rows; batch / channel strides are free, so channel slices
of a larger buffer are accepted as sources, residuals and destinations.
"""
import ctypes as C
from typing import Optional, Sequence, Tuple, Union

import torch

from . import _lib
from ._lib import DvcConv2d, DvcDst, DvcSrc, check

Tensor = torch.Tensor
Srcs = Union[Tensor, Sequence[Tensor]]


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _check4(t: Tensor, name: str, dtype=torch.float32):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a tensor, got {type(t)}")
    if not t.is_cuda:
        raise _lib.DvcError(f"{name}: tensor is on {t.device}; the HIP path needs a GPU tensor (no CPU fallback)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if t.dim() != 4:
        raise ValueError(f"{name}: expected NCHW, got shape {tuple(t.shape)}")
    if t.size(3) > 1 and t.stride(3) != 1 or (t.size(2) > 1 and t.stride(2) != t.size(3)):
        raise ValueError(f"{name}: rows must be contiguous (strides {t.stride()})")


def _check_vec(t: Optional[Tensor], numel: Optional[int], name: str, dtype=torch.float32, device=None):
    """A flat parameter-like operand handed to the library as a raw pointer (bias, gates, packed
    weights, tables): must be a contiguous GPU tensor of the stated dtype and element count, on the
    same device as the data -- anything else would be an out-of-bounds device read."""
    if t is None:
        return
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a tensor, got {type(t)}")
    if not t.is_cuda:
        raise _lib.DvcError(f"{name}: tensor is on {t.device}; the HIP path needs a GPU tensor (no CPU fallback)")
    if device is not None and t.device != device:
        raise ValueError(f"{name}: on {t.device}, the data is on {device}")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: must be contiguous (strides {t.stride()})")
    if numel is not None and t.numel() != numel:
        raise ValueError(f"{name}: expected {numel} elements, got {t.numel()} (shape {tuple(t.shape)})")


def _bs(t: Tensor) -> int:
    return t.stride(0) if t.size(0) > 1 else t.size(1) * max(t.stride(1), t.size(2) * t.size(3))


def _cs(t: Tensor) -> int:
    return t.stride(1) if t.size(1) > 1 else t.size(2) * t.size(3)


def _src(t: Tensor, name="src") -> DvcSrc:
    _check4(t, name)
    return DvcSrc(t.data_ptr(), _bs(t), _cs(t), t.size(1), 0)


def _dst(t: Tensor, name="out") -> DvcDst:
    _check4(t, name)
    return DvcDst(t.data_ptr(), _bs(t), _cs(t))


def _null_dst() -> DvcDst:
    return DvcDst(None, 0, 0)


def _ptr(t: Optional[Tensor]):
    return None if t is None else t.data_ptr()


PRECISIONS = {"fp32": 0, "bf16": 1, "bf16x2": 2, "fp16x2": 3}


def pack_conv_weight(weight: Tensor, stride: int = 1, precision: int = 0) -> Tensor:
    """OIHW weight -> the slab layout dvc_conv2d reads (re-pack whenever the weight changes).
    `precision` (PRECISIONS): 0 exact fp32; 1 bf16-rounded operands; 2 / 3 operands split into bf16 / fp16
    hi + lo parts (7x7: stride 1 only)."""
    if not (weight.is_cuda and weight.is_contiguous() and weight.dtype == torch.float32 and weight.dim() == 4):
        raise ValueError("pack_conv_weight: weight must be a contiguous fp32 OIHW GPU tensor")
    cout, cin, k, k2 = weight.shape
    if k != k2:
        raise ValueError("pack_conv_weight: square kernels only")
    lib = _lib.load()
    precision = int(precision)
    if precision:
        n = check(lib.dvc_conv2d_reduced_packed_weight_floats(cout, cin, k, stride, precision), "dvc_conv2d_reduced_packed_weight_floats")
        packed = torch.empty(n, dtype=torch.float32, device=weight.device)
        check(lib.dvc_conv2d_reduced_pack_weight(weight.data_ptr(), cout, cin, k, stride, precision, packed.data_ptr(), _stream()),
              "dvc_conv2d_reduced_pack_weight")
        return packed
    n = check(lib.dvc_conv2d_packed_weight_floats(cout, cin, k, stride), "dvc_conv2d_packed_weight_floats")
    packed = torch.empty(n, dtype=torch.float32, device=weight.device)
    check(lib.dvc_conv2d_pack_weight(weight.data_ptr(), cout, cin, k, stride, packed.data_ptr(), _stream()),
          "dvc_conv2d_pack_weight")
    return packed


def pack_conv_weight_dgrad(weight: Tensor) -> Tensor:
    """The packed image of the weight of `weight`'s data-gradient convolution (flipped taps, channel axes swapped),
    for conv2d(grad, (cin, cout, k, k), None, packed=...)."""
    if not (weight.is_cuda and weight.is_contiguous() and weight.dtype == torch.float32 and weight.dim() == 4):
        raise ValueError("pack_conv_weight_dgrad: weight must be a contiguous fp32 OIHW GPU tensor")
    cout, cin, k, _ = weight.shape
    lib = _lib.load()
    n = check(lib.dvc_conv2d_packed_weight_floats(cin, cout, k, 1), "dvc_conv2d_packed_weight_floats")
    packed = torch.empty(n, dtype=torch.float32, device=weight.device)
    check(lib.dvc_conv2d_pack_weight_dgrad(weight.data_ptr(), cout, cin, k, packed.data_ptr(), _stream()),
          "dvc_conv2d_pack_weight_dgrad")
    return packed


def conv2d(srcs: Srcs, weight: Tensor, bias: Optional[Tensor], stride: int = 1,
           in_slope: Optional[float] = None, out_slope: Optional[float] = None,
           residual: Optional[Tensor] = None, res_scale: Optional[Tensor] = None,
           residual2: Optional[Tensor] = None, pixel_shuffle: bool = False,
           out: Optional[Tensor] = None, packed: Optional[Tensor] = None, precision: int = 0,
           plane_sums: bool = False):
    """See DvcConv2d in include/dvc_hip.h.  `plane_sums=True` returns (out, sums): where the launch can emit per-tile
    sums of its output planes with its stores (dvc_conv2d_plane_sum_slots), `sums` is the (N, Cout, slots) tensor
    se_gate_from_sums() takes, otherwise None (use se_gate on `out`).  `precision`: see PRECISIONS; fp32 accumulation in every mode (`packed` must come from
    pack_conv_weight(..., precision=) with the same value). `srcs` are concatenated along C; slopes of
    None mean "no activation", 0.0 means ReLU.  `weight` is the OIHW tensor (its shape
    describes the layer); `packed` its pack_conv_weight() image -- packed on the fly if
    not given (callers that reuse a weight should cache it)."""
    if isinstance(srcs, torch.Tensor):
        srcs = (srcs,)
    if not 1 <= len(srcs) <= _lib.DVC_MAX_SRC:
        raise ValueError(f"conv2d: {len(srcs)} sources (1..{_lib.DVC_MAX_SRC} supported)")
    p = DvcConv2d()
    n, _, h, w = srcs[0].shape
    cin = 0
    for i, s in enumerate(srcs):
        if s.shape[0] != n or s.shape[2] != h or s.shape[3] != w:
            raise ValueError(f"conv2d: source {i} shape {tuple(s.shape)} does not match {(n, '*', h, w)}")
        p.src[i] = _src(s, f"conv2d src[{i}]")
        cin += s.size(1)
    if isinstance(weight, torch.Tensor):
        wshape, wdev = tuple(weight.shape), weight.device
        if not (weight.is_cuda and weight.is_contiguous() and weight.dtype == torch.float32):
            raise ValueError("conv2d: weight must be a contiguous fp32 GPU tensor")
    else:   # only the layer's (Cout, Cin, k, k): the caller brings the packed image
        wshape, wdev = tuple(weight), srcs[0].device
        if packed is None:
            raise ValueError("conv2d: a weight shape needs its packed image")
    cout, wcin, k, k2 = wshape
    if wcin != cin or k != k2:
        raise ValueError(f"conv2d: weight {wshape} does not match {cin} input channels")
    pad = k // 2
    ho = (h + 2 * pad - k) // stride + 1
    wo = (w + 2 * pad - k) // stride + 1
    oshape = (n, cout // 4, 2 * ho, 2 * wo) if pixel_shuffle else (n, cout, ho, wo)
    if out is None:
        out = torch.empty(oshape, dtype=torch.float32, device=wdev)
    elif tuple(out.shape) != oshape:
        raise ValueError(f"conv2d: out shape {tuple(out.shape)} != {oshape}")
    p.n_src = len(srcs)
    p.batch, p.in_h, p.in_w = n, h, w
    if packed is None:
        packed = pack_conv_weight(weight, stride, precision=precision)
    _check_vec(packed, None, "conv2d packed weight", device=wdev)
    _check_vec(bias, cout, "conv2d bias", device=wdev)
    p.weight_packed = packed.data_ptr()
    p.precision = int(precision)
    p.bias = _ptr(bias)
    p.out_channels = cout
    p.ksize = k
    p.stride = stride
    p.in_act = 0 if in_slope is None else 1
    p.out_act = 0 if out_slope is None else 1
    p.in_slope = 1.0 if in_slope is None else float(in_slope)
    p.out_slope = 1.0 if out_slope is None else float(out_slope)
    p.pixel_shuffle = 1 if pixel_shuffle else 0
    p.out = _dst(out)
    if residual is not None:
        if tuple(residual.shape) != oshape:
            raise ValueError(f"conv2d: residual shape {tuple(residual.shape)} != {oshape}")
        _check4(residual, "conv2d residual")
        p.residual = residual.data_ptr()
        p.res_batch_stride = _bs(residual)
        p.res_chan_stride = _cs(residual)
    if res_scale is not None:
        if residual is None:
            raise ValueError("conv2d: res_scale needs a residual")
        _check_vec(res_scale, n * oshape[1], "conv2d res_scale", device=wdev)
        p.res_scale = res_scale.data_ptr()
    if residual2 is not None:
        if residual is None or tuple(residual2.shape) != oshape:
            raise ValueError("conv2d: residual2 needs residual and the output shape")
        _check4(residual2, "conv2d residual2")
        p.residual2 = residual2.data_ptr()
        p.res2_batch_stride = _bs(residual2)
        p.res2_chan_stride = _cs(residual2)
    sums = None
    if plane_sums:
        slots = check(_lib.load().dvc_conv2d_plane_sum_slots(C.byref(p)), "dvc_conv2d_plane_sum_slots")
        if slots > 0:
            sums = torch.empty((n, cout, slots), dtype=torch.float32, device=wdev)
            p.plane_sums = sums.data_ptr()
    check(_lib.load().dvc_conv2d(C.byref(p), _stream()), "dvc_conv2d")
    return (out, sums) if plane_sums else out


def flow_warp(src: Tensor, flow: Tensor, out: Optional[Tensor] = None) -> Tensor:
    n, c, h, w = src.shape
    if tuple(flow.shape) != (n, 2, h, w):
        raise ValueError(f"flow_warp: flow shape {tuple(flow.shape)} != {(n, 2, h, w)}")
    if out is None:
        out = torch.empty_like(src, memory_format=torch.contiguous_format)
    elif tuple(out.shape) != (n, c, h, w):
        raise ValueError(f"flow_warp: out shape {tuple(out.shape)} != {(n, c, h, w)}")
    check(_lib.load().dvc_flow_warp(C.byref(_src(src)), C.byref(_src(flow, "flow")), C.byref(_dst(out)),
                                    n, h, w, _stream()), "dvc_flow_warp")
    return out


def avg_pool2(x: Tensor, scale: float = 1.0) -> Tensor:
    n, c, h, w = x.shape
    out = torch.empty((n, c, h // 2, w // 2), dtype=torch.float32, device=x.device)
    check(_lib.load().dvc_avg_pool2(C.byref(_src(x)), C.byref(_dst(out)), n, h, w, float(scale), _stream()),
          "dvc_avg_pool2")
    return out


def max_pool2(x: Tensor) -> Tensor:
    n, c, h, w = x.shape
    out = torch.empty((n, c, h // 2, w // 2), dtype=torch.float32, device=x.device)
    check(_lib.load().dvc_max_pool2(C.byref(_src(x)), C.byref(_dst(out)), n, h, w, _stream()), "dvc_max_pool2")
    return out


def bilinear_up2(x: Tensor, scale: float = 1.0) -> Tensor:
    n, c, h, w = x.shape
    out = torch.empty((n, c, 2 * h, 2 * w), dtype=torch.float32, device=x.device)
    check(_lib.load().dvc_bilinear_up2(C.byref(_src(x)), C.byref(_dst(out)), n, h, w, float(scale), _stream()),
          "dvc_bilinear_up2")
    return out


def spynet_level_input(im1: Tensor, im2: Tensor, flow_prev: Optional[Tensor]) -> Tensor:
    n, _, h, w = im1.shape
    out = torch.empty((n, 8, h, w), dtype=torch.float32, device=im1.device)
    fp = C.byref(_src(flow_prev, "flow_prev")) if flow_prev is not None else None
    check(_lib.load().dvc_spynet_level_input(C.byref(_src(im1, "im1")), C.byref(_src(im2, "im2")), fp,
                                             C.byref(_dst(out)), n, h, w, _stream()), "dvc_spynet_level_input")
    return out


def se_gate(x: Tensor, fc1: Tensor, fc2: Tensor) -> Tensor:
    """sigmoid(fc2 @ relu(fc1 @ mean_hw(x))) -> (N*C,) gate."""
    n, c, h, w = x.shape
    if fc1.dim() != 2 or fc1.size(1) != c:
        raise ValueError(f"se_gate: fc1 shape {tuple(fc1.shape)} does not match {c} channels")
    _check_vec(fc1, fc1.size(0) * c, "se_gate fc1", device=x.device)
    _check_vec(fc2, c * fc1.size(0), "se_gate fc2", device=x.device)
    if tuple(fc2.shape) != (c, fc1.size(0)):
        raise ValueError(f"se_gate: fc2 shape {tuple(fc2.shape)} != {(c, fc1.size(0))}")
    lib = _lib.load()
    partial = torch.empty(lib.dvc_se_scratch_floats(n, c), dtype=torch.float32, device=x.device)
    gate = torch.empty(n * c, dtype=torch.float32, device=x.device)
    check(lib.dvc_se_gate(C.byref(_src(x)), n, h, w, fc1.data_ptr(), fc2.data_ptr(), fc1.size(0),
                          partial.data_ptr(), gate.data_ptr(), _stream()), "dvc_se_gate")
    return gate


def se_gate_from_sums(sums: Tensor, h: int, w: int, fc1: Tensor, fc2: Tensor) -> Tensor:
    """se_gate of an (N, C, h, w) tensor whose per-tile plane sums `sums` (N, C, slots) a convolution wrote with
    its stores (conv2d(..., plane_sums=True)) -> (N*C,) gate."""
    if sums.dim() != 3:
        raise ValueError(f"se_gate_from_sums: sums shape {tuple(sums.shape)} is not (N, C, slots)")
    n, c, slots = sums.shape
    _check_vec(sums, n * c * slots, "se_gate_from_sums sums")
    if fc1.dim() != 2 or fc1.size(1) != c or tuple(fc2.shape) != (c, fc1.size(0)):
        raise ValueError(f"se_gate_from_sums: fc shapes {tuple(fc1.shape)}, {tuple(fc2.shape)} do not match {c} channels")
    _check_vec(fc1, fc1.size(0) * c, "se_gate_from_sums fc1", device=sums.device)
    _check_vec(fc2, c * fc1.size(0), "se_gate_from_sums fc2", device=sums.device)
    lib = _lib.load()
    partial = torch.empty(lib.dvc_se_scratch_floats(n, c), dtype=torch.float32, device=sums.device)
    gate = torch.empty(n * c, dtype=torch.float32, device=sums.device)
    check(lib.dvc_se_gate_from_sums(sums.data_ptr(), slots, n, c, h, w, fc1.data_ptr(), fc2.data_ptr(), fc1.size(0),
                                    partial.data_ptr(), gate.data_ptr(), _stream()), "dvc_se_gate_from_sums")
    return gate


def copy_into(src: Tensor, out: Tensor) -> Tensor:
    n, c, h, w = src.shape
    if tuple(out.shape) != tuple(src.shape):
        raise ValueError("copy_into: shape mismatch")
    check(_lib.load().dvc_copy(C.byref(_src(src)), C.byref(_dst(out)), n, h, w, _stream()), "dvc_copy")
    return out


def clamp(x: Tensor, lo: float, hi: float) -> Tensor:
    n, c, h, w = x.shape
    out = torch.empty((n, c, h, w), dtype=torch.float32, device=x.device)
    check(_lib.load().dvc_clamp(C.byref(_src(x)), C.byref(_dst(out)), n, h, w, float(lo), float(hi), _stream()),
          "dvc_clamp")
    return out


def eb_forward(z: Tensor, packed: Tensor, medians: Tensor, noise: Optional[Tensor] = None,
               want_likelihood=True, want_symbols=False):
    """-> (z_hat, likelihood | None, symbols | None)"""
    n, c, h, w = z.shape
    _check_vec(packed, c * _lib.EB_PARAMS_PER_CHANNEL, "eb_forward packed parameters", device=z.device)
    _check_vec(medians, c, "eb_forward medians", device=z.device)
    _check_vec(noise, n * c * h * w, "eb_forward noise", device=z.device)
    z_hat = torch.empty((n, c, h, w), dtype=torch.float32, device=z.device)
    lik = torch.empty_like(z_hat) if want_likelihood else None
    sym = torch.empty((n, c, h, w), dtype=torch.int32, device=z.device) if want_symbols else None
    if noise is not None and not (noise.is_contiguous() and tuple(noise.shape) == (n, c, h, w)):
        raise ValueError("eb_forward: noise must be contiguous and shaped like z")
    check(_lib.load().dvc_eb_forward(C.byref(_src(z, "z")), _ptr(noise), packed.data_ptr(), medians.data_ptr(),
                                     C.byref(_dst(z_hat)), C.byref(_dst(lik) if lik is not None else _null_dst()),
                                     _ptr(sym), n, h, w, _stream()), "dvc_eb_forward")
    return z_hat, lik, sym


def eb_dequantize(symbols: Tensor, medians: Tensor) -> Tensor:
    n, c, h, w = symbols.shape
    _check_vec(symbols, None, "eb_dequantize symbols", dtype=torch.int32)
    _check_vec(medians, c, "eb_dequantize medians", device=symbols.device)
    z_hat = torch.empty((n, c, h, w), dtype=torch.float32, device=symbols.device)
    check(_lib.load().dvc_eb_dequantize(symbols.data_ptr(), medians.data_ptr(), C.byref(_dst(z_hat)), n, c, h, w,
                                        _stream()), "dvc_eb_dequantize")
    return z_hat


def dual_prior_pass1(y: Tensor, means: Tensor, scales: Tensor) -> Tensor:
    n, c, h, w = y.shape
    out = torch.empty((n, c, h, w), dtype=torch.float32, device=y.device)
    check(_lib.load().dvc_dual_prior_pass1(C.byref(_src(y, "y")), C.byref(_src(means, "means")),
                                           C.byref(_src(scales, "scales")), C.byref(_dst(out)), n, h, w, _stream()),
          "dvc_dual_prior_pass1")
    return out


def process_with_mask(y: Tensor, means: Tensor, scales: Tensor, mask: Tensor):
    """-> (y_quant, y_hat, means_hat, scales_hat) of the reference's process_with_mask (video_model.py:161-167);
    `mask`: (1, 1, h, w) or (h, w) floats."""
    n, c, h, w = y.shape
    _check_vec(mask, h * w, "process_with_mask mask", device=y.device)
    outs = [torch.empty((n, c, h, w), dtype=torch.float32, device=y.device) for _ in range(4)]
    check(_lib.load().dvc_process_with_mask(C.byref(_src(y, "y")), C.byref(_src(means, "means")),
                                            C.byref(_src(scales, "scales")), mask.data_ptr(), *[C.byref(_dst(o)) for o in outs],
                                            n, h, w, _stream()), "dvc_process_with_mask")
    return tuple(outs)


def dual_prior_pass2(y: Tensor, means: Tensor, scales: Tensor, spatial: Tensor, noise: Optional[Tensor] = None,
                     want=("y_hat", "means_hat", "scales_hat", "likelihood"), scale_table: Optional[Tensor] = None):
    """-> dict with the requested tensors; 'symbols' in `want` also yields 'indexes'
    (both (2, N, C/2, h, w) int32: plane 0 = first-pass sites, plane 1 = second-pass sites)."""
    n, c, h, w = y.shape
    res = {}
    dsts = {}
    for k in ("y_hat", "means_hat", "scales_hat", "likelihood"):
        if k in want:
            res[k] = torch.empty((n, c, h, w), dtype=torch.float32, device=y.device)
            dsts[k] = _dst(res[k])
        else:
            dsts[k] = _null_dst()
    sym = idx = None
    if "symbols" in want:
        if scale_table is None:
            raise ValueError("dual_prior_pass2: scale_table required for symbols/indexes")
        _check_vec(scale_table, None, "dual_prior_pass2 scale_table", device=y.device)
        sym = torch.empty((2, n, c // 2, h, w), dtype=torch.int32, device=y.device)
        idx = torch.empty_like(sym)
        res["symbols"], res["indexes"] = sym, idx
    if noise is not None and not (noise.is_contiguous() and tuple(noise.shape) == (n, c, h, w)):
        raise ValueError("dual_prior_pass2: noise must be contiguous and shaped like y")
    check(_lib.load().dvc_dual_prior_pass2(
        C.byref(_src(y, "y")), C.byref(_src(means, "means")), C.byref(_src(scales, "scales")),
        C.byref(_src(spatial, "spatial")), _ptr(noise), C.byref(dsts["y_hat"]), C.byref(dsts["means_hat"]),
        C.byref(dsts["scales_hat"]), C.byref(dsts["likelihood"]), _ptr(sym), _ptr(idx), _ptr(scale_table),
        0 if scale_table is None else scale_table.numel(), n, h, w, _stream()), "dvc_dual_prior_pass2")
    return res


def dual_prior_dec_indexes(s: Tensor, plane: int, channels: int, scale_table: Tensor) -> Tensor:
    n, _, h, w = s.shape
    _check_vec(scale_table, None, "dual_prior_dec_indexes scale_table", device=s.device)
    idx = torch.empty((n, channels // 2, h, w), dtype=torch.int32, device=s.device)
    check(_lib.load().dvc_dual_prior_dec_indexes(C.byref(_src(s)), plane, idx.data_ptr(), scale_table.data_ptr(),
                                                 scale_table.numel(), n, channels, h, w, _stream()),
          "dvc_dual_prior_dec_indexes")
    return idx


def dual_prior_dec_apply(symbols: Tensor, m: Tensor, plane: int, channels: int,
                         yhat1: Optional[Tensor] = None) -> Tensor:
    n, _, h, w = m.shape
    out = torch.empty((n, channels, h, w), dtype=torch.float32, device=m.device)
    if symbols.dtype != torch.int32 or not symbols.is_contiguous() or symbols.numel() != n * (channels // 2) * h * w:
        raise ValueError("dual_prior_dec_apply: symbols must be contiguous int32 (N, C/2, h, w)")
    y1 = C.byref(_src(yhat1, "yhat1")) if yhat1 is not None else None
    check(_lib.load().dvc_dual_prior_dec_apply(symbols.data_ptr(), C.byref(_src(m)), plane, y1, C.byref(_dst(out)),
                                               n, channels, h, w, _stream()), "dvc_dual_prior_dec_apply")
    return out


def likelihood_bits(lik: Tensor) -> Tensor:
    """(N,) tensor: sum over (C,H,W) of -log2(lik)."""
    n, c, h, w = lik.shape
    lib = _lib.load()
    partial = torch.empty(lib.dvc_bits_scratch_floats(n), dtype=torch.float32, device=lik.device)
    bits = torch.empty(n, dtype=torch.float32, device=lik.device)
    check(lib.dvc_likelihood_bits(C.byref(_src(lik)), n, h, w, partial.data_ptr(), bits.data_ptr(), _stream()),
          "dvc_likelihood_bits")
    return bits


def sq_err_sum(a: Tensor, b: Tensor) -> Tensor:
    n, c, h, w = a.shape
    if tuple(b.shape) != tuple(a.shape):
        raise ValueError("sq_err_sum: shape mismatch")
    lib = _lib.load()
    partial = torch.empty(lib.dvc_bits_scratch_floats(n), dtype=torch.float32, device=a.device)
    sse = torch.empty(n, dtype=torch.float32, device=a.device)
    check(lib.dvc_sq_err_sum(C.byref(_src(a)), C.byref(_src(b)), n, h, w, partial.data_ptr(), sse.data_ptr(),
                             _stream()), "dvc_sq_err_sum")
    return sse


# ------------------------------------------------------------------------------------ gradients
def conv2d_wgrad(srcs: Srcs, grad_pre: Tensor, wshape, stride: int = 1, in_slope: Optional[float] = None,
                 with_bias: bool = False):
    """Weight gradient (OIHW) of conv2d; grad_pre = contiguous (N, Cout, ho, wo) gradient before the
    output activation (see act_grad).  `with_bias`: also return the bias gradient (sum of grad_pre over
    batch and pixels, from the same pass) as a second tensor."""
    if isinstance(srcs, torch.Tensor):
        srcs = (srcs,)
    cout, cin, k, _ = wshape
    n, _, h, w = srcs[0].shape
    if not grad_pre.is_contiguous() or grad_pre.dtype != torch.float32 or grad_pre.size(1) != cout:
        raise ValueError("conv2d_wgrad: grad_pre must be a contiguous fp32 (N, Cout, ho, wo) tensor")
    arr = (DvcSrc * len(srcs))(*[_src(s, f"wgrad src[{i}]") for i, s in enumerate(srcs)])
    if sum(s.size(1) for s in srcs) != cin:
        raise ValueError("conv2d_wgrad: sources do not add up to the weight's input channels")
    lib = _lib.load()
    nfl = check(lib.dvc_conv2d_wgrad_scratch_floats(cout, cin, k, stride, n, h, w), "dvc_conv2d_wgrad_scratch_floats")
    scratch = torch.empty(nfl, dtype=torch.float32, device=grad_pre.device)
    gw = torch.empty((cout, cin, k, k), dtype=torch.float32, device=grad_pre.device)
    gb = torch.empty(cout, dtype=torch.float32, device=grad_pre.device) if with_bias else None
    check(lib.dvc_conv2d_wgrad(arr, len(srcs), n, h, w, 0 if in_slope is None else 1,
                               1.0 if in_slope is None else float(in_slope), grad_pre.data_ptr(), cout, k, stride,
                               scratch.data_ptr(), gw.data_ptr(), gb.data_ptr() if with_bias else None, _stream()),
          "dvc_conv2d_wgrad")
    return (gw, gb) if with_bias else gw


def act_grad(grad: Tensor, act_out: Optional[Tensor], slope: float, pixel_shuffle: bool, conv_shape) -> Tensor:
    n, c, ho, wo = conv_shape
    out = torch.empty((n, c, ho, wo), dtype=torch.float32, device=grad.device)
    ao = C.byref(_src(act_out, "act_out")) if act_out is not None else None
    check(_lib.load().dvc_act_grad(C.byref(_src(grad, "grad")), ao, float(slope), 1 if pixel_shuffle else 0,
                                   out.data_ptr(), n, c, ho, wo, _stream()), "dvc_act_grad")
    return out


def mul_act_grad(grad: Tensor, x: Tensor, slope: float, out: Optional[Tensor] = None,
                 plus: Optional[Tensor] = None) -> Tensor:
    """grad * (x > 0 ? 1 : slope) (+ plus)."""
    n, c, h, w = grad.shape
    if out is None:
        out = torch.empty((n, c, h, w), dtype=torch.float32, device=grad.device)
    if plus is not None and tuple(plus.shape) != (n, c, h, w):
        raise ValueError(f"mul_act_grad: plus shape {tuple(plus.shape)} != {(n, c, h, w)}")
    check(_lib.load().dvc_mul_act_grad_add(C.byref(_src(grad, "grad")), C.byref(_src(x, "x")), float(slope),
                                           C.byref(_src(plus, "plus")) if plus is not None else None,
                                           C.byref(_dst(out)), n, h, w, _stream()), "dvc_mul_act_grad")
    return out


def zero_upsample2(x: Tensor, out_h: int, out_w: int) -> Tensor:
    n, c, h, w = x.shape
    out = torch.empty((n, c, out_h, out_w), dtype=torch.float32, device=x.device)
    check(_lib.load().dvc_zero_upsample2(C.byref(_src(x)), C.byref(_dst(out)), n, h, w, out_h, out_w, _stream()),
          "dvc_zero_upsample2")
    return out


def add_scaled(a: Tensor, b: Tensor, scale: Optional[Tensor] = None, c: Optional[Tensor] = None) -> Tensor:
    """a + b * scale[n*C + ch] (+ c)."""
    n, ch, h, w = a.shape
    out = torch.empty((n, ch, h, w), dtype=torch.float32, device=a.device)
    cc = C.byref(_src(c, "c")) if c is not None else None
    check(_lib.load().dvc_add_scaled(C.byref(_src(a, "a")), C.byref(_src(b, "b")), _ptr(scale), cc, C.byref(_dst(out)),
                                     n, h, w, _stream()), "dvc_add_scaled")
    return out


def scale_planes(x: Tensor, scale: Tensor) -> Tensor:
    n, ch, h, w = x.shape
    out = torch.empty((n, ch, h, w), dtype=torch.float32, device=x.device)
    check(_lib.load().dvc_scale_planes(C.byref(_src(x)), scale.data_ptr(), C.byref(_dst(out)), n, h, w, _stream()),
          "dvc_scale_planes")
    return out


def plane_dot(a: Tensor, b: Optional[Tensor] = None, sum_batch: bool = False) -> Tensor:
    """(N*C,) plane sums of a*b (or of a); (C,) summed over the batch too when sum_batch."""
    n, ch, h, w = a.shape
    lib = _lib.load()
    scratch = torch.empty(lib.dvc_plane_dot_scratch_floats(n, ch), dtype=torch.float32, device=a.device)
    out = torch.empty(ch if sum_batch else n * ch, dtype=torch.float32, device=a.device)
    bb = C.byref(_src(b, "b")) if b is not None else None
    check(lib.dvc_plane_dot(C.byref(_src(a, "a")), bb, n, h, w, 1 if sum_batch else 0, scratch.data_ptr(),
                            out.data_ptr(), _stream()), "dvc_plane_dot")
    return out


def flow_warp_backward(src: Tensor, flow: Tensor, grad_out: Tensor, want_src=True, want_flow=True):
    n, ch, h, w = src.shape
    gs = torch.zeros((n, ch, h, w), dtype=torch.float32, device=src.device) if want_src else None
    gf = torch.empty((n, 2, h, w), dtype=torch.float32, device=src.device) if want_flow else None
    check(_lib.load().dvc_flow_warp_backward(
        C.byref(_src(src, "src")), C.byref(_src(flow, "flow")), C.byref(_src(grad_out, "grad_out")),
        C.byref(_dst(gs)) if gs is not None else None, C.byref(_dst(gf)) if gf is not None else None,
        n, h, w, _stream()), "dvc_flow_warp_backward")
    return gs, gf


def avg_pool2_backward(grad_out: Tensor, h: int, w: int, scale: float = 1.0) -> Tensor:
    n, ch = grad_out.shape[:2]
    out = torch.empty((n, ch, h, w), dtype=torch.float32, device=grad_out.device)
    check(_lib.load().dvc_avg_pool2_backward(C.byref(_src(grad_out)), C.byref(_dst(out)), n, h, w, float(scale),
                                             _stream()), "dvc_avg_pool2_backward")
    return out


def max_pool2_backward(x: Tensor, grad_out: Tensor) -> Tensor:
    n, ch, h, w = x.shape
    out = torch.empty((n, ch, h, w), dtype=torch.float32, device=x.device)
    check(_lib.load().dvc_max_pool2_backward(C.byref(_src(x, "x")), C.byref(_src(grad_out, "grad_out")),
                                             C.byref(_dst(out)), n, h, w, _stream()), "dvc_max_pool2_backward")
    return out


def bilinear_up2_backward(grad_out: Tensor, h: int, w: int, scale: float = 1.0) -> Tensor:
    n, ch = grad_out.shape[:2]
    out = torch.empty((n, ch, h, w), dtype=torch.float32, device=grad_out.device)
    check(_lib.load().dvc_bilinear_up2_backward(C.byref(_src(grad_out)), C.byref(_dst(out)), n, h, w, float(scale),
                                                _stream()), "dvc_bilinear_up2_backward")
    return out


def eb_backward(x_eval: Tensor, packed: Tensor, grad_lik: Tensor, want_gx: bool = True):
    """-> (grad wrt x_eval | None, grad wrt the raw packed parameters (C, 58))."""
    n, c, h, w = x_eval.shape
    lib = _lib.load()
    gx = torch.empty((n, c, h, w), dtype=torch.float32, device=x_eval.device) if want_gx else None
    scratch = torch.empty(lib.dvc_eb_backward_scratch_floats(c), dtype=torch.float32, device=x_eval.device)
    gp = torch.empty((c, _lib.EB_PARAMS_PER_CHANNEL), dtype=torch.float32, device=x_eval.device)
    check(lib.dvc_eb_backward(C.byref(_src(x_eval, "x_eval")), packed.data_ptr(), C.byref(_src(grad_lik, "grad_lik")),
                              C.byref(_dst(gx)) if gx is not None else None, scratch.data_ptr(), gp.data_ptr(),
                              n, h, w, _stream()), "dvc_eb_backward")
    return gx, gp


def dual_prior_pass1_backward(grad_yhat1: Tensor) -> Tensor:
    n, c, h, w = grad_yhat1.shape
    gy = torch.empty((n, c, h, w), dtype=torch.float32, device=grad_yhat1.device)
    check(_lib.load().dvc_dual_prior_pass1_backward(C.byref(_src(grad_yhat1)), C.byref(_dst(gy)), n, h, w, _stream()),
          "dvc_dual_prior_pass1_backward")
    return gy


def dual_prior_pass2_backward(y: Tensor, means: Tensor, scales: Tensor, spatial: Tensor, noise: Optional[Tensor],
                              grad_yhat: Optional[Tensor], grad_lik: Optional[Tensor]):
    n, c, h, w = y.shape
    dev = y.device
    gy = torch.empty((n, c, h, w), dtype=torch.float32, device=dev)
    gm = torch.empty_like(gy)
    gs = torch.empty_like(gy)
    gsp = torch.empty((n, 2 * c, h, w), dtype=torch.float32, device=dev)
    gyh = C.byref(_src(grad_yhat, "grad_yhat")) if grad_yhat is not None else None
    gl = C.byref(_src(grad_lik, "grad_lik")) if grad_lik is not None else None
    check(_lib.load().dvc_dual_prior_pass2_backward(
        C.byref(_src(y, "y")), C.byref(_src(means, "means")), C.byref(_src(scales, "scales")),
        C.byref(_src(spatial, "spatial")), _ptr(noise), gyh, gl, C.byref(_dst(gy)), C.byref(_dst(gm)),
        C.byref(_dst(gs)), C.byref(_dst(gsp)), n, h, w, _stream()), "dvc_dual_prior_pass2_backward")
    return gy, gm, gs, gsp


def likelihood_bits_backward(lik: Tensor, grad_bits: Tensor) -> Tensor:
    n, c, h, w = lik.shape
    out = torch.empty((n, c, h, w), dtype=torch.float32, device=lik.device)
    check(_lib.load().dvc_likelihood_bits_backward(C.byref(_src(lik)), grad_bits.data_ptr(), C.byref(_dst(out)),
                                                   n, h, w, _stream()), "dvc_likelihood_bits_backward")
    return out


def sq_err_backward(a: Tensor, b: Tensor, grad_sse: Tensor) -> Tensor:
    n, c, h, w = a.shape
    out = torch.empty((n, c, h, w), dtype=torch.float32, device=a.device)
    check(_lib.load().dvc_sq_err_backward(C.byref(_src(a, "a")), C.byref(_src(b, "b")), grad_sse.data_ptr(),
                                          C.byref(_dst(out)), n, h, w, _stream()), "dvc_sq_err_backward")
    return out
