"""DMC inter-frame codec on MI355X: the host-side mirror of the reference's
dmc/models/video_model.py (`DMC`, `MotionContextModel`, `FrameContextModel`, and the
context / transform nets), same constructor defaults, attribute / parameter names,
method signatures and return structures.  The tensor work is done by the HIP kernels
of libdvc_hip.so through `deepvideocodec_amd.ops`; torch only carries device memory.

Differences from the reference that a caller can observe:
  * a tensor on the CPU is rejected instead of silently taking a CPU path.
  * forward() / forward_inter() are differentiable when torch's grad mode is on (the training
    path: same forward kernels on the autograd tape, gradient kernels behind them); under
    torch.no_grad() they take the fused inference launches.  encode_inter()/decode_inter()
    never record a tape.
  * compress() does not round-trip z through the range coder to obtain z_hat
    (reference: video_model.py:238-239): z_hat = round(z - median) + median is what that
    round trip returns, so it is computed on the GPU while the host codes the bytes.
"""
import math
import os
from concurrent.futures import ThreadPoolExecutor
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn as nn

from .. import autograd as A
from .. import ops
from .base_model import CompressionModel, get_scale_table
from .entropy_models import GaussianConditional, uniform_noise
from .layers import (ConvStack3, Conv2d, Indexed, ME_Spynet, ResBlock, UNet, bilineardownsacling, flow_warp,
                     get_enc_dec_models, get_hyper_enc_dec_models, subpel_conv3x3)
from .utils import resize_registered_buffers

Tensor = torch.Tensor
# host threads that range-code the symbol streams (six per frame), sized to this rank's share of the host cores
from ..launch import coder_pool_workers, host_budget  # noqa: E402
_POOL = ThreadPoolExecutor(max_workers=coder_pool_workers(host_budget()["cores_per_rank"]))
# DVC_DECODE_OVERLAP=0 keeps decode_inter on one stream (for measuring what the overlap buys)
_DECODE_OVERLAP = os.environ.get("DVC_DECODE_OVERLAP", "1") != "0"


class FeatureExtractor(nn.Module):
    """reference: video_model.py:15-34."""

    def __init__(self, channel=64):
        super().__init__()
        self.conv1 = Conv2d(channel, channel, 3)
        self.res_block1 = ResBlock(channel)
        self.conv2 = Conv2d(channel, channel, 3, stride=2)
        self.res_block2 = ResBlock(channel)
        self.conv3 = Conv2d(channel, channel, 3, stride=2)
        self.res_block3 = ResBlock(channel)

    def forward(self, feature):
        layer1 = self.res_block1(self.conv1(feature))
        layer2 = self.res_block2(self.conv2(layer1))
        layer3 = self.res_block3(self.conv3(layer2))
        return layer1, layer2, layer3


class MultiScaleContextFusion(nn.Module):
    """reference: video_model.py:37-65.  The closing `context + context_out` adds ride on the
    last convolution of each branch (second residual operand of the conv launch)."""

    def __init__(self, channel_in=64, channel_out=64):
        super().__init__()
        self.conv3_up = subpel_conv3x3(channel_in, channel_out, 2)
        self.res_block3_up = ResBlock(channel_out)
        self.conv3_out = Conv2d(channel_out, channel_out, 3)
        self.res_block3_out = ResBlock(channel_out)
        self.conv2_up = subpel_conv3x3(channel_out * 2, channel_out, 2)
        self.res_block2_up = ResBlock(channel_out)
        self.conv2_out = Conv2d(channel_out * 2, channel_out, 3)
        self.res_block2_out = ResBlock(channel_out)
        self.conv1_out = Conv2d(channel_out * 2, channel_out, 3)
        self.res_block1_out = ResBlock(channel_out)

    def forward(self, context1, context2, context3, out2=None, out3=None):
        """`out2` / `out3` (inference only): where to write the fused half- and quarter-resolution contexts,
        e.g. the second halves of the concat buffers their consumers read."""
        context3_up = self.res_block3_up(self.conv3_up(context3))
        context3_new = self.res_block3_out(self.conv3_out(context3), plus=context3, out=out3)
        cat2 = (context3_up, context2)
        context2_up = self.res_block2_up(self.conv2_up(cat2))
        context2 = self.res_block2_out(self.conv2_out(cat2), plus=context2, out=out2)
        context1 = self.res_block1_out(self.conv1_out((context2_up, context1)), plus=context1)
        return context1, context2, context3_new


class ContextualEncoder(nn.Module):
    """reference: video_model.py:68-87."""

    def __init__(self, N=64, M=96):
        super().__init__()
        self.N = N
        self.conv1 = Conv2d(N + 3, N, 3, stride=2)
        self.res1 = ResBlock(N * 2, bottleneck=True, slope=0.1, start_from_relu=True, end_with_relu=True)
        self.conv2 = Conv2d(N * 2, N, 3, stride=2)
        self.res2 = ResBlock(N * 2, bottleneck=True, slope=0.1, start_from_relu=True, end_with_relu=True)
        self.conv3 = Conv2d(N * 2, N, 3, stride=2)
        self.conv4 = Conv2d(N, M, 3, stride=2)

    def forward(self, x, context1, context2, context3):
        N = self.N
        n, _, h, w = x.shape
        # the bottleneck ResBlocks need their concatenated input as one tensor (it is also the
        # skip): the producer conv writes its half in place, the context half is copied in.
        if torch.is_grad_enabled():   # training path: the concat is a differentiable copy
            cat1 = torch.cat((self.conv1((x, context1)), context2), 1)
            cat2 = torch.cat((self.conv2(self.res1(cat1)), context3), 1)
            return self.conv4(self.conv3(self.res2(cat2)))
        cat1 = _concat_buffer(context2, N)
        self.conv1((x, context1), out=cat1[:, :N])
        cat2 = _concat_buffer(context3, N)
        self.conv2(self.res1(cat1), out=cat2[:, :N])
        feature = self.conv3(self.res2(cat2))
        return self.conv4(feature)


class ContextualDecoder(nn.Module):
    """reference: video_model.py:90-109."""

    def __init__(self, N=64, M=96):
        super().__init__()
        self.N = N
        self.up1 = subpel_conv3x3(M, N, 2)
        self.up2 = subpel_conv3x3(N, N, 2)
        self.res1 = ResBlock(N * 2, bottleneck=True, slope=0.1, start_from_relu=True, end_with_relu=True)
        self.up3 = subpel_conv3x3(N * 2, N, 2)
        self.res2 = ResBlock(N * 2, bottleneck=True, slope=0.1, start_from_relu=True, end_with_relu=True)
        self.up4 = subpel_conv3x3(N * 2, 32, 2)

    def forward(self, x, context2, context3):
        N = self.N
        n, _, h, w = x.shape
        feature = self.up1(x)
        if torch.is_grad_enabled():
            feature = self.res1(torch.cat((self.up2(feature), context3), 1))
            feature = self.res2(torch.cat((self.up3(feature), context2), 1))
            return self.up4(feature)
        cat1 = _concat_buffer(context3, N)
        self.up2(feature, out=cat1[:, :N])
        feature = self.res1(cat1)
        cat2 = _concat_buffer(context2, N)
        self.up3(feature, out=cat2[:, :N])
        feature = self.res2(cat2)
        return self.up4(feature)


class ReconGeneration(nn.Module):
    """reference: video_model.py:112-125; returns (feature, recon)."""

    def __init__(self, ctx_channel=64, res_channel=32, channel=64):
        super().__init__()
        self.first_conv = Conv2d(ctx_channel + res_channel, channel, 3)
        self.unet_1 = UNet(channel)
        self.unet_2 = UNet(channel)
        self.recon_conv = Conv2d(channel, 3, 3)

    def forward(self, ctx, res):
        feature = self.first_conv((ctx, res))
        feature = self.unet_1(feature)
        feature = self.unet_2(feature)
        return feature, self.recon_conv(feature)


def _new_concat_context(n, channels, h, w, device):
    """A (n, 2*channels, h, w) buffer whose second half is handed out as the place for a context tensor: the
    bottleneck ResBlocks of the contextual encoder / decoder read cat[conv output, context] as ONE tensor (it is
    also their skip operand), so the context is produced where its consumer needs it instead of being copied."""
    buf = torch.empty((n, 2 * channels, h, w), dtype=torch.float32, device=device)
    view = buf[:, channels:]
    view._dvc_concat = [buf]        # one consumer may claim the buffer's first half
    return view


def _concat_buffer(context: Tensor, channels: int) -> Tensor:
    """The concat buffer `context` already sits in (second half) if it has one that nobody has claimed yet;
    otherwise a fresh buffer with `context` copied into its second half."""
    slot = getattr(context, "_dvc_concat", None)
    if slot:
        return slot.pop()
    n, _, h, w = context.shape
    buf = torch.empty((n, 2 * channels, h, w), dtype=torch.float32, device=context.device)
    ops.copy_into(context, buf[:, channels:])
    return buf


def _checkpoint(fn, *args):
    from torch.utils.checkpoint import checkpoint
    return checkpoint(fn, *args, use_reentrant=False)


# ----------------------------------------------------------------------------------------
class _PendingStrings:
    """Device-side symbols of one context model waiting for the host range coder."""

    def __init__(self, model, z_sym: Tensor, y_sym: Tensor, y_idx: Tensor, z_shape):
        self.model = model
        self.z_shape = z_shape
        self._dev = (z_sym, y_sym, y_idx)
        self._host = None

    def start_download(self):
        """Device -> pinned host copies on the current stream, followed by an event: the host can wait for
        exactly these symbols while later kernels of the frame keep the GPU busy."""
        self._host = [_pinned_like(t) for t in self._dev]
        for h, d in zip(self._host, self._dev):
            h.copy_(d, non_blocking=True)
        self._ready = torch.cuda.Event()
        self._ready.record()

    def encode(self) -> Dict:
        """Waits for the download (not for the rest of the stream) and range-codes the three streams."""
        if self._host is None:
            self.start_download()
        self._ready.synchronize()
        z_sym, y_sym, y_idx = (t.numpy() for t in self._host)
        gc, eb = self.model.gaussian_conditional, self.model.entropy_bottleneck
        f0 = _POOL.submit(gc.compress_symbols, y_sym[0], y_idx[0])
        f1 = _POOL.submit(gc.compress_symbols, y_sym[1], y_idx[1])
        fz = _POOL.submit(eb.compress_symbols, z_sym)
        return {"strings": [f0.result(), f1.result(), fz.result()], "shape": self.z_shape}


def _pinned_like(t: Tensor) -> Tensor:
    return torch.empty(t.shape, dtype=t.dtype, device="cpu", pin_memory=True)


def _download(t: Tensor) -> np.ndarray:
    """Device tensor -> numpy through a pinned buffer; waits for this copy only."""
    h = _pinned_like(t)
    h.copy_(t, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    ev.synchronize()
    return h.numpy()


def _upload(a: np.ndarray, device) -> Tensor:
    """numpy -> device through a pinned buffer, asynchronous on the current stream."""
    h = torch.from_numpy(np.ascontiguousarray(a)).pin_memory()
    return h.to(device, non_blocking=True)


class _DualPriorContextModel(CompressionModel):
    """Hyperprior + checkerboard dual prior shared by the motion and frame context models
    (reference: video_model.py:152-291 and its verbatim twin :324-466)."""

    latent_channels: int
    # a dict set by a test receives the latent, its predicted means and the hyper-latent of the next inference
    # forward() (the values the quantisers see); None in normal operation
    trace: Optional[dict] = None

    @staticmethod
    def get_mask(height, width, device):
        """reference: video_model.py:152-159 (kept for API parity; the kernels derive the
        checkerboard from coordinates: mask_0[i,j] = 1 iff (i+j) is even)."""
        ii = torch.arange(height, device=device).view(height, 1)
        jj = torch.arange(width, device=device).view(1, width)
        mask_0 = (((ii + jj) % 2) == 0).float().view(1, 1, height, width)
        return mask_0, 1.0 - mask_0

    @torch.no_grad()
    def process_with_mask(self, y, means, scales, mask):
        """reference: video_model.py:161-167 -> (y_quant, y_hat, means_hat, scales_hat).  The coding path does not
        call this (its two checkerboard passes are fused kernels); kept for callers of the reference's helper."""
        return ops.process_with_mask(y, means, scales, mask.contiguous())

    @torch.no_grad()
    def forward_dual_prior(self, y, means, scales, mode="trainval"):
        """reference: video_model.py:169-216, same return structure: (y_hat, means_hat, scales_hat), or with
        mode='compress' (y_hat, y_quant_w_0, y_quant_w_1, scales_w_0, scales_w_1) where the w_0 / w_1 planes hold
        channel half 0 on even / odd sites and half 1 on odd / even sites.  Inference only (the differentiable
        form is forward())."""
        _, _, h, w = y.shape
        mask_0, mask_1 = self.get_mask(h, w, y.device)
        y_0, y_1 = y.chunk(2, 1)           # (channel-slice views: the kernels take strided planes)
        m_0, m_1 = means.chunk(2, 1)
        s_0, s_1 = scales.chunk(2, 1)
        q00, yh00, mh00, sh00 = ops.process_with_mask(y_0, m_0, s_0, mask_0)
        q11, yh11, mh11, sh11 = ops.process_with_mask(y_1, m_1, s_1, mask_1)
        m_0, s_0, m_1, s_1 = self.y_spatial_prior((yh00, yh11, means, scales)).chunk(4, 1)
        q01, yh01, mh01, sh01 = ops.process_with_mask(y_0, m_0, s_0, mask_1)
        q10, yh10, mh10, sh10 = ops.process_with_mask(y_1, m_1, s_1, mask_0)
        add = ops.add_scaled
        y_hat = torch.cat((add(yh00, yh01), add(yh11, yh10)), 1)
        if mode == "compress":
            return y_hat, add(q00, q11), add(q01, q10), add(sh00, sh11), add(sh01, sh10)
        return y_hat, torch.cat((add(mh00, mh01), add(mh11, mh10)), 1), torch.cat((add(sh00, sh01), add(sh11, sh10)), 1)

    def _prior_params(self, z_hat: Tensor, y_ref: Optional[Tensor], context: Optional[Tensor]):
        raise NotImplementedError

    def _zeros_ref(self, z_hat: Tensor) -> Tensor:
        n, _, hz, wz = z_hat.shape
        return torch.zeros((n, self.latent_channels, hz * 4, wz * 4), dtype=torch.float32, device=z_hat.device)

    def _scale_table(self) -> Tensor:
        t = self.gaussian_conditional.scale_table
        if t.numel() == 0:
            raise ValueError("Uninitialized scale table. Run update() first")
        return t.float().contiguous()

    def _forward(self, y, y_ref, context):
        grad = torch.is_grad_enabled()
        eb = self.entropy_bottleneck
        z = self.hyper_encoder(y)
        z_hat_or_noisy, z_likelihoods = eb(z)
        # the decoder path always sees round-STE(z - median) + median (video_model.py:220-224), whatever the
        # rate was computed on (z + noise in training mode)
        if grad:
            z_hat = A.round_ste(z, eb.packed_parameters(), eb._medians_flat())
        elif self.training:
            z_hat, _, _ = ops.eb_forward(z, eb.packed_parameters(), eb._medians_flat(), want_likelihood=False)
        else:
            z_hat = z_hat_or_noisy
        means, scales = self._prior_params(z_hat, y_ref, context)
        yhat1 = A.dual_prior_pass1(y, means, scales) if grad else ops.dual_prior_pass1(y, means, scales)
        spatial = self.y_spatial_prior((yhat1, means, scales))
        noise = None
        if self.training:
            noise = uniform_noise(y.shape, y.device)
        if grad:
            y_hat, y_likelihoods = A.dual_prior_pass2(y, means, scales, spatial, noise)
            return y_hat, {"y": y_likelihoods, "z": z_likelihoods}
        want = ("y_hat", "likelihood") if self.trace is None else ("y_hat", "likelihood", "means_hat")
        r = ops.dual_prior_pass2(y, means, scales, spatial, noise=noise, want=want)
        if self.trace is not None:   # what the quantiser saw (parity tests compare pre-rounding values)
            self.trace.update(y=y, means_hat=r["means_hat"], z=z)
        return r["y_hat"], {"y": r["likelihood"], "z": z_likelihoods}

    def _compress_device(self, y, y_ref, context):
        z = self.hyper_encoder(y)
        z_hat, z_sym = self.entropy_bottleneck.quantize_symbols(z)
        means, scales = self._prior_params(z_hat, y_ref, context)
        yhat1 = ops.dual_prior_pass1(y, means, scales)
        spatial = self.y_spatial_prior((yhat1, means, scales))
        r = ops.dual_prior_pass2(y, means, scales, spatial, want=("y_hat", "symbols"),
                                 scale_table=self._scale_table())
        return r["y_hat"], _PendingStrings(self, z_sym, r["symbols"], r["indexes"], z.size()[-2:])

    def _compress(self, y, y_ref, context):
        self.gaussian_conditional.coder_tables()  # fail early if update() was not run
        y_hat, pending = self._compress_device(y, y_ref, context)
        return y_hat, pending.encode()

    def _decompress(self, strings, shape, y_ref, context):
        assert isinstance(strings, list) and len(strings) == 3
        gc, eb = self.gaussian_conditional, self.entropy_bottleneck
        dev = eb.quantiles.device
        C = self.latent_channels
        table = self._scale_table()
        z_hat = eb.decompress(strings[2], shape)
        means, scales = self._prior_params(z_hat, y_ref, context)
        idx0 = ops.dual_prior_dec_indexes(scales, 0, C, table)
        q0 = _upload(gc.decompress_symbols(strings[0], _download(idx0)), dev)
        yhat1 = ops.dual_prior_dec_apply(q0, means, 0, C)
        spatial = self.y_spatial_prior((yhat1, means, scales))
        idx1 = ops.dual_prior_dec_indexes(spatial, 1, C, table)
        q1 = _upload(gc.decompress_symbols(strings[1], _download(idx1)), dev)
        return ops.dual_prior_dec_apply(q1, spatial, 1, C, yhat1=yhat1)


class MotionContextModel(_DualPriorContextModel):
    """reference: video_model.py:128-291."""

    def __init__(self, ch_mv: int = 64):
        super().__init__(entropy_bottleneck_channels=ch_mv)
        self.latent_channels = ch_mv
        self.hyper_encoder, self.hyper_decoder = get_hyper_enc_dec_models(ch_mv, ch_mv)
        self.y_prior_fusion = ConvStack3(ch_mv * 3, ch_mv * 3, ch_mv * 3, ch_mv * 2, slope=0.2)
        self.y_spatial_prior = ConvStack3(ch_mv * 3, ch_mv * 3, ch_mv * 3, ch_mv * 2, slope=0.2)
        self.gaussian_conditional = GaussianConditional(None)

    def _prior_params(self, z_hat, y_ref, context=None):
        params = self.hyper_decoder(z_hat)
        if y_ref is None:
            y_ref = self._zeros_ref(z_hat)
        return self.y_prior_fusion((params, y_ref)).chunk(2, 1)

    def forward(self, y, y_ref):
        return self._forward(y, y_ref, None)

    @torch.no_grad()
    def compress(self, y, y_ref):
        return self._compress(y, y_ref, None)

    @torch.no_grad()
    def decompress(self, strings, shape, y_ref):
        return self._decompress(strings, shape, y_ref, None)


class FrameContextModel(_DualPriorContextModel):
    """reference: video_model.py:294-466."""

    def __init__(self, N: int = 64, M: int = 64):
        super().__init__(entropy_bottleneck_channels=N)
        self.latent_channels = M
        self.hyper_encoder, self.hyper_decoder = get_hyper_enc_dec_models(M, N)
        self.temporal_prior_encoder = Indexed(_0=Conv2d(N, M * 3 // 2, 3, stride=2),
                                              _2=Conv2d(M * 3 // 2, M * 2, 3, stride=2))
        self.y_prior_fusion = ConvStack3(M * 5, M * 4, M * 3, M * 2, slope=0.2)
        self.y_spatial_prior = ConvStack3(M * 3, M * 3, M * 3, M * 2, slope=0.2)
        self.gaussian_conditional = GaussianConditional(None)

    def _prior_params(self, z_hat, y_ref, context):
        params = self.hyper_decoder(z_hat)
        if y_ref is None:
            y_ref = self._zeros_ref(z_hat)
        t = self.temporal_prior_encoder[0](context, out_slope=0.1)
        temporal = self.temporal_prior_encoder[2](t)
        return self.y_prior_fusion((temporal, params, y_ref)).chunk(2, 1)

    def forward(self, y, y_ref, context):
        return self._forward(y, y_ref, context)

    @torch.no_grad()
    def compress(self, y, y_ref, context):
        return self._compress(y, y_ref, context)

    @torch.no_grad()
    def decompress(self, strings, shape, y_ref, context):
        return self._decompress(strings, shape, y_ref, context)


# ----------------------------------------------------------------------------------------
class DMC(nn.Module):
    """reference: video_model.py:469-679."""

    def __init__(self, ch_mv=64, N=64, M=96):
        super().__init__()
        self.optic_flow = ME_Spynet()
        self.motion_encoder, self.motion_decoder = get_enc_dec_models(2, 2, ch_mv)
        self.motion_context_model = MotionContextModel(ch_mv)
        self.feature_adaptor_I = Conv2d(3, N, 3)
        self.feature_adaptor_P = Conv2d(N, N, 1)
        self.feature_extractor = FeatureExtractor()
        self.context_fusion_net = MultiScaleContextFusion()
        self.contextual_encoder = ContextualEncoder(N, M)
        self.contextual_decoder = ContextualDecoder(N, M)
        self.frame_context_model = FrameContextModel(N, M)
        self.recon_generation_net = ReconGeneration()
        self._initialize_weights()
        # evaluation loops that decode every frame right after encoding it may set this (see _start_lookahead)
        self.decode_lookahead = False
        # training: drop the activations of the two largest blocks (reconstruction UNets, context fusion: ~55 %
        # of the frame's arithmetic and most of its saved tensors) after the forward pass and recompute them
        # during backward -- trades ~1/3 more forward work for the memory that 7-frame batches of 32 need
        self.recompute = False

    def _initialize_weights(self):
        """Xavier-normal(gain sqrt 2) weights, bias 0.01 on every conv (reference: :508-513)."""
        for m in self.modules():
            if isinstance(m, Conv2d):
                torch.nn.init.xavier_normal_(m.weight, math.sqrt(2))
                torch.nn.init.constant_(m.bias, 0.01)

    # ---- motion compensation ------------------------------------------------------------
    def multi_scale_feature_extractor(self, dpb):
        if dpb["feature_ref"] is None:
            feature = self.feature_adaptor_I(dpb["x_ref"])
        else:
            feature = self.feature_adaptor_P(dpb["feature_ref"])
        return self.feature_extractor(feature)

    def motion_compensation(self, mv, dpb, ref_features=None):
        """reference: video_model.py:497-506.  `ref_features`: the output of multi_scale_feature_extractor(dpb)
        when the caller has already computed it (it does not depend on the motion)."""
        warpframe = flow_warp(dpb["x_ref"], mv)
        mv2 = bilineardownsacling(mv, 0.5)
        mv3 = bilineardownsacling(mv2, 0.5)
        ref_feature1, ref_feature2, ref_feature3 = ref_features or self.multi_scale_feature_extractor(dpb)
        context1 = flow_warp(ref_feature1, mv)
        context2 = flow_warp(ref_feature2, mv2)
        context3 = flow_warp(ref_feature3, mv3)
        if torch.is_grad_enabled():
            if self.recompute:
                context1, context2, context3 = _checkpoint(self.context_fusion_net, context1, context2, context3)
            else:
                context1, context2, context3 = self.context_fusion_net(context1, context2, context3)
        else:
            n, c, h, w = context1.shape
            out2 = _new_concat_context(n, c, h // 2, w // 2, context1.device)
            out3 = _new_concat_context(n, c, h // 4, w // 4, context1.device)
            context1, _, _ = self.context_fusion_net(context1, context2, context3, out2=out2, out3=out3)
            context2, context3 = out2, out3
        return context1, context2, context3, warpframe

    # ---- training / evaluation forward ----------------------------------------------------
    def forward(self, frames, motion_pretrain=False, frame_pretrain=False):
        if not isinstance(frames, List):
            raise RuntimeError(f"Invalid number of frames: {len(frames)}.")
        reconstructions = []
        frames_likelihoods = []
        x_rec = frames[0]  # the I frame is passed through uncoded (reference: :522-534)
        dpb = {"y_mv_ref": None, "y_ref": None, "feature_ref": None, "x_ref": x_rec}
        for i in range(1, len(frames)):
            x_rec, likelihoods, context = self.forward_inter(frames[i], dpb, motion_pretrain, frame_pretrain)
            reconstructions.append(x_rec)
            frames_likelihoods.append(likelihoods)
            if len(frames) >= 3:
                if motion_pretrain:
                    # the reference raises KeyError here (its forward_inter returns an empty
                    # context, :566 vs :545); keep the previous latents and chain on the warp.
                    dpb = {"y_mv_ref": dpb["y_mv_ref"], "y_ref": None, "feature_ref": None, "x_ref": x_rec}
                else:
                    dpb = {"y_mv_ref": context["y_mv_ref"], "y_ref": context["y_ref"],
                           "feature_ref": context["feature_ref"], "x_ref": x_rec}
        return {"x_hat": reconstructions, "likelihoods": frames_likelihoods}

    def forward_inter(self, x_cur, dpb, motion_pretrain=False, frame_pretrain=False):
        x_ref = dpb["x_ref"]
        mv = self.optic_flow(x_cur, x_ref)
        y_mv = self.motion_encoder(mv)
        y_mv_hat, mv_likelihoods = self.motion_context_model(y_mv, dpb["y_mv_ref"])
        mv_hat = self.motion_decoder(y_mv_hat)
        context1, context2, context3, x_warp = self.motion_compensation(mv_hat, dpb)
        if motion_pretrain:
            return x_warp, {"motion": mv_likelihoods}, {}
        y = self.contextual_encoder(x_cur, context1, context2, context3)
        y_hat, frame_likelihoods = self.frame_context_model(y, dpb["y_ref"], context3)
        x_rec_feature = self.contextual_decoder(y_hat, context2, context3)
        if self.recompute and torch.is_grad_enabled():
            feature, x_rec = _checkpoint(self.recon_generation_net, x_rec_feature, context1)
        else:
            feature, x_rec = self.recon_generation_net(x_rec_feature, context1)
        return (x_rec, {"motion": mv_likelihoods, "frame": frame_likelihoods},
                {"y_mv_ref": y_mv_hat, "y_ref": y_hat, "feature_ref": feature, "x_ref": x_rec})

    # ---- actual coding ------------------------------------------------------------------------
    @torch.no_grad()
    def encode_inter(self, x_cur, dpb):
        """reference: video_model.py:581-599.  All GPU work of the frame is enqueued before the
        first host wait; the six symbol streams are then range-coded on host threads."""
        mcm, fcm = self.motion_context_model, self.frame_context_model
        mcm.gaussian_conditional.coder_tables(), fcm.gaussian_conditional.coder_tables()
        mcm.entropy_bottleneck.coder_tables(), fcm.entropy_bottleneck.coder_tables()
        self._take_lookahead({"x_ref": None, "feature_ref": None})   # drop what an undecoded frame left behind
        mv = self.optic_flow(x_cur, dpb["x_ref"])
        y_mv = self.motion_encoder(mv)
        y_mv_hat, mv_pending = mcm._compress_device(y_mv, dpb["y_mv_ref"], None)
        mv_pending.start_download()
        mv_hat = self.motion_decoder(y_mv_hat)
        context1, context2, context3, _ = self.motion_compensation(mv_hat, dpb)
        y = self.contextual_encoder(x_cur, context1, context2, context3)
        _, frame_pending = fcm._compress_device(y, dpb["y_ref"], context3)
        frame_pending.start_download()
        if self.decode_lookahead:
            self._start_lookahead(dpb)
        mv_strings = mv_pending.encode()       # waits for the motion symbols only: the GPU is still working
        frame_strings = frame_pending.encode()
        return {
            "strings": {"motion": mv_strings["strings"], "frame": frame_strings["strings"]},
            "shape": {"motion": mv_strings["shape"], "frame": frame_strings["shape"]},
        }

    @torch.no_grad()
    def encode_inter_with_recon(self, x_cur, dpb):
        """encode_inter plus the decoder-side reconstruction, computed once: the reference's loop
        (dmc/test.py:187-188) decodes right after encoding and so recomputes the motion decoder,
        the motion compensation and the contexts from identical tensors.  Here the encoder's
        quantised latents (bit-identical to what the range decoder returns) go straight through the
        contextual decoder and the reconstruction net.  Returns (encode_inter's dict, x_rec,
        decode_inter's info dict); x_rec / info equal decode_inter(strings, shape, dpb) exactly."""
        mcm, fcm = self.motion_context_model, self.frame_context_model
        mcm.gaussian_conditional.coder_tables(), fcm.gaussian_conditional.coder_tables()
        mcm.entropy_bottleneck.coder_tables(), fcm.entropy_bottleneck.coder_tables()
        mv = self.optic_flow(x_cur, dpb["x_ref"])
        y_mv = self.motion_encoder(mv)
        y_mv_hat, mv_pending = mcm._compress_device(y_mv, dpb["y_mv_ref"], None)
        mv_pending.start_download()
        mv_hat = self.motion_decoder(y_mv_hat)
        context1, context2, context3, _ = self.motion_compensation(mv_hat, dpb)
        y = self.contextual_encoder(x_cur, context1, context2, context3)
        y_hat, frame_pending = fcm._compress_device(y, dpb["y_ref"], context3)
        frame_pending.start_download()
        x_rec_feature = self.contextual_decoder(y_hat, context2, context3)
        feature, x_rec = self.recon_generation_net(x_rec_feature, context1)
        mv_strings = mv_pending.encode()
        frame_strings = frame_pending.encode()
        enc = {
            "strings": {"motion": mv_strings["strings"], "frame": frame_strings["strings"]},
            "shape": {"motion": mv_strings["shape"], "frame": frame_strings["shape"]},
        }
        return enc, x_rec, {"x_ref": x_rec, "feature_ref": feature, "y_ref": y_hat, "y_mv_ref": y_mv_hat}

    @torch.no_grad()
    def decode_inter(self, strings, shapes, dpb):
        """reference: video_model.py:601-614.  The reference features (feature adaptor + extractor, ~5 % of the
        frame's arithmetic) do not depend on the motion, so they run on a side stream while the host range-decodes
        the motion symbols -- the decoder's two host round trips per context model otherwise leave the GPU idle."""
        main = torch.cuda.current_stream()
        side = self._side_stream(main.device)
        ref_features = self._take_lookahead(dpb)
        if ref_features is None and _DECODE_OVERLAP:
            side.wait_stream(main)                  # the buffer's tensors were produced on the caller's stream
            with torch.cuda.stream(side):
                ref_features = self.multi_scale_feature_extractor(dpb)
        y_mv_hat = self.motion_context_model.decompress(strings["motion"], shapes["motion"], dpb["y_mv_ref"])
        mv_hat = self.motion_decoder(y_mv_hat)
        if ref_features is not None:
            main.wait_stream(side)
            for f in ref_features:
                f.record_stream(main)
        context1, context2, context3, x_warp = self.motion_compensation(mv_hat, dpb, ref_features)
        y_hat = self.frame_context_model.decompress(strings["frame"], shapes["frame"], dpb["y_ref"], context3)
        x_rec_feature = self.contextual_decoder(y_hat, context2, context3)
        feature, x_rec = self.recon_generation_net(x_rec_feature, context1)
        return x_rec, {"x_ref": x_rec, "feature_ref": feature, "y_ref": y_hat, "y_mv_ref": y_mv_hat}

    def _start_lookahead(self, dpb):
        """With `decode_lookahead` set (the caller decodes every frame right after encoding it, as the
        evaluation loop dmc/test.py:186-188 does): queue the decoder's reference-feature extraction for this
        buffer behind the encoder's kernels, on the side stream, so that it runs while the host range-codes
        the frame's symbols instead of after decode_inter has been entered.  The decoder's work is unchanged
        (it does not reuse the encoder's tensors), only started earlier; decode_inter picks the result up when
        it is called with the same buffer tensors and recomputes otherwise."""
        import threading
        main = torch.cuda.current_stream()
        side = self._side_stream(main.device)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            feats = self.multi_scale_feature_extractor(dpb)
        self.__dict__.setdefault("_lookahead", {})[threading.get_ident()] = (dpb["x_ref"], dpb["feature_ref"], feats)

    def _take_lookahead(self, dpb):
        import threading
        # (popped whether it matches or not: a stale entry must neither pin GPU memory nor meet a later frame)
        hit = self.__dict__.get("_lookahead", {}).pop(threading.get_ident(), None)
        if hit is not None and hit[0] is dpb["x_ref"] and hit[1] is dpb["feature_ref"]:
            return hit[2]
        return None

    def _side_stream(self, device):
        """One helper stream per (host thread, device): GOPs coded from several threads do not share it."""
        import threading
        key = (threading.get_ident(), device.index)
        streams = self.__dict__.setdefault("_side_streams", {})
        if key not in streams:
            streams[key] = torch.cuda.Stream(device=device)
        return streams[key]

    # ---- housekeeping ---------------------------------------------------------------------------
    def prepare(self):
        """Build every kernel-side weight image now (they are otherwise built on first use) and wait
        for them; call once before driving the model from several host threads / streams."""
        for m in self.modules():
            if isinstance(m, Conv2d):
                m.packed_weight(m._precision())
        if next(self.parameters()).is_cuda:
            torch.cuda.synchronize()
        return self

    def set_precision(self, precision: str = "fp32", train: bool = False):
        """"fp32" (default): exact fp32 convolutions, the parity path.  The other modes change how the
        convolutions form their products (fp32 accumulation and fp32 tensors throughout; layers with at most four
        output channels stay exact fp32; encoder and decoder must use the same setting):
        "bf16": operands rounded to bf16 (3-4x faster convolutions, visibly different results);
        "bf16x2": operands split into bf16 hi + lo parts (16 significant bits), three 16-bit matrix
        products per term; "fp16x2": the same with fp16 parts (22 significant bits; operands must stay
        below ~1.3e5 in magnitude).  By default this concerns inference only and the convolutions under autograd stay
        fp32; `train=True` sets the mode for those instead (forward and data-gradient convolutions; the weight
        gradients are always formed in fp32) and leaves the inference setting alone."""
        if precision not in ops.PRECISIONS:
            raise ValueError(f"unknown precision {precision!r}")
        for m in self.modules():
            if isinstance(m, Conv2d):
                if train:
                    m.train_precision = ops.PRECISIONS[precision]
                else:
                    m.precision = ops.PRECISIONS[precision]
        return self

    def aux_loss(self):
        """List with one auxiliary (quantile) loss per context model (reference: :616-624)."""
        return [m.aux_loss() for m in self.modules() if isinstance(m, CompressionModel)]

    def load_state_dict(self, state_dict, strict=True):
        for name in ("motion_context_model", "frame_context_model"):
            m = getattr(self, name)
            resize_registered_buffers(m.gaussian_conditional, f"{name}.gaussian_conditional",
                                      ["_quantized_cdf", "_offset", "_cdf_length", "scale_table"], state_dict)
            resize_registered_buffers(m.entropy_bottleneck, f"{name}.entropy_bottleneck",
                                      ["_quantized_cdf", "_offset", "_cdf_length"], state_dict)
        return super().load_state_dict(state_dict, strict=strict)

    @classmethod
    def from_state_dict(cls, state_dict):
        net = cls()
        net.load_state_dict(state_dict)
        return net

    def update(self, scale_table=None, force=False):
        """Build the CDF tables of the four entropy modules (reference: :665-679)."""
        if scale_table is None:
            scale_table = get_scale_table()
        updated = False
        for m in (self.motion_context_model, self.frame_context_model):
            updated |= m.gaussian_conditional.update_scale_table(scale_table, force=force)
            updated |= m.entropy_bottleneck.update(force=force)
        return updated
