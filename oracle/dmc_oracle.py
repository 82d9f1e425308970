"""Functional fp32 PyTorch/CPU restatement of the DMC inter-frame path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Every function takes the
flat state dict `sd` (reference key names) and cites the reference lines it
follows.  Citations are relative to the reference checkout, `dmc/models/`.

The graph wiring here is pinned against the reference's own files by
tests/golden/*.npz (see tests/golden/make_golden.py); the entropy-model
arithmetic is `oracle.entropy_oracle` ("parity unpinned", see there).
"""
import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

from .entropy_oracle import (EntropyBottleneck, GaussianConditional, LIKELIHOOD_BOUND,
                             SCALE_BOUND, lower_bound, standardized_cumulative)

SD = Dict[str, torch.Tensor]


# ------------------------------------------------------------------ primitives
def act(x, slope):
    """LeakyReLU(slope); slope == 0 is ReLU, slope == 1 (None) is identity."""
    if slope is None or slope == 1.0:
        return x
    if slope == 0.0:
        return F.relu(x)
    return F.leaky_relu(x, slope)


def conv(sd: SD, name: str, x, stride=1):
    w = sd[name + ".weight"]
    return F.conv2d(x, w, sd[name + ".bias"], stride=stride, padding=w.shape[-1] // 2)


def subpel(sd: SD, name: str, x, r=2):
    """conv -> PixelShuffle(r): layers.py:45-56 (the conv is child `.0`)."""
    return F.pixel_shuffle(conv(sd, name + ".0", x), r)


def quantize_ste(x):
    """utils.py:149-152 (round half to even; identity gradient)."""
    return (torch.round(x) - x).detach() + x


def flow_warp(feature, flow):
    """layers.py:175-198: backward warp through a normalised grid,
    bilinear, border padding, align_corners=True.  flow[:,0] is horizontal."""
    n, _, h, w = flow.shape
    hor = torch.linspace(-1.0, 1.0, w, dtype=feature.dtype).view(1, 1, 1, w).expand(n, -1, h, -1)
    ver = torch.linspace(-1.0, 1.0, h, dtype=feature.dtype).view(1, 1, h, 1).expand(n, -1, -1, w)
    base = torch.cat([hor, ver], 1)
    fl = torch.cat([flow[:, 0:1] / ((feature.size(3) - 1.0) / 2.0),
                    flow[:, 1:2] / ((feature.size(2) - 1.0) / 2.0)], 1)
    grid = (base + fl).permute(0, 2, 3, 1)
    return F.grid_sample(feature, grid, mode="bilinear", padding_mode="border", align_corners=True)


def bilinear_down2(x):
    """layers.py:201-206."""
    return F.interpolate(x, (x.size(2) // 2, x.size(3) // 2), mode="bilinear", align_corners=False)


def bilinear_up2(x):
    """layers.py:209-214."""
    return F.interpolate(x, (x.size(2) * 2, x.size(3) * 2), mode="bilinear", align_corners=False)


# ------------------------------------------------------------------ blocks
def res_block(sd, p, x, slope=0.01, start_from_relu=True, end_with_relu=False):
    """layers.py:59-81 (`ResBlock`): x + last(conv2(act(conv1(first(x)))))."""
    out = act(x, slope) if start_from_relu else x
    out = act(conv(sd, p + ".conv1", out), slope)
    out = conv(sd, p + ".conv2", out)
    if end_with_relu:
        out = act(out, slope)
    return x + out


def residual_block(sd, p, x):
    """layers.py:146-169: act(conv2(act(conv1 x))) + x, slope .01."""
    out = act(conv(sd, p + ".conv1", x), 0.01)
    out = act(conv(sd, p + ".conv2", out), 0.01)
    return out + x


def residual_block_with_stride(sd, p, x):
    """layers.py:84-115: stride-2 conv, slopes .01 then .1, 1x1 s2 shortcut."""
    out = act(conv(sd, p + ".conv1", x, stride=2), 0.01)
    out = act(conv(sd, p + ".conv2", out), 0.1)
    return out + conv(sd, p + ".downsample", x, stride=2)


def residual_block_upsample(sd, p, x):
    """layers.py:118-143: subpel 1x1 main + subpel 1x1 shortcut."""
    out = act(subpel(sd, p + ".subpel_conv", x), 0.01)
    out = act(conv(sd, p + ".conv", out), 0.1)
    return out + subpel(sd, p + ".upsample", x)


def se_layer(sd, p, x):
    """layers.py:267-280: global mean -> FC(C->C/16) ReLU FC Sigmoid -> scale."""
    y = torch.mean(x, dim=(-1, -2))
    y = F.relu(F.linear(y, sd[p + ".fc.0.weight"]))
    y = torch.sigmoid(F.linear(y, sd[p + ".fc.2.weight"]))
    return x * y[:, :, None, None]


def conv_block_residual(sd, p, x):
    """layers.py:283-297."""
    x1 = conv(sd, p + ".conv.2", act(conv(sd, p + ".conv.0", x), 0.01))
    x1 = se_layer(sd, p + ".conv.3", x1)
    return conv(sd, p + ".up_dim", x) + x1


def unet(sd, p, x):
    """layers.py:300-341."""
    x1 = conv_block_residual(sd, p + ".conv1", x)
    x2 = conv_block_residual(sd, p + ".conv2", F.max_pool2d(x1, 2, 2))
    x3 = conv_block_residual(sd, p + ".conv3", F.max_pool2d(x2, 2, 2))
    for i in range(4):
        x3 = res_block(sd, f"{p}.context_refine.{i}", x3, slope=0.0)
    d3 = torch.cat((x2, subpel(sd, p + ".up3", x3)), dim=1)
    d3 = conv_block_residual(sd, p + ".up_conv3", d3)
    d2 = torch.cat((x1, subpel(sd, p + ".up2", d3)), dim=1)
    return conv_block_residual(sd, p + ".up_conv2", d2)


def me_basic(sd, p, x):
    """layers.py:217-233: five 7x7 convs, ReLU between."""
    for i in range(1, 5):
        x = F.relu(conv(sd, f"{p}.conv{i}", x))
    return conv(sd, p + ".conv5", x)


def me_spynet(sd, im1, im2, p="optic_flow"):
    """layers.py:236-264: 4-level coarse-to-fine flow (im1 current, im2 reference)."""
    L = 4
    im1_list, im2_list = [im1], [im2]
    for level in range(L - 1):
        im1_list.append(F.avg_pool2d(im1_list[level], kernel_size=2, stride=2))
        im2_list.append(F.avg_pool2d(im2_list[level], kernel_size=2, stride=2))
    fine = im2_list[L - 1].shape
    flow = torch.zeros(im1.size(0), 2, fine[2] // 2, fine[3] // 2, dtype=im1.dtype)
    for level in range(L):
        flow_up = bilinear_up2(flow) * 2.0
        k = L - 1 - level
        inp = torch.cat([im1_list[k], flow_warp(im2_list[k], flow_up), flow_up], 1)
        flow = flow_up + me_basic(sd, f"{p}.moduleBasic.{level}", inp)
    return flow


def motion_encoder(sd, mv, p="motion_encoder"):
    """layers.py:345-353."""
    x = mv
    for i in range(3):
        x = residual_block_with_stride(sd, f"{p}.{2 * i}", x)
        x = residual_block(sd, f"{p}.{2 * i + 1}", x)
    return conv(sd, f"{p}.6", x, stride=2)


def motion_decoder(sd, y, p="motion_decoder"):
    """layers.py:355-364."""
    x = y
    for i in range(3):
        x = residual_block(sd, f"{p}.{2 * i}", x)
        x = residual_block_upsample(sd, f"{p}.{2 * i + 1}", x)
    x = residual_block(sd, f"{p}.6", x)
    return subpel(sd, f"{p}.7", x)


def hyper_encoder(sd, p, y):
    """layers.py:370-380."""
    x = act(conv(sd, p + ".0", y), 0.01)
    x = act(conv(sd, p + ".2", x), 0.01)
    x = act(conv(sd, p + ".4", x, stride=2), 0.01)
    x = act(conv(sd, p + ".6", x), 0.01)
    return conv(sd, p + ".8", x, stride=2)


def hyper_decoder(sd, p, z):
    """layers.py:382-392."""
    x = act(conv(sd, p + ".0", z), 0.01)
    x = act(subpel(sd, p + ".2", x), 0.01)
    x = act(conv(sd, p + ".4", x), 0.01)
    x = act(subpel(sd, p + ".6", x), 0.01)
    return conv(sd, p + ".8", x)


def prior_stack(sd, p, x):
    """Three 3x3 convs with LeakyReLU(.2) between: video_model.py:134-148,306-320."""
    x = act(conv(sd, p + ".0", x), 0.2)
    x = act(conv(sd, p + ".2", x), 0.2)
    return conv(sd, p + ".4", x)


def feature_extractor(sd, f, p="feature_extractor"):
    """video_model.py:15-34."""
    l1 = res_block(sd, p + ".res_block1", conv(sd, p + ".conv1", f))
    l2 = res_block(sd, p + ".res_block2", conv(sd, p + ".conv2", l1, stride=2))
    l3 = res_block(sd, p + ".res_block3", conv(sd, p + ".conv3", l2, stride=2))
    return l1, l2, l3


def context_fusion(sd, c1, c2, c3, p="context_fusion_net"):
    """video_model.py:37-65."""
    c3_up = res_block(sd, p + ".res_block3_up", subpel(sd, p + ".conv3_up", c3))
    c3_out = res_block(sd, p + ".res_block3_out", conv(sd, p + ".conv3_out", c3))
    cat2 = torch.cat((c3_up, c2), dim=1)
    c2_up = res_block(sd, p + ".res_block2_up", subpel(sd, p + ".conv2_up", cat2))
    c2_out = res_block(sd, p + ".res_block2_out", conv(sd, p + ".conv2_out", cat2))
    c1_out = res_block(sd, p + ".res_block1_out", conv(sd, p + ".conv1_out", torch.cat((c2_up, c1), dim=1)))
    return c1 + c1_out, c2 + c2_out, c3 + c3_out


def motion_compensation(sd, mv, dpb):
    """video_model.py:490-506."""
    warpframe = flow_warp(dpb["x_ref"], mv)
    mv2 = bilinear_down2(mv) / 2
    mv3 = bilinear_down2(mv2) / 2
    if dpb.get("feature_ref") is None:
        feature = conv(sd, "feature_adaptor_I", dpb["x_ref"])
    else:
        feature = conv(sd, "feature_adaptor_P", dpb["feature_ref"])
    f1, f2, f3 = feature_extractor(sd, feature)
    c1 = flow_warp(f1, mv)
    c2 = flow_warp(f2, mv2)
    c3 = flow_warp(f3, mv3)
    c1, c2, c3 = context_fusion(sd, c1, c2, c3)
    return c1, c2, c3, warpframe


def bottleneck_res_block(sd, p, x):
    """ResBlock(2N, bottleneck, slope .1, relu in and out): video_model.py:72-76,95-99."""
    return res_block(sd, p, x, slope=0.1, start_from_relu=True, end_with_relu=True)


def contextual_encoder(sd, x, c1, c2, c3, p="contextual_encoder"):
    """video_model.py:68-87."""
    f = conv(sd, p + ".conv1", torch.cat([x, c1], dim=1), stride=2)
    f = bottleneck_res_block(sd, p + ".res1", torch.cat([f, c2], dim=1))
    f = conv(sd, p + ".conv2", f, stride=2)
    f = bottleneck_res_block(sd, p + ".res2", torch.cat([f, c3], dim=1))
    f = conv(sd, p + ".conv3", f, stride=2)
    return conv(sd, p + ".conv4", f, stride=2)


def contextual_decoder(sd, y_hat, c2, c3, p="contextual_decoder"):
    """video_model.py:90-109."""
    f = subpel(sd, p + ".up1", y_hat)
    f = subpel(sd, p + ".up2", f)
    f = bottleneck_res_block(sd, p + ".res1", torch.cat([f, c3], dim=1))
    f = subpel(sd, p + ".up3", f)
    f = bottleneck_res_block(sd, p + ".res2", torch.cat([f, c2], dim=1))
    return subpel(sd, p + ".up4", f)


def recon_generation(sd, ctx, res, p="recon_generation_net"):
    """video_model.py:112-125; returns (feature, recon)."""
    f = conv(sd, p + ".first_conv", torch.cat((ctx, res), dim=1))
    f = unet(sd, p + ".unet_1", f)
    f = unet(sd, p + ".unet_2", f)
    return f, conv(sd, p + ".recon_conv", f)


# ------------------------------------------------------------------ entropy side
def checkerboard_masks(h, w):
    """video_model.py:152-159: mask_0[i,j] = 1 iff (i+j) even."""
    ii = torch.arange(h).view(h, 1)
    jj = torch.arange(w).view(1, w)
    m0 = (((ii + jj) % 2) == 0).float().view(1, 1, h, w)
    return m0, 1.0 - m0


def process_with_mask(y, means, scales, mask):
    """video_model.py:161-167."""
    means_hat = means * mask
    scales_hat = scales * mask
    y_quant = quantize_ste((y - means_hat) * mask)
    return y_quant, y_quant + means_hat, means_hat, scales_hat


def dual_prior(sd, p, y, means, scales, compress=False):
    """video_model.py:169-216 (`forward_dual_prior`)."""
    _, _, h, w = y.shape
    m0, m1 = checkerboard_masks(h, w)
    y_0, y_1 = y.chunk(2, 1)
    mu_0, mu_1 = means.chunk(2, 1)
    sc_0, sc_1 = scales.chunk(2, 1)
    q00, yh00, mh00, sh00 = process_with_mask(y_0, mu_0, sc_0, m0)
    q11, yh11, mh11, sh11 = process_with_mask(y_1, mu_1, sc_1, m1)
    params = torch.cat((yh00, yh11, means, scales), dim=1)
    mu_0, sc_0, mu_1, sc_1 = prior_stack(sd, p + ".y_spatial_prior", params).chunk(4, 1)
    q01, yh01, mh01, sh01 = process_with_mask(y_0, mu_0, sc_0, m1)
    q10, yh10, mh10, sh10 = process_with_mask(y_1, mu_1, sc_1, m0)
    y_hat = torch.cat((yh00 + yh01, yh11 + yh10), dim=1)
    means_hat = torch.cat((mh00 + mh01, mh11 + mh10), dim=1)
    scales_hat = torch.cat((sh00 + sh01, sh11 + sh10), dim=1)
    if compress:
        return y_hat, q00 + q11, q01 + q10, sh00 + sh11, sh01 + sh10
    return y_hat, means_hat, scales_hat


class EntropyState:
    """The four compressai modules of one DMC (motion/frame x bottleneck/gaussian),
    rebuilt from the state dict."""

    def __init__(self, sd: SD, training=False):
        self.eb = {}
        self.gc = {}
        for key in ("motion_context_model", "frame_context_model"):
            ch = sd[f"{key}.entropy_bottleneck.quantiles"].shape[0]
            eb = EntropyBottleneck(ch)
            gc = GaussianConditional(None)
            with torch.no_grad():
                for n, prm in eb.named_parameters():
                    prm.copy_(sd[f"{key}.entropy_bottleneck.{n}"])
            eb.train(training)
            gc.train(training)
            self.eb[key] = eb
            self.gc[key] = gc

    def update(self, force=True):
        table = torch.exp(torch.linspace(math.log(0.11), math.log(256), 64))  # base_model.py:43-49
        for key in self.eb:
            self.gc[key].update_scale_table(table, force=force)
            self.eb[key].update(force=force)


def context_model_params(sd, key, y, y_ref, z_hat, context3=None):
    """hyper-decoder + prior fusion -> (means, scales): video_model.py:226-229 / 398-402."""
    params = hyper_decoder(sd, key + ".hyper_decoder", z_hat)
    if y_ref is None:
        n, _, hz, wz = z_hat.shape
        y_ref = torch.zeros(n, params.size(1) // 2, hz * 4, wz * 4, dtype=z_hat.dtype)
    if key == "frame_context_model":
        x = act(conv(sd, key + ".temporal_prior_encoder.0", context3, stride=2), 0.1)
        temporal = conv(sd, key + ".temporal_prior_encoder.2", x, stride=2)
        fused = prior_stack(sd, key + ".y_prior_fusion", torch.cat((temporal, params, y_ref), dim=1))
    else:
        fused = prior_stack(sd, key + ".y_prior_fusion", torch.cat((params, y_ref), dim=1))
    return fused.chunk(2, 1)


def context_model_forward(sd, ent: EntropyState, key, y, y_ref, context3=None, trace=None):
    """video_model.py:218-233 / 390-406.  `trace` (a dict) receives the pre-rounding values
    (z - median, y - mean) so a test can tell a genuine rounding tie from a wrong result."""
    eb, gc = ent.eb[key], ent.gc[key]
    z = hyper_encoder(sd, key + ".hyper_encoder", y)
    _, z_lik = eb(z)
    med = eb._get_medians()
    z_hat = quantize_ste(z - med) + med
    means, scales = context_model_params(sd, key, y, y_ref, z_hat, context3)
    y_hat, means_hat, scales_hat = dual_prior(sd, key, y, means, scales)
    _, y_lik = gc(y, scales_hat, means_hat)
    if trace is not None:
        trace[key] = {"z_pre": z - med, "z_hat": z_hat, "y_pre": y - means_hat, "y_hat": y_hat, "y": y,
                      "means": means, "scales": scales}
    return y_hat, {"y": y_lik, "z": z_lik}


def context_model_compress(sd, ent: EntropyState, key, y, y_ref, context3=None):
    """video_model.py:235-253 / 408-427."""
    eb, gc = ent.eb[key], ent.gc[key]
    z = hyper_encoder(sd, key + ".hyper_encoder", y)
    z_strings = eb.compress(z)
    z_hat = eb.decompress(z_strings, z.size()[-2:])
    means, scales = context_model_params(sd, key, y, y_ref, z_hat, context3)
    y_hat, qw0, qw1, sw0, sw1 = dual_prior(sd, key, y, means, scales, compress=True)
    s0 = gc.compress(qw0, gc.build_indexes(sw0))
    s1 = gc.compress(qw1, gc.build_indexes(sw1))
    return y_hat, {"strings": [s0, s1, z_strings], "shape": z.size()[-2:]}


def context_model_decompress(sd, ent: EntropyState, key, strings, shape, y_ref, context3=None):
    """video_model.py:255-291 / 429-466."""
    eb, gc = ent.eb[key], ent.gc[key]
    z_hat = eb.decompress(strings[2], shape)
    means, scales = context_model_params(sd, key, None, y_ref, z_hat, context3)
    _, _, h, w = means.shape
    m0, m1 = checkerboard_masks(h, w)
    mu_0, mu_1 = means.chunk(2, 1)
    sc_0, sc_1 = scales.chunk(2, 1)
    q_r0 = gc.decompress(strings[0], gc.build_indexes(sc_0 * m0 + sc_1 * m1))
    yh00 = (q_r0 + mu_0) * m0
    yh11 = (q_r0 + mu_1) * m1
    params = torch.cat((yh00, yh11, means, scales), dim=1)
    mu_0, sc_0, mu_1, sc_1 = prior_stack(sd, key + ".y_spatial_prior", params).chunk(4, 1)
    q_r1 = gc.decompress(strings[1], gc.build_indexes(sc_0 * m1 + sc_1 * m0))
    yh01 = (q_r1 + mu_0) * m1
    yh10 = (q_r1 + mu_1) * m0
    return torch.cat((yh00 + yh01, yh11 + yh10), dim=1)


# ------------------------------------------------------------------ DMC
def forward_inter(sd, ent, x_cur, dpb, trace=None):
    """video_model.py:556-579 (eval path, no pretrain flags)."""
    mv = me_spynet(sd, x_cur, dpb["x_ref"])
    y_mv = motion_encoder(sd, mv)
    y_mv_hat, mv_lik = context_model_forward(sd, ent, "motion_context_model", y_mv, dpb.get("y_mv_ref"),
                                             trace=trace)
    mv_hat = motion_decoder(sd, y_mv_hat)
    c1, c2, c3, x_warp = motion_compensation(sd, mv_hat, dpb)
    y = contextual_encoder(sd, x_cur, c1, c2, c3)
    y_hat, fr_lik = context_model_forward(sd, ent, "frame_context_model", y, dpb.get("y_ref"), c3, trace=trace)
    res = contextual_decoder(sd, y_hat, c2, c3)
    feature, x_rec = recon_generation(sd, res, c1)
    return (x_rec, {"motion": mv_lik, "frame": fr_lik},
            {"y_mv_ref": y_mv_hat, "y_ref": y_hat, "feature_ref": feature, "x_ref": x_rec})


def forward(sd, ent, frames: List[torch.Tensor]):
    """video_model.py:515-554."""
    if not isinstance(frames, list):
        raise RuntimeError("Invalid number of frames")
    dpb = {"y_mv_ref": None, "y_ref": None, "feature_ref": None, "x_ref": frames[0]}
    recs, liks = [], []
    for i in range(1, len(frames)):
        x_rec, lik, ctx = forward_inter(sd, ent, frames[i], dpb)
        recs.append(x_rec)
        liks.append(lik)
        if len(frames) >= 3:
            dpb = dict(ctx)
    return {"x_hat": recs, "likelihoods": liks}


def encode_inter(sd, ent, x_cur, dpb):
    """video_model.py:581-599."""
    mv = me_spynet(sd, x_cur, dpb["x_ref"])
    y_mv = motion_encoder(sd, mv)
    y_mv_hat, mv_s = context_model_compress(sd, ent, "motion_context_model", y_mv, dpb.get("y_mv_ref"))
    mv_hat = motion_decoder(sd, y_mv_hat)
    c1, c2, c3, _ = motion_compensation(sd, mv_hat, dpb)
    y = contextual_encoder(sd, x_cur, c1, c2, c3)
    _, fr_s = context_model_compress(sd, ent, "frame_context_model", y, dpb.get("y_ref"), c3)
    return {"strings": {"motion": mv_s["strings"], "frame": fr_s["strings"]},
            "shape": {"motion": mv_s["shape"], "frame": fr_s["shape"]}}


def decode_inter(sd, ent, strings, shapes, dpb):
    """video_model.py:601-614."""
    y_mv_hat = context_model_decompress(sd, ent, "motion_context_model", strings["motion"],
                                        shapes["motion"], dpb.get("y_mv_ref"))
    mv_hat = motion_decoder(sd, y_mv_hat)
    c1, c2, c3, _ = motion_compensation(sd, mv_hat, dpb)
    y_hat = context_model_decompress(sd, ent, "frame_context_model", strings["frame"],
                                     shapes["frame"], dpb.get("y_ref"), c3)
    res = contextual_decoder(sd, y_hat, c2, c3)
    feature, x_rec = recon_generation(sd, res, c1)
    return x_rec, {"x_ref": x_rec, "feature_ref": feature, "y_ref": y_hat, "y_mv_ref": y_mv_hat}


# ------------------------------------------------------------------ metrics
def likelihood_bpp(likelihoods_list, num_pixels):
    """dmc/train.py:74-93: sum over frames/labels/fields of sum(log lik)/(-ln2 * num_pixels)."""
    total = 0.0
    for frame in likelihoods_list:
        for label in frame.values():
            for v in label.values():
                total = total + torch.log(v).sum(dim=(1, 2, 3)) / (-math.log(2) * num_pixels)
    return total


def psnr(x, rec):
    """dmc/test.py:205-206."""
    mse = torch.mean((x - rec) ** 2).item()
    return -10.0 * math.log10(mse)
