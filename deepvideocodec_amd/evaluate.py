"""Evaluation loop of one sequence (reference: dmc/test.py:106-241, `evaluate_one_video`).

Per frame: pad to a multiple of 64 (dmc/test.py:75-88, 159-160); every `gop`-th frame goes to the
intra codec and resets the decoded-picture buffer (:162-175), every other frame through
encode_inter -> decode_inter -> clamp -> buffer update (:186-195); bodies are appended to the
bitstream (:175, :200-203); PSNR = -10 log10(mean squared error) on the cropped reconstruction
(:205-206); bpp = 8 * bitstream bytes / (H * W * frames) (:213-217).

Differences, both forced by what is available offline: frames come as tensors (no cv2 / PNG
reading), and the intra codec is a parameter -- the reference downloads cheng2020_anchor weights.
`RawIntra` (frame passed through uncoded, 0 bytes written) is the stand-in DMC.forward itself
uses for frame 0 (dmc/models/video_model.py:522-534).

With a process group initialised, GOPs are sharded over ranks (see sharding.py) and the
statistics reduced; each rank returns its own bitstream segments.
"""
import io
import math
from collections import OrderedDict
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from . import bitstream, sharding


def pad(x: torch.Tensor, p: int = 64) -> torch.Tensor:
    """Centre zero-padding to a multiple of p (dmc/test.py:75-88)."""
    h, w = x.size(2), x.size(3)
    H, W = (h + p - 1) // p * p, (w + p - 1) // p * p
    left, top = (W - w) // 2, (H - h) // 2
    return F.pad(x, (left, W - w - left, top, H - h - top), mode="constant", value=0)


def crop(x: torch.Tensor, size) -> torch.Tensor:
    """Inverse of pad (dmc/test.py:91-103)."""
    H, W = x.size(2), x.size(3)
    h, w = size
    left, top = (W - w) // 2, (H - h) // 2
    return x[:, :, top:top + h, left:left + w]


class RawIntra:
    """Intra stand-in: the frame itself is the reference picture and nothing is written."""

    def __call__(self, x_pad: torch.Tensor) -> Tuple[torch.Tensor, Optional[Dict]]:
        return x_pad, None


def _psnr(x: torch.Tensor, rec: torch.Tensor) -> float:
    if x.is_cuda:
        from . import ops
        sse = ops.sq_err_sum(x.contiguous(), rec.contiguous()).item()
    else:
        sse = ((x - rec) ** 2).sum().item()
    mse = sse / x.numel()
    return -10.0 * math.log10(mse) if mse > 0 else float("inf")


def code_gop(net, frames: Sequence[torch.Tensor], fd, intra: Callable, clamp: Callable) -> Tuple[int, float, int]:
    """Code one GOP into fd; returns (bytes written, sum of per-frame PSNR, frames counted).

    An intra frame that was passed through uncoded (`RawIntra`: zero bytes, zero error, PSNR infinite)
    is left out of the PSNR sum and of the frame count, so the statistics describe the frames that were
    actually coded; with a real intra codec every frame counts, as in the reference (dmc/test.py:205-217)."""
    size = (frames[0].size(2), frames[0].size(3))
    nbytes, sum_psnr, counted, dpb = 0, 0.0, 0, None
    for fi, x in enumerate(frames):
        x_pad = pad(x)
        if fi == 0:
            rec_pad, enc = intra(x_pad)
            dpb = {"x_ref": rec_pad, "feature_ref": None, "y_ref": None, "y_mv_ref": None}
            if enc is None:
                continue
            nbytes += bitstream.write_body(fd, enc["shape"], enc["strings"])
        else:
            enc = net.encode_inter(x_pad, dpb)
            rec_pad, info = net.decode_inter(enc["strings"], enc["shape"], dpb)
            rec_pad = clamp(rec_pad)
            dpb = {"x_ref": rec_pad, "feature_ref": info["feature_ref"], "y_ref": info["y_ref"],
                   "y_mv_ref": info["y_mv_ref"]}
            nbytes += bitstream.write_p_frame(fd, enc)
        sum_psnr += _psnr(x, crop(rec_pad, size))
        counted += 1
    return nbytes, sum_psnr, counted


def evaluate_sequence(net, frames: Sequence[torch.Tensor], gop: int = 32, intra: Optional[Callable] = None,
                      name: str = "sequence", quality: int = 0) -> Tuple[OrderedDict, List[Tuple[int, bytes]]]:
    """Code a sequence GOP by GOP; returns (stats, this rank's [(gop index, bitstream segment)]).

    stats has the reference's keys (dmc/test.py:219-223): video, quality, bpp, psnr -- over all ranks and
    over the frames that were coded (see code_gop)."""
    import torch.distributed as dist
    intra = intra or RawIntra()
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)
    if frames[0].is_cuda:
        from . import ops
        clamp = lambda t: ops.clamp(t, 0.0, 1.0)
    else:
        clamp = lambda t: t.clamp(0, 1)
    ranges = sharding.gop_ranges(len(frames), gop)
    segments, nbytes, sum_psnr, count = [], 0, 0.0, 0
    had_lookahead = getattr(net, "decode_lookahead", None)
    if had_lookahead is not None:
        net.decode_lookahead = True        # code_gop decodes every frame right after encoding it
    try:
        with torch.no_grad():
            for g in sharding.gops_of_rank(len(ranges), rank, world):
                s, e = ranges[g]
                fd = io.BytesIO()
                b, p, c = code_gop(net, frames[s:e], fd, intra, clamp)
                segments.append((g, fd.getvalue()))
                nbytes, sum_psnr, count = nbytes + b, sum_psnr + p, count + c
    finally:
        if had_lookahead is not None:      # the caller's setting comes back, whatever happened
            net.decode_lookahead = had_lookahead
    dev = frames[0].device if frames[0].is_cuda else None
    nbytes, sum_psnr, count = sharding.reduce_stats(nbytes, sum_psnr, count, device=dev)
    h, w = frames[0].size(2), frames[0].size(3)
    count = max(count, 1)
    stats = OrderedDict(video=name, quality=quality, bpp=nbytes * 8.0 / (h * w * count), psnr=sum_psnr / count)
    return stats, segments


def decode_gop(net, fd, first_frame_pad: torch.Tensor, n_frames: int, clamp: Callable) -> List[torch.Tensor]:
    """Decode the P frames of one GOP segment written by code_gop with RawIntra (the intra picture is
    supplied by the caller); returns the padded reconstructions of frames 1..n_frames-1."""
    dpb = {"x_ref": first_frame_pad, "feature_ref": None, "y_ref": None, "y_mv_ref": None}
    out = []
    with torch.no_grad():
        for _ in range(1, n_frames):
            strings, shapes = bitstream.read_p_frame(fd)
            rec, info = net.decode_inter(strings, shapes, dpb)
            rec = clamp(rec)
            dpb = {"x_ref": rec, "feature_ref": info["feature_ref"], "y_ref": info["y_ref"], "y_mv_ref": info["y_mv_ref"]}
            out.append(rec)
    return out
