#!/usr/bin/env python3
"""Per-shape time table of the convolution launches of one coded P frame (encode_inter +
decode_inter at 1080p by default): which layer shapes the frame time goes to and at what
fraction of the fp32 MFMA peak each runs.  Events bracket every launch on the launch stream.

    python tools/layer_breakdown.py [--height 1080 --width 1920]
"""
import argparse
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import conv_variant  # noqa: E402
from deepvideocodec_amd import ops  # noqa: E402
from deepvideocodec_amd.models import DMC  # noqa: E402
from deepvideocodec_amd.testing import pad_to_multiple, seeded_state_dict, synthetic_gop  # noqa: E402

PEAK = 157.3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    args = ap.parse_args()
    net = DMC()
    net.load_state_dict(seeded_state_dict(seed=0))
    net = net.cuda().eval()
    net.update(force=True)
    frames = [pad_to_multiple(f, 64).cuda() for f in synthetic_gop(3, args.height, args.width, seed=0)]
    recs = []
    orig = ops.conv2d

    def probed(srcs, weight, bias, stride=1, **kw):
        s = (srcs,) if isinstance(srcs, torch.Tensor) else srcs
        n, _, h, w = s[0].shape
        cout, cin, k, _ = weight.shape
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = orig(srcs, weight, bias, stride=stride, **kw)
        e1.record()
        recs.append(((k, stride, cin, cout, h, w), e0, e1))
        return out

    with torch.no_grad():
        dpb = {"x_ref": frames[0], "feature_ref": None, "y_ref": None, "y_mv_ref": None}
        for i, x in enumerate(frames[1:]):
            if i == 1:
                ops.conv2d = probed      # second P frame: feature_ref etc. are populated
            enc = net.encode_inter(x, dpb)
            rec, info = net.decode_inter(enc["strings"], enc["shape"], dpb)
            dpb = {"x_ref": ops.clamp(rec, 0.0, 1.0), "feature_ref": info["feature_ref"], "y_ref": info["y_ref"],
                   "y_mv_ref": info["y_mv_ref"]}
        ops.conv2d = orig
    torch.cuda.synchronize()
    agg = collections.defaultdict(lambda: [0, 0.0])
    for key, e0, e1 in recs:
        agg[key][0] += 1
        agg[key][1] += e0.elapsed_time(e1)
    total = sum(v[1] for v in agg.values())
    print(f"{len(recs)} conv launches, {total:.2f} ms per P frame (encode + decode)")
    print(f"{'k s  cin->cout @ HxW':34s} {'variant':>9s} {'calls':>5s} {'ms':>8s} {'share':>6s} {'TFLOP/s':>8s} {'frac':>5s}")
    for (k, s, cin, cout, h, w), (cnt, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        ho, wo = (h + 2 * (k // 2) - k) // s + 1, (w + 2 * (k // 2) - k) // s + 1
        fl = 2.0 * cout * ho * wo * cin * k * k * cnt
        mt, nt = conv_variant(1, cout, ho, wo, k, s)
        tf = fl / ms / 1e9
        print(f"k{k} s{s} {cin:4d}->{cout:4d} @{h:4d}x{w:<4d}          <{mt},{nt}> {cnt:5d} {ms:8.3f} {100 * ms / total:5.1f}% {tf:8.1f} {tf / PEAK:5.2f}")


if __name__ == "__main__":
    main()
