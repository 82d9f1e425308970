#!/usr/bin/env python3
"""Where the host time of one 1080p P frame goes with a single GOP in flight (GPU only; diagnostic):
wall time inside the range coder calls and inside the waits for device -> host copies, per frame."""
import os
import sys
import time
from collections import defaultdict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepvideocodec_amd import ops  # noqa: E402
from deepvideocodec_amd.models import DMC, video_model  # noqa: E402
from deepvideocodec_amd.testing import pad_to_multiple, seeded_state_dict, synthetic_gop  # noqa: E402

acc = defaultdict(float)


def timed(name, fn):
    def w(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc[name] += time.perf_counter() - t0
    return w


def main():
    dev = torch.device("cuda", 0)
    net = DMC()
    net.load_state_dict(seeded_state_dict(0))
    net = net.to(dev).eval()
    net.update(force=True)
    net.prepare()
    for m in (net.motion_context_model, net.frame_context_model):
        tag = "motion" if m is net.motion_context_model else "frame"
        m.gaussian_conditional.decompress_symbols = timed(f"decode {tag} planes", m.gaussian_conditional.decompress_symbols)
        m.entropy_bottleneck.decompress_symbols = timed(f"decode {tag} z", m.entropy_bottleneck.decompress_symbols)
    video_model._download = timed("wait for index download (GPU finishing queued work)", video_model._download)
    video_model._PendingStrings.encode = timed("encode: wait + range coding", video_model._PendingStrings.encode)
    frames = [pad_to_multiple(f, 64).to(dev) for f in synthetic_gop(8, 1080, 1920, seed=0)]
    clamp = lambda t: ops.clamp(t, 0.0, 1.0)
    with torch.no_grad():
        for rep in range(2):
            acc.clear()
            dpb = {"x_ref": frames[0], "feature_ref": None, "y_ref": None, "y_mv_ref": None}
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            t_enc = t_dec = 0.0
            for x in frames[1:]:
                a = time.perf_counter()
                enc = net.encode_inter(x, dpb)
                b = time.perf_counter()
                rec, info = net.decode_inter(enc["strings"], enc["shape"], dpb)
                rec = clamp(rec)
                torch.cuda.synchronize()
                c = time.perf_counter()
                t_enc += b - a
                t_dec += c - b
                dpb = {"x_ref": rec, "feature_ref": info["feature_ref"], "y_ref": info["y_ref"], "y_mv_ref": info["y_mv_ref"]}
            n = len(frames) - 1
            total = time.perf_counter() - t0
    print(f"per frame: {total / n * 1e3:.2f} ms = encode_inter {t_enc / n * 1e3:.2f} + decode_inter (to completion) {t_dec / n * 1e3:.2f}")
    for k, v in sorted(acc.items()):
        print(f"   {k:55s} {v / n * 1e3:7.2f} ms")


if __name__ == "__main__":
    main()
