#!/usr/bin/env python3
"""Micro-benchmark of the HBM-bound kernels of the DMC path at the 1080p shapes (GPU only): achieved TB/s of
algorithmic bytes per launch, event-timed on the launch stream.

    python tools/hbm_bench.py [--iters 20]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepvideocodec_amd import ops  # noqa: E402


def timed(fn, iters):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default="", help="run only the rows whose name starts with this (e.g. flow_warp)")
    args = ap.parse_args()
    torch.manual_seed(0)
    dev = "cuda"
    rows = []
    for c, h, w in ((64, 1088, 1920), (64, 544, 960), (64, 272, 480), (3, 1088, 1920)):
        src = torch.randn(1, c, h, w, device=dev)
        # smooth flow of a few pixels, as a motion field is (a white-noise flow breaks the coalescing of the gathers)
        flow = torch.nn.functional.interpolate(torch.randn(1, 2, h // 32, w // 32, device=dev) * 3, size=(h, w),
                                               mode="bilinear", align_corners=False).contiguous()
        out = torch.empty_like(src)
        t = timed(lambda: ops.flow_warp(src, flow, out=out), args.iters)
        rows.append((f"flow_warp {c}ch {h}x{w}", (2 * c + 2) * 4 * h * w, t))
    for c, h, w in ((64, 1088, 1920), (32, 1088, 1920), (64, 544, 960), (128, 272, 480)):
        x = torch.randn(1, c, h, w, device=dev)
        fc1 = torch.randn(c // 16, c, device=dev) * 0.1
        fc2 = torch.randn(c, c // 16, device=dev) * 0.1
        t = timed(lambda: ops.se_gate(x, fc1, fc2), args.iters)
        rows.append((f"se_gate {c}ch {h}x{w} (two kernels)", c * 4 * h * w, t))
    for c, h, w in ((32, 1088, 1920), (64, 544, 960)):
        x = torch.randn(1, c, h, w, device=dev)
        t = timed(lambda: ops.max_pool2(x), args.iters)
        rows.append((f"max_pool2 {c}ch {h}x{w}", c * 5 * h * w, t))
    a = torch.randn(1, 64, 1088, 1920, device=dev)
    b = torch.randn(1, 64, 1088, 1920, device=dev)
    o = torch.empty_like(a)
    t = timed(lambda: torch.add(a, b, out=o), args.iters)
    rows.append(("torch.add 64ch 1088x1920 (yardstick)", 3 * a.numel() * 4, t))
    for name, nbytes, t in rows:
        if not name.startswith(args.only):
            continue
        print(f"{name:42s} {t * 1e6:8.1f} us  {nbytes / t / 1e12:6.2f} TB/s")


if __name__ == "__main__":
    main()
