#!/usr/bin/env python3
"""Micro-benchmark of single convolution shapes from the DMC graph (GPU only).

    python tools/conv_bench.py [--iters 20] [--shapes main]

Prints achieved TFLOP/s (algorithmic FLOPs / event time on the launch stream) per shape.
Used under rocprofv3 to collect PMC counters for one kernel variant at a time.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepvideocodec_amd import ops  # noqa: E402

# name: (k, stride, cin, cout, H, W)
SHAPES = {
    "main": (3, 1, 64, 64, 1088, 1920),
    "c128_64": (3, 1, 128, 64, 1088, 1920),
    "half64": (3, 1, 64, 64, 544, 960),
    "half128_256": (3, 1, 128, 256, 544, 960),
    "quarter128": (3, 1, 128, 128, 272, 480),
    "eighth64": (3, 1, 64, 64, 136, 240),
    "lat480_384": (3, 1, 480, 384, 68, 120),
    "lat288": (3, 1, 288, 288, 68, 120),
    "lat192": (3, 1, 192, 192, 68, 120),
    "z64": (3, 1, 64, 64, 17, 30),
    "s2_67_64": (3, 2, 67, 64, 1088, 1920),
    "s2_64_64h": (3, 2, 64, 64, 544, 960),
    "k7_8_32": (7, 1, 8, 32, 1088, 1920),
    "k7_32_64": (7, 1, 32, 64, 1088, 1920),
    "k7_64_32": (7, 1, 64, 32, 1088, 1920),
    "k7_32_16": (7, 1, 32, 16, 1088, 1920),
    "k7_16_2": (7, 1, 16, 2, 1088, 1920),
    "k1_64_32": (1, 1, 64, 32, 1088, 1920),
    "k1_64_128h": (1, 1, 64, 128, 544, 960),
    "c64_32": (3, 1, 64, 32, 1088, 1920),
    "c32_32": (3, 1, 32, 32, 1088, 1920),
    "c96_64": (3, 1, 96, 64, 1088, 1920),
    "c64_3": (3, 1, 64, 3, 1088, 1920),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--shapes", default="all")
    ap.add_argument("--precision", type=int, default=0, help="0 fp32, 1 bf16, 2 bf16 hi+lo, 3 fp16 hi+lo (k 1/3 only)")
    ap.add_argument("--wgrad", type=int, default=0, help="1: time the weight-gradient kernel of the shape instead")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--residuals", type=int, default=0, help="residual operands added in the epilogue (0, 1 or 2; stride 1)")
    ap.add_argument("--custom", default="", help="extra shapes 'k,s,cin,cout,H,W;...' (run instead of --shapes)")
    args = ap.parse_args()
    names = list(SHAPES) if args.shapes == "all" else args.shapes.split(",")
    if args.custom:
        names = []
        for spec in args.custom.split(";"):
            v = tuple(int(t) for t in spec.split(","))
            SHAPES["x" + "_".join(map(str, v))] = v
            names.append("x" + "_".join(map(str, v)))
    torch.manual_seed(0)
    total_t = total_f = 0.0
    for name in names:
        k, s, cin, cout, h, w = SHAPES[name]
        x = torch.randn(args.batch, cin, h, w, device="cuda")
        wt = torch.randn(cout, cin, k, k, device="cuda") * 0.05
        b = torch.randn(cout, device="cuda")
        bf = args.precision if cout > 4 else 0
        if args.wgrad:
            ho, wo = (h + 2 * (k // 2) - k) // s + 1, (w + 2 * (k // 2) - k) // s + 1
            g = torch.randn(args.batch, cout, ho, wo, device="cuda")
            for _ in range(2):
                ops.conv2d_wgrad(x, g, wt.shape, s, with_bias=True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                ops.conv2d_wgrad(x, g, wt.shape, s, with_bias=True)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / args.iters
            flops = 2.0 * args.batch * cout * ho * wo * cin * k * k
            print(f"{name:14s} wgrad k{k} s{s} {cin:4d}->{cout:4d} @{args.batch}x{h}x{w}: {ms * 1e3:9.1f} us  "
                  f"{flops / ms / 1e9:7.1f} TFLOP/s")
            total_t += ms
            total_f += flops
            continue
        packed = ops.pack_conv_weight(wt, s, precision=bf)
        res = {}
        if args.residuals >= 1 and s == 1:
            res["residual"] = torch.randn(args.batch, cout, h, w, device="cuda")
        if args.residuals >= 2 and s == 1:
            res["residual2"] = torch.randn(args.batch, cout, h, w, device="cuda")
        out = ops.conv2d(x, wt, b, stride=s, packed=packed, precision=bf, **res)
        for _ in range(2):
            ops.conv2d(x, wt, b, stride=s, packed=packed, out=out, precision=bf, **res)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            ops.conv2d(x, wt, b, stride=s, packed=packed, out=out, precision=bf, **res)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.iters
        flops = 2.0 * args.batch * cout * out.shape[2] * out.shape[3] * cin * k * k
        print(f"{name:14s} k{k} s{s} {cin:4d}->{cout:4d} @{h}x{w}: {ms * 1e3:9.1f} us  {flops / ms / 1e9:7.1f} TFLOP/s")
        total_t += ms
        total_f += flops
    print(f"sum: {total_t:.3f} ms, {total_f / total_t / 1e9:.1f} TFLOP/s")


if __name__ == "__main__":
    main()
