#!/usr/bin/env python3
"""Where a 3x3 convolution launch spends its time, per workgroup and per CU (GPU only; diagnostic).

Needs the diagnostic library:  make -C deepvideocodec_amd/csrc stamps
    DVC_HIP_LIB=deepvideocodec_amd/lib/libdvc_hip_stamps.so python tools/conv_stamps.py [cin cout H W]
Wave 0 of every workgroup stamps the 100 MHz real-time counter at entry, after the prologue (first chunk
staged), after the K loop and after the epilogue, plus its hardware id (XCD, SE, CU).  The script prints
the phase durations and, per CU, how the workgroups that ran there overlap.
"""
import ctypes as C
import os
import sys
from collections import defaultdict

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepvideocodec_amd import _lib, ops  # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    prec = 0
    for a in sys.argv[1:]:
        if a.startswith("--precision="):
            prec = int(a.split("=")[1])
    cin, cout, h, w = (int(v) for v in args[:4]) if len(args) >= 4 else (64, 64, 1088, 1920)
    lib = _lib.load()
    lib.dvc_debug_conv_stamps.argtypes = [C.c_void_p]
    x = torch.randn(1, cin, h, w, device="cuda")
    wt = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
    b = torch.randn(cout, device="cuda")
    packed = ops.pack_conv_weight(wt, 1, precision=prec)
    out = ops.conv2d(x, wt, b, packed=packed, precision=prec)
    import time
    t_end = time.time() + 2.0   # two seconds of back-to-back launches: the clock has ramped up from idle by then
    while time.time() < t_end:
        for _ in range(20):
            ops.conv2d(x, wt, b, packed=packed, out=out, precision=prec)
        torch.cuda.synchronize()
    n_wg = 1 << 16
    buf = torch.zeros(n_wg * 16, dtype=torch.int64, device="cuda")
    assert lib.dvc_debug_conv_stamps(buf.data_ptr()) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.conv2d(x, wt, b, packed=packed, out=out, precision=prec)
    e1.record()
    torch.cuda.synchronize()
    lib.dvc_debug_conv_stamps(None)
    st = buf.cpu().numpy().reshape(-1, 16)
    st = st[st[:, 0] != 0]
    t = st[:, :4].astype(np.float64) * 0.01          # us (100 MHz)
    t0 = t[:, 0].min()
    t -= t0
    cyc = (st[:, 5] - st[:, 4]).astype(np.float64)
    hw = st[:, 6]
    cu_key = ((hw >> 32) & 0xF) * 65536 + ((hw >> 8) & 0xFF)     # XCD, (SE, SH, CU)
    print(f"{len(st)} workgroups, launch {e0.elapsed_time(e1) * 1e3:.1f} us by events, "
          f"{t[:, 3].max():.1f} us first entry -> last exit; {len(set(cu_key.tolist()))} distinct CUs")
    dur = t[:, 3] - t[:, 0]
    print(f"per workgroup: total {dur.mean():.2f} us (min {dur.min():.2f} max {dur.max():.2f}); prologue "
          f"{(t[:, 1] - t[:, 0]).mean():.2f}, K loop {(t[:, 2] - t[:, 1]).mean():.2f}, epilogue {(t[:, 3] - t[:, 2]).mean():.2f}; "
          f"clock {np.median(cyc / np.maximum(dur, 1e-9)) / 1e3:.2f} GHz")
    tiles = st[:, 8:16].astype(np.float64) * 0.01 - t0
    if os.environ.get("DVC_STAMP_CHUNKS") == "wsp":
        # wave-specialised kernel: slots 8..10 multiplying wave 0 (second chunk of a tile: start, MFMAs done, barrier
        # passed); staging wave 4: slots 11, 12 its streaming interval (third chunk: start, requests issued and weight
        # slabs landed), slots 14, 15, 13 its converting interval (second chunk: start, LDS writes landed, barrier passed)
        c = st[:, 8:16].astype(np.float64)
        for nm, sg in (("multiply: the chunk's 216 MFMAs", c[:, 1] - c[:, 0]), ("multiply: barrier wait", c[:, 2] - c[:, 1]),
                       ("stage, streaming interval: requests issued, weights landed", c[:, 4] - c[:, 3]),
                       ("stage, converting interval: convert + commit", c[:, 7] - c[:, 6]),
                       ("stage, converting interval: barrier wait", c[:, 5] - c[:, 7])):
            print(f"  {nm:62s} {sg.mean():9.0f} {np.median(sg):9.0f} {np.percentile(sg, 90):9.0f}  (shader cycles: mean / median / p90)")
    elif os.environ.get("DVC_STAMP_CHUNKS") == "stepped":
        # reduced-operand stepped K loop (library built with -DDVC_STAMP_CHUNKS): slots 8..15 are shader-clock stamps
        # of the second chunk: before / after the MFMAs of tap rows 0, 1, 2; after the last row's wait; after its barrier
        c = st[:, 8:16].astype(np.float64)
        names = ["row 0 MFMAs", "row 0 end -> row 1 MFMAs (wait, barrier, requests)", "row 1 MFMAs",
                 "row 1 end -> row 2 MFMAs (+ convert / commit next chunk)", "row 2 MFMAs", "row 2 wait", "row 2 barrier"]
        segs = [c[:, i + 1] - c[:, i] for i in range(7)]
        print("second K chunk of wave 0, shader cycles (mean / median / p90):")
        for nm, sg in zip(names, segs):
            print(f"  {nm:58s} {sg.mean():9.0f} {np.median(sg):9.0f} {np.percentile(sg, 90):9.0f}")
    elif os.environ.get("DVC_STAMP_CHUNKS"):
        # library built with -DDVC_STAMP_CHUNKS: slots 8..15 are shader-clock stamps inside K chunks 2 and 3
        c = st[:, 8:16].astype(np.float64)
        names = ["start the next chunk's loads", "tap loop (MFMAs)", "commit rest of next chunk to LDS", "wait for DMA",
                 "barrier", "chunk 3: start .. tap group 10", "chunk 3: tap groups 10..25, per group"]
        segs = [c[:, 1] - c[:, 0], c[:, 2] - c[:, 1], c[:, 3] - c[:, 2], c[:, 4] - c[:, 3], c[:, 5] - c[:, 4],
                c[:, 6] - c[:, 5], (c[:, 7] - c[:, 6]) / 16.0]
        print("K chunk phases of wave 0, shader cycles (mean / median / p90):")
        for nm, sg in zip(names, segs):
            print(f"  {nm:34s} {sg.mean():9.0f} {np.median(sg):9.0f} {np.percentile(sg, 90):9.0f}")
    elif (st[:, 8:16] != 0).all():
        steps = np.diff(np.concatenate([t[:, 2:3], tiles], axis=1), axis=1)
        print("epilogue, per output tile of wave 0 (us, mean):", steps.mean(axis=0).round(2), "clock in K loop n/a")
    start_spread = np.sort(t[:, 0])
    print("entry times (us), percentiles 0/10/50/90/100:", np.percentile(start_spread, [0, 10, 50, 90, 100]).round(1))
    by_cu = defaultdict(list)
    for i, k in enumerate(cu_key.tolist()):
        by_cu[k].append(i)
    # per CU: fraction of the launch during which 0 / 1 / 2 workgroups are in their K loop
    grid = np.arange(0, t[:, 3].max(), 0.05)
    occ_hist = np.zeros(4)
    idle_gaps = []
    for k, idx in by_cu.items():
        inloop = np.zeros_like(grid)
        resident = np.zeros_like(grid)
        for i in idx:
            inloop += (grid >= t[i, 1]) & (grid < t[i, 2])
            resident += (grid >= t[i, 0]) & (grid < t[i, 3])
        for c in range(3):
            occ_hist[c] += (inloop == c).sum()
        occ_hist[3] += (resident < 2).sum()
    occ_hist /= len(by_cu) * len(grid)
    print(f"per CU, share of the launch with 0 / 1 / 2 workgroups inside the K loop: {occ_hist[0]:.3f} / {occ_hist[1]:.3f} / "
          f"{occ_hist[2]:.3f}; with fewer than 2 workgroups resident: {occ_hist[3]:.3f}")
    # one CU's timeline
    k = sorted(by_cu, key=lambda k: -len(by_cu[k]))[len(by_cu) // 2]
    print(f"timeline of one CU ({len(by_cu[k])} workgroups): entry / loop start / loop end / exit (us)")
    for i in sorted(by_cu[k], key=lambda i: t[i, 0]):
        print("   " + "  ".join(f"{v:8.2f}" for v in t[i]))


if __name__ == "__main__":
    main()
