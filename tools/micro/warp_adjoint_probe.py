#!/usr/bin/env python3
"""Times every flow_warp_backward call of one training step and describes its flow field: magnitude, the share of
8 x 32 output tiles whose sampled source window exceeds the kernel's LDS window, the share of clipped positions."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    dev = torch.device("cuda", 0)
    from deepvideocodec_amd import ops, train as TR
    from deepvideocodec_amd.models import DMC
    from deepvideocodec_amd.testing import seeded_state_dict, synthetic_gop
    net = DMC()
    net.load_state_dict(seeded_state_dict(seed=0))
    net = net.to(dev).train()
    crit = TR.RateDistortionLoss(lmbda=2048.0)
    opt, aux_opt = TR.configure_optimizers(net, 1e-4, 1e-3)
    frames = [f.to(dev) for f in synthetic_gop(3, 256, 256, seed=0, batch=batch)]
    TR.train_step(net, crit, frames, opt, aux_opt, 1.0, 45)
    real = ops.flow_warp_backward
    rows = []

    def probe(src, flow, g, ws=True, wf=True):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = real(src, flow, g, ws, wf)
        e1.record()
        n, c, h, w = src.shape
        ys = torch.arange(h, device=dev).view(1, h, 1) + flow[:, 1]
        xs = torch.arange(w, device=dev).view(1, 1, w) + flow[:, 0]
        clipped = ((xs <= 0) | (xs >= w - 1) | (ys <= 0) | (ys >= h - 1)).float().mean().item()
        xi = xs.clamp(0, w - 1).floor()
        yi = ys.clamp(0, h - 1).floor()
        th, tw = 8, 32
        if h % th == 0 and w % tw == 0:
            def box(v):
                t = v.view(n, h // th, th, w // tw, tw)
                return t.amax((2, 4)) - t.amin((2, 4)) + 2
            big = (box(xi) * box(yi) > 1280).float().mean().item()
            if c > 3 and h == 256:
                for (a, b2) in ((8, 32), (4, 32), (4, 16), (2, 32), (1, 32), (2, 16)):
                    def bx(v):
                        t = v.view(n, h // a, a, w // b2, b2)
                        return t.amax((2, 4)) - t.amin((2, 4)) + 2
                    sz = bx(xi) * bx(yi)
                    print(f"   tile {a}x{b2}: window > 1280 {(sz > 1280).float().mean().item():.3f}  > 2560 {(sz > 2560).float().mean().item():.3f}"
                          f"  > 5120 {(sz > 5120).float().mean().item():.3f}  > 10240 {(sz > 10240).float().mean().item():.3f}  median {sz.float().median().item():.0f}", flush=True)
        else:
            big = float("nan")
        rows.append((e0, e1, (n, c, h, w), ws, wf, flow.abs().mean().item(), flow.abs().max().item(), clipped, big))
        return out

    ops.flow_warp_backward = probe
    import deepvideocodec_amd.autograd as AG
    AG.ops.flow_warp_backward = probe
    TR.train_step(net, crit, frames, opt, aux_opt, 1.0, 45)
    torch.cuda.synchronize()
    for e0, e1, shp, ws, wf, mean, mx, clipped, big in rows:
        print(f"{str(shp):22s} gsrc={int(ws)} gflow={int(wf)}  {e0.elapsed_time(e1) * 1e3:9.1f} us  |flow| mean {mean:8.2f} max {mx:9.1f}"
              f"  clipped {clipped:.3f}  tiles past the window {big:.3f}", flush=True)


if __name__ == "__main__":
    main()
