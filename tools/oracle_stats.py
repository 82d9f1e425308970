#!/usr/bin/env python3
"""Per-layer activation statistics and rate breakdown of the CPU oracle on a synthetic GOP
(diagnostic for the calibration gains).  usage: oracle_stats.py H W frames [seed]"""
import math
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepvideocodec_amd.testing import pad_to_multiple, seeded_state_dict, synthetic_gop  # noqa: E402
from oracle import dmc_oracle as O  # noqa: E402


def main():
    h, w, nf = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    seed = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    torch.set_num_threads(8)
    sd = seeded_state_dict(seed=0)
    ent = O.EntropyState(sd)
    stats = {}
    plain = O.conv

    def conv(sd_, name, x, stride=1):
        out = plain(sd_, name, x, stride)
        stats.setdefault(name, []).append((out.std().item(), out.abs().max().item()))
        return out
    O.conv = conv
    frames = [pad_to_multiple(f, 64) for f in synthetic_gop(nf, h, w, seed=seed)]
    n_pix = frames[0].shape[2] * frames[0].shape[3]
    dpb = {"x_ref": frames[0], "feature_ref": None, "y_ref": None, "y_mv_ref": None}
    with torch.no_grad():
        for i in range(1, nf):
            t0 = time.time()
            trace = {}
            x, lik, ctx = O.forward_inter(sd, ent, frames[i], dpb, trace=trace)
            dt = time.time() - t0
            b = {f"{a}.{k}": -torch.log2(v).sum().item() / n_pix for a, d in lik.items() for k, v in d.items()}
            print(f"frame {i}: {dt:.1f}s psnr {O.psnr(frames[i], x):.2f} bpp {sum(b.values()):.4f}",
                  {k: round(v, 4) for k, v in b.items()})
            for key, tr in trace.items():
                yp = tr["y_pre"]
                print(f"  {key}: |y-mu| mean {yp.abs().mean():.3f} max {yp.abs().max():.1f}  nonzero sym "
                      f"{(yp.round() != 0).float().mean():.4f}  z_pre std {tr['z_pre'].std():.3f} max {tr['z_pre'].abs().max():.1f}")
            dpb = dict(ctx)
    if os.environ.get("LAYERS"):
        for name, v in stats.items():
            print(f"{name:60s}", " ".join(f"{s:8.3f}/{m:8.1f}" for s, m in v))


if __name__ == "__main__":
    main()
