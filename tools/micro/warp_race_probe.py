#!/usr/bin/env python3
"""Repeats the full-resolution feature warp on the codec's own (noisy) decoded motion under concurrent load and
describes any words that differ between repeats (channel, tile, position inside the tile)."""
import os
import sys
import threading

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    precision = sys.argv[2] if len(sys.argv) > 2 else "fp16x2"
    from deepvideocodec_amd import ops
    from deepvideocodec_amd.models import DMC
    from deepvideocodec_amd.testing import pad_to_multiple, seeded_state_dict, synthetic_gop
    dev = torch.device("cuda:0")
    net = DMC()
    net.load_state_dict(seeded_state_dict(seed=0))
    net = net.to(dev).eval()
    net.update(force=True)
    net.set_precision(precision)
    net.prepare()
    gop_a = [pad_to_multiple(f, 64).to(dev) for f in synthetic_gop(2, 1080, 1920, seed=1)]
    gop_b = [pad_to_multiple(f, 64).to(dev) for f in synthetic_gop(4, 1080, 1920, seed=2)]
    stop = threading.Event()

    def load():
        torch.cuda.set_device(dev)
        st = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(st), torch.no_grad():
            while not stop.is_set():
                dpb = {"x_ref": gop_b[0], "feature_ref": None, "y_ref": None, "y_mv_ref": None}
                for x in gop_b[1:]:
                    if stop.is_set():
                        break
                    enc, rec, info = net.encode_inter_with_recon(x, dpb)
                    dpb = dict(info)
                    dpb["x_ref"] = ops.clamp(rec, 0.0, 1.0)
                st.synchronize()

    th = threading.Thread(target=load)
    th.start()
    try:
        st = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(st), torch.no_grad():
            dpb = {"x_ref": gop_a[0], "feature_ref": None, "y_ref": None, "y_mv_ref": None}
            x = gop_a[1]
            mv = net.optic_flow(x, dpb["x_ref"])
            y_mv = net.motion_encoder(mv)
            y_mv_hat = net.motion_context_model._compress_device(y_mv, None, None)[0]
            mv_hat = net.motion_decoder(y_mv_hat)
            feats = net.multi_scale_feature_extractor(dpb)
            src = feats[0].contiguous()
            st.synchronize()
            print("flow |mean|", float(mv_hat.abs().mean()), "max", float(mv_hat.abs().max()), flush=True)
            ref = ops.flow_warp(src, mv_hat)
            st.synchronize()
            nbad = 0
            for it in range(reps):
                a = ops.flow_warp(src, mv_hat)
                b = ops.flow_warp(src, mv_hat)      # back to back, as the stage check does
                st.synchronize()
                for nm, out in (("a", a), ("b", b)):
                    d = out.view(torch.int32) != ref.view(torch.int32)
                    if bool(d.any()):
                        nbad += 1
                        idx = d.nonzero()
                        ch = idx[:, 1].unique().tolist()
                        ty = (idx[:, 2] // 16).unique().tolist()
                        tx = (idx[:, 3] // 64).unique().tolist()
                        print(f"rep {it}{nm}: {int(d.sum())} words differ; channels {ch[:12]} tiles y {ty[:8]} x {tx[:8]}; "
                              f"rows in tile {(idx[:, 2] % 16).unique().tolist()} cols in tile {(idx[:, 3] % 64).unique().tolist()[:16]}",
                              flush=True)
                        i = tuple(idx[0].tolist())
                        print("   first", i, float(out[i]), float(ref[i]), flush=True)
            print(f"{nbad} of {2 * reps} launches differ", flush=True)
    finally:
        stop.set()
        th.join()


if __name__ == "__main__":
    main()
