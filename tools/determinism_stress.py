#!/usr/bin/env python3
"""Run-to-run determinism of the coding path's stages while another host thread keeps the GPU busy on its
own stream (GPU only).  Every stage is run twice on the same inputs and compared bit for bit; the encoder and
the decoder must derive identical entropy parameters, so any difference here desynchronises the range coder.

    python tools/determinism_stress.py [--precision fp16x2] [--height 1080 --width 1920] [--frames 4]
"""
import argparse
import os
import sys
import threading

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def same(a, b):
    if isinstance(a, (tuple, list)):
        return all(same(x, y) for x, y in zip(a, b))
    return bool((a.contiguous().view(torch.int32) == b.contiguous().view(torch.int32)).all())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="fp16x2")
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--frames", type=int, default=4)
    ap.add_argument("--load", type=int, default=1, help="0: no concurrent load thread")
    ap.add_argument("--warp-reps", type=int, default=0, help="> 0: only repeat flow_warp this many times per scale")
    args = ap.parse_args()
    from deepvideocodec_amd import ops
    from deepvideocodec_amd.models import DMC
    from deepvideocodec_amd.models.layers import flow_warp
    from deepvideocodec_amd.testing import pad_to_multiple, seeded_state_dict, synthetic_gop

    dev = torch.device("cuda:0")
    net = DMC()
    net.load_state_dict(seeded_state_dict(seed=0))
    net = net.to(dev).eval()
    net.update(force=True)
    net.set_precision(args.precision)
    net.prepare()
    gop_a = [pad_to_multiple(f, 64).to(dev) for f in synthetic_gop(args.frames, args.height, args.width, seed=1)]
    gop_b = [pad_to_multiple(f, 64).to(dev) for f in synthetic_gop(4, args.height, args.width, seed=2)]
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
    if args.load:
        th.start()
    bad = {}
    if args.warp_reps:
        try:
            st = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(st), torch.no_grad():
                h, w = gop_a[0].shape[2:]
                for c, sc in ((64, 1), (64, 2), (64, 4), (3, 1)):
                    src = torch.randn(1, c, h // sc, w // sc, device=dev)
                    flow = torch.nn.functional.interpolate(torch.randn(1, 2, h // sc // 16, w // sc // 16, device=dev) * 2,
                                                           size=(h // sc, w // sc), mode="bilinear").contiguous()
                    ref = ops.flow_warp(src, flow)
                    st.synchronize()
                    nbad = 0
                    for _ in range(args.warp_reps):
                        out = ops.flow_warp(src, flow)
                        st.synchronize()
                        d = out.view(torch.int32) != ref.view(torch.int32)
                        if bool(d.any()):
                            nbad += 1
                            idx = d.nonzero()[:4].tolist()
                            print(f"  {c}ch 1/{sc}: {int(d.sum())} words differ, first at {idx}: "
                                  f"{[(float(out[tuple(i)]), float(ref[tuple(i)])) for i in idx]}", flush=True)
                    print(f"flow_warp {c}ch scale 1/{sc}: {nbad} of {args.warp_reps} repeats differ", flush=True)
        finally:
            stop.set()
            if args.load:
                th.join()
        return
    try:
        st = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(st), torch.no_grad():
            dpb = {"x_ref": gop_a[0], "feature_ref": None, "y_ref": None, "y_mv_ref": None}
            for fi, x in enumerate(gop_a[1:], start=1):
                def check(name, fn):
                    r1 = fn()
                    r2 = fn()
                    st.synchronize()
                    ok = same(r1, r2)
                    if not ok:
                        bad[name] = bad.get(name, 0) + 1
                    print(f"frame {fi} {name:28s} {'same' if ok else 'DIFFERENT'}", flush=True)
                    return r1
                mv = check("optic_flow", lambda: net.optic_flow(x, dpb["x_ref"]))
                y_mv = check("motion_encoder", lambda: net.motion_encoder(mv))
                y_mv_hat = check("motion ctx compress (y_hat)",
                                 lambda: net.motion_context_model._compress_device(y_mv, dpb["y_mv_ref"], None)[0])
                mv_hat = check("motion_decoder", lambda: net.motion_decoder(y_mv_hat))
                feats = check("feature extractor", lambda: net.multi_scale_feature_extractor(dpb))
                check("flow_warp scale 1", lambda: flow_warp(feats[0], mv_hat))
                check("flow_warp x_ref", lambda: flow_warp(dpb["x_ref"], mv_hat))
                ctx = check("motion_compensation", lambda: net.motion_compensation(mv_hat, dpb)[:3])
                y = check("contextual_encoder", lambda: net.contextual_encoder(x, *net.motion_compensation(mv_hat, dpb)[:3]))
                c3 = net.motion_compensation(mv_hat, dpb)[2]
                check("frame ctx compress (y_hat)", lambda: net.frame_context_model._compress_device(y, dpb["y_ref"], c3)[0])
                enc, rec, info = net.encode_inter_with_recon(x, dpb)
                dpb = dict(info)
                dpb["x_ref"] = ops.clamp(rec, 0.0, 1.0)
    finally:
        stop.set()
        if args.load:
            th.join()
    print("DIFFERENT stages:", bad if bad else "none")


if __name__ == "__main__":
    main()
