"""Per-shape breakdown of the convolution time of one coded P frame pair (encode_inter + decode_inter)
at 1088x1920: every convolution launch is bracketed with events and tallied by (kernel size, stride,
Cin, Cout, H, W).  Prints the shapes by share of the total convolution time with their achieved TFLOP/s.
The decoder's side-stream work is kept on the one stream here, so that no other kernel runs beside a
bracketed launch (--overlap 1 restores the coding loop's two-stream decoder).

    python tools/conv_breakdown.py [--height 1080 --width 1920] [--top 40]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--top", type=int, default=40)
    ap.add_argument("--overlap", type=int, default=0)
    ap.add_argument("--precision", default="fp32", help="fp32 / bf16 / bf16x2 / fp16x2 (DMC.set_precision)")
    args = ap.parse_args()
    from bench import conv_variant
    from deepvideocodec_amd import ops
    from deepvideocodec_amd.models import DMC, video_model
    from deepvideocodec_amd.testing import pad_to_multiple, seeded_state_dict, synthetic_gop

    dev = torch.device("cuda:0")
    net = DMC()
    net.load_state_dict(seeded_state_dict(seed=0))
    net = net.to(dev).eval()
    net.update(force=True)
    net.set_precision(args.precision)
    if not args.overlap:
        video_model._DECODE_OVERLAP = False
    frames = [pad_to_multiple(f, 64).to(dev) for f in synthetic_gop(4, args.height, args.width, seed=5)]
    records = []
    orig = ops.conv2d
    probing = [False]

    def probed(srcs, weight, bias, stride=1, **kw):
        if not probing[0]:
            return orig(srcs, weight, bias, stride=stride, **kw)
        s = (srcs,) if isinstance(srcs, torch.Tensor) else srcs
        n, _, h, w = s[0].shape
        cout, cin, k, _ = weight.shape
        ho, wo = (h + 2 * (k // 2) - k) // stride + 1, (w + 2 * (k // 2) - k) // stride + 1
        thin_ok = (len(s) == 1 and not kw.get("pixel_shuffle") and kw.get("res_scale") is None
                   and not kw.get("precision"))
        mt, nt = conv_variant(n, cout, ho, wo, k, stride, thin_ok, bool(kw.get("pixel_shuffle")))
        extras = "".join(c for c, on in (("c", len(s) > 1), ("r", kw.get("residual") is not None),
                                         ("R", kw.get("residual2") is not None), ("g", kw.get("res_scale") is not None),
                                         ("p", bool(kw.get("pixel_shuffle")))) if on)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = orig(srcs, weight, bias, stride=stride, **kw)
        e1.record()
        records.append(((k, stride, cin, cout, h, w, mt, nt, extras), 2.0 * n * cout * ho * wo * cin * k * k, e0, e1))
        return out

    ops.conv2d = probed
    dpb = {"x_ref": frames[0], "feature_ref": None, "y_ref": None, "y_mv_ref": None}
    with torch.no_grad():
        for i, x in enumerate(frames[1:]):
            probing[0] = i >= 1   # first P frame: warm-up
            enc = net.encode_inter(x, dpb)
            rec, info = net.decode_inter(enc["strings"], enc["shape"], dpb)
            dpb = {"x_ref": rec.clamp(0, 1), "feature_ref": info["feature_ref"], "y_ref": info["y_ref"],
                   "y_mv_ref": info["y_mv_ref"]}
    torch.cuda.synchronize()
    per = {}
    for key, fl, e0, e1 in records:
        d = per.setdefault(key, [0.0, 0.0, 0])
        d[0] += fl
        d[1] += e0.elapsed_time(e1) * 1e-3
        d[2] += 1
    tot_t = sum(v[1] for v in per.values())
    tot_f = sum(v[0] for v in per.values())
    print(f"{len(records)} launches over 2 P frames, {tot_t / 2 * 1e3:.2f} ms and {tot_f / 2 / 1e9:.0f} GFLOP per frame, "
          f"{tot_f / tot_t / 1e12:.1f} TFLOP/s")
    print("  k s  cin cout     h    w  mt nt extras  calls  us/call  TFLOP/s  share  cum")
    cum = 0.0
    for key, (fl, tm, cnt) in sorted(per.items(), key=lambda kv: -kv[1][1])[:args.top]:
        k, s, cin, cout, h, w, mt, nt, ex = key
        cum += tm / tot_t
        print(f"  {k} {s} {cin:4d} {cout:4d} {h:5d} {w:4d}  {mt:2d} {nt:2d} {ex:6s} {cnt:6d} {tm / cnt * 1e6:8.1f} "
              f"{fl / tm / 1e12:8.1f} {tm / tot_t * 100:6.2f} {cum * 100:5.1f}")


if __name__ == "__main__":
    main()
