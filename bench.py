#!/usr/bin/env python3
"""Headline benchmark: DMC encode frames/s at 1080p, GOP = 12, on 1/2/4/8 MI355X.

A "step" is one GOP of synthetic 1080p video pushed through the reference's evaluation
loop body (dmc/test.py:186-195): for each of the 11 P frames encode_inter -> decode_inter ->
clamp -> dpb update (the I frame is passed through raw, as DMC.forward does).  Frames are
resident in HBM before the timed region.  value = coded P frames of all ranks / wall time.

  python bench.py --gpus 1 --steps 2 --warmup 1
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

GOPs are independent (each starts from an I frame with an empty decoded-picture buffer), so
ranks shard GOPs with no data-path collective ("weak" scaling: one GOP per rank per step).
The JSON line also carries the roofline of the dominant kernel (the fp32 MFMA convolution,
measured live with events on the launch stream) and a CPU baseline (the oracle's fp32
PyTorch path on the host cores, bounded sample).
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X dense fp32 matrix peak
PEAK_HBM_GBS = 8000.0           # HBM3E peak (roof of the --precision bf16 option)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None,
                    help="ranks (one process per GPU).  Started without a launcher and N > 1, the script starts "
                         "its own `torch.distributed.run --nproc-per-node N` as a child process")
    ap.add_argument("--dry-run", type=int, default=0,
                    help="1: rehearse the rank set-up, GOP assignment, barriers and the max-over-ranks timing on the "
                         "CPU with the gloo backend; no GPU is touched and no frame is coded")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--gop", type=int, default=12)
    ap.add_argument("--cpu-baseline", type=int, default=1, help="0 disables the CPU baseline leg")
    ap.add_argument("--cpu-scale", type=int, default=1,
                    help="CPU sample is the frame downscaled by this factor per side (1: measured at the full size)")
    ap.add_argument("--precision", default="fp32", choices=("fp32", "bf16", "bf16x2", "fp16x2"),
                    help="bf16 / bf16x2 / fp16x2: 16-bit-operand convolutions (not the parity path; reported with its deviation "
                         "from the fp32 path)")
    ap.add_argument("--probe-overlap", type=int, default=1,
                    help="0: keep the decoder's side-stream work on the main stream while the roofline probe runs. "
                         "Every launch then follows a host round trip more closely, and the idle time of the GPU "
                         "after those round trips lands inside the bracketed intervals (the dominant kernel reads "
                         "about 1.3 %% longer than in the profiler's kernel trace); 1 is the coding loop as timed")
    ap.add_argument("--alt-precision", default="fp16x2,bf16x2,bf16",
                    help="with --precision fp32: also time the same steps in these operand modes (comma-separated; 'none' "
                         "for no extra leg) and report them next to the headline with their deviation from the fp32 "
                         "results (never the headline value)")
    ap.add_argument("--training-alt", default="fp16x2",
                    help="with --training 1: also time the training steps with this operand mode in the forward and "
                         "data-gradient convolutions ('none': fp32 leg only); reported as training.alt")
    ap.add_argument("--reuse", type=int, default=0,
                    help="1: encoder-side reuse (reconstruction from the encoder pass, no separate decode); "
                         "not the headline configuration")
    ap.add_argument("--training", type=int, default=1,
                    help="1: after the coding legs, also time the training step of dmc/train.py:285-346 (batch 32 per GPU, "
                         "256x256 crops, T=3, joint stage; 1 warm-up + 3 steps) and report it as `training`; 0 skips it")
    ap.add_argument("--gops-in-flight", type=int, default=None,
                    help="independent GOPs coded concurrently per GPU (one host thread + HIP stream each); "
                         "a step codes this many GOPs per GPU.  Default: 3 where the rank has six or more host "
                         "cores to itself (cpus allowed / ranks of the node), fewer otherwise")
    return ap.parse_args()


def codec_gop(net, frames, clamp, reuse=False):
    """dmc/test.py:162-196 for one GOP; returns (bytes, list of reconstructions).  `reuse`: take the
    reconstruction from the encoder pass instead of decoding the strings again (same outputs)."""
    dpb = {"x_ref": frames[0], "feature_ref": None, "y_ref": None, "y_mv_ref": None}
    nbytes = 0
    recs = []
    for x in frames[1:]:
        if reuse:
            enc, rec, info = net.encode_inter_with_recon(x, dpb)
        else:
            enc = net.encode_inter(x, dpb)
            rec, info = net.decode_inter(enc["strings"], enc["shape"], dpb)
        rec = clamp(rec)
        dpb = {"x_ref": rec, "feature_ref": info["feature_ref"], "y_ref": info["y_ref"], "y_mv_ref": info["y_mv_ref"]}
        nbytes += sum(len(s[0]) for v in enc["strings"].values() for s in v)
        # the per-body framing of dmc/test.py:50-56: 3 uints per body + 1 uint per string
        nbytes += 2 * (3 * 4 + 3 * 4)
        recs.append(rec)
    return nbytes, recs


def conv_variant(n, cout, ho, wo, k, stride, thin_ok=False, shuffle=False):
    """(MT, NT) template arguments the library picks for a convolution (the tile is 4*NT x 32
    pixels x 32*MT output channels; NT is the largest of 4/2/1 that still gives >= 512 workgroups,
    3x3 stride-2 layers stop at 2) -- used only to attribute probe timings to kernel names.
    (0, 0) stands for the direct kernel of layers with at most 4 output channels."""
    if thin_ok and cout <= 4 and stride == 1 and k in (3, 7) and n * ((ho + 7) // 8) * ((wo + 31) // 32) >= 256:
        return 0, 0
    if k == 1 and stride == 1 and not shuffle and (ho * wo) % 32 == 0:
        ho, wo = ho * wo // 32, 32   # 1x1 stride 1 runs on the plane as rows of 32 pixels
    mt = 2 if cout > 32 else 1
    nt = 2 if (stride == 2 and k != 1) else 4
    cblocks, tx = (cout + 32 * mt - 1) // (32 * mt), (wo + 31) // 32
    while nt > 1 and ((ho + 4 * nt - 1) // (4 * nt)) * tx * cblocks * n < 480:
        nt //= 2
    if mt == 2 and nt == 1 and ((ho + 3) // 4) * tx * cblocks * n < 384:
        mt = 1   # half-block variant: 32-channel workgroups on 64-wide packed weights
    return mt, nt


def measured_traffic(kernel):
    """HBM-side bytes per launch of `kernel` from the newest committed counter summary
    (profiles/r*_pmc_traffic.json, written by tools/pmc_summary.py from separate rocprofv3
    --pmc FETCH_SIZE / WRITE_SIZE passes over this same command) with where it came from;
    (None, None) when there is none."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None, None
    with open(files[-1]) as f:
        doc = json.load(f)
    rec = doc.get("kernels", {}).get(kernel)
    if rec is None:
        return None, None
    return rec["traffic_bytes_per_launch"], {"file": os.path.relpath(files[-1], ROOT), "command": doc.get("command"),
                                             "note": "read from this committed counter summary (separate rocprofv3 --pmc "
                                                     "FETCH_SIZE / WRITE_SIZE passes, FETCH doubled for gfx950), not "
                                                     "measured in this run"}


class ConvProbe:
    """Brackets every convolution launch with events on the launch stream and tallies the
    algorithmic FLOPs (2 * N * Cout * Ho * Wo * Cin * k * k) per kernel variant."""

    def __init__(self):
        from deepvideocodec_amd import ops
        self.ops = ops
        self.orig = ops.conv2d
        self.orig_warp, self.orig_se = ops.flow_warp, ops.se_gate
        self.orig_pool, self.orig_sums = ops.max_pool2, ops.se_gate_from_sums
        self.records = []
        self.hbm_records = []   # (kernel, algorithmic bytes, start event, end event) of the HBM-bound kernels

    def __enter__(self):
        def probed(srcs, weight, bias, stride=1, **kw):
            s = (srcs,) if isinstance(srcs, torch.Tensor) else srcs
            n, _, h, w = s[0].shape
            cout, cin, k, _ = weight.shape
            ho, wo = (h + 2 * (k // 2) - k) // stride + 1, (w + 2 * (k // 2) - k) // stride + 1
            flops = 2.0 * n * cout * ho * wo * cin * k * k
            # compulsory bytes: every input, residual and output element once, plus the weights
            nres = sum(kw.get(r) is not None for r in ("residual", "residual2"))
            nbytes = 4.0 * (n * cin * h * w + (1 + nres) * n * cout * ho * wo + cout * cin * k * k)
            thin_ok = (len(s) == 1 and not kw.get("pixel_shuffle") and kw.get("res_scale") is None
                       and not kw.get("precision"))
            mt, nt = conv_variant(n, cout, ho, wo, k, stride, thin_ok, bool(kw.get("pixel_shuffle")))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = self.orig(srcs, weight, bias, stride=stride, **kw)
            e1.record()
            self.records.append(((k, stride, mt, nt), flops, e0, e1, nbytes))
            return out
        def bracket(name, nbytes, fn, *a, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*a, **kw)
            e1.record()
            self.hbm_records.append((name, nbytes, e0, e1))
            return out

        def probed_warp(src, flow, out=None):
            n, c, h, w = src.shape   # every source and output element once, plus the flow: (2 C + 2) * 4 bytes per pixel
            return bracket("flow_warp", 4.0 * n * (2 * c + 2) * h * w, self.orig_warp, src, flow, out=out)

        def probed_se(x, fc1, fc2):
            return bracket("se_gate (plane sums + gate)", 4.0 * x.numel(), self.orig_se, x, fc1, fc2)
        def probed_pool(x):   # every input element once, a quarter as many outputs
            return bracket("max_pool2", 4.0 * x.numel() * 1.25, self.orig_pool, x)

        def probed_sums(sums, h, w, fc1, fc2):   # the per-tile sums the convolution's epilogue wrote, folded and gated
            return bracket("se_gate_from_sums (fold of the conv epilogue's plane sums + gate)", 4.0 * sums.numel(),
                           self.orig_sums, sums, h, w, fc1, fc2)
        self.ops.conv2d = probed
        self.ops.flow_warp, self.ops.se_gate = probed_warp, probed_se
        self.ops.max_pool2, self.ops.se_gate_from_sums = probed_pool, probed_sums
        return self

    def __exit__(self, *a):
        self.ops.conv2d = self.orig
        self.ops.flow_warp, self.ops.se_gate = self.orig_warp, self.orig_se
        self.ops.max_pool2, self.ops.se_gate_from_sums = self.orig_pool, self.orig_sums

    def hbm_summary(self):
        """Per HBM-bound kernel: launches, algorithmic bytes and achieved GB/s over the probed frames."""
        torch.cuda.synchronize()
        per = {}
        for name, nbytes, e0, e1 in self.hbm_records:
            d = per.setdefault(name, [0.0, 0.0, 0])
            d[0] += nbytes
            d[1] += e0.elapsed_time(e1) * 1e-3
            d[2] += 1
        return {k: {"launches": v[2], "algorithmic_bytes": v[0], "ms": v[1] * 1e3, "achieved": v[0] / v[1] / 1e9,
                    "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": v[0] / v[1] / 1e9 / PEAK_HBM_GBS}
                for k, v in per.items() if v[1] > 0}

    def summary(self):
        torch.cuda.synchronize()
        per = {}
        for key, flops, e0, e1, nbytes in self.records:
            d = per.setdefault(key, [0.0, 0.0, 0, 0.0])
            d[0] += flops
            d[1] += e0.elapsed_time(e1) * 1e-3
            d[2] += 1
            d[3] += nbytes
        return per


def cpu_baseline(sd, args, threads):
    """The oracle's encode+decode of P frames on the host cores (fp32 PyTorch, the C++ range
    coder of this repo standing in for compressai's), on a down-scaled sample of the workload."""
    import numpy as np
    from deepvideocodec_amd.models.entropy_models import _CoderTables
    from deepvideocodec_amd.testing import pad_to_multiple, synthetic_gop
    from oracle import dmc_oracle as O
    from oracle import rans_oracle

    torch.set_num_threads(threads)
    ent = O.EntropyState(sd)
    ent.update(force=True)
    # route the oracle's per-module coder through the C++ range coder with that module's tables
    # (the pure-Python oracle coder would dominate the timing; compressai's coder is C++ too)
    import oracle.entropy_oracle as EO
    for m in list(ent.eb.values()) + list(ent.gc.values()):
        m._bench_tables = _CoderTables(m._quantized_cdf, m._cdf_length, m._offset)

    def compress(self, inputs, indexes, means=None):
        symbols = EO.quantize(inputs, "symbols", means)
        return [self._bench_tables.encode(symbols[i].numpy(), indexes[i].numpy()) for i in range(symbols.size(0))]

    def decompress(self, strings, indexes, dtype=torch.float, means=None):
        out = torch.empty(indexes.size(), dtype=dtype)
        for i, s in enumerate(strings):
            out[i] = torch.from_numpy(self._bench_tables.decode(s, indexes[i].numpy())).reshape(out[i].shape).to(dtype)
        return out if means is None else out + means

    EO._EntropyModelBase.compress = compress
    EO._EntropyModelBase.decompress = decompress

    def sample(scale, n_timed):
        h, w = args.height // scale, args.width // scale
        frames = [pad_to_multiple(f, 64) for f in synthetic_gop(1 + 1 + n_timed, h, w, seed=0)]
        dpb = {"x_ref": frames[0], "feature_ref": None, "y_ref": None, "y_mv_ref": None}
        times = []
        with torch.no_grad():
            for x in frames[1:]:
                t0 = time.perf_counter()
                e = O.encode_inter(sd, ent, x, dpb)
                rec, info = O.decode_inter(sd, ent, e["strings"], e["shape"], dpb)
                rec = rec.clamp(0, 1)
                times.append(time.perf_counter() - t0)
                dpb = {"x_ref": rec, "feature_ref": info["feature_ref"], "y_ref": info["y_ref"],
                       "y_mv_ref": info["y_mv_ref"]}
        t = sum(times[1:]) / len(times[1:])   # the first frame warms up the allocator / thread pool
        area = (frames[0].shape[2] * frames[0].shape[3]) / float(pad64(args.height) * pad64(args.width))
        return (1.0 / t) * area, t, tuple(frames[0].shape[2:]), len(times) - 1

    sc = max(1, args.cpu_scale)
    fps_full, t, shape, n = sample(sc, 1)
    how = "measured at the benchmark's frame size" if sc == 1 else "scaled by area to the benchmark's frame size"
    out = {"value": fps_full, "unit": "frames/s", "cores": threads, "kind": "port",
           "sample": f"oracle encode_inter+decode_inter of {n} P frame at {shape[0]}x{shape[1]} "
                     f"({t:.2f} s, after 1 warm-up frame), {how}; fp32, {threads} threads"}
    # the reference's own evaluation script pins one thread (dmc/test.py:27): a quarter-size frame, scaled by area
    torch.set_num_threads(1)
    fps1, t1, shape1, n1 = sample(4 * sc, 1)
    torch.set_num_threads(threads)
    out["single_thread"] = {"value": fps1, "cores": 1,
                            "sample": f"{n1} P frame at {shape1[0]}x{shape1[1]} ({t1:.2f} s), scaled by area (not measured "
                                      "at the full size: one thread needs minutes per 1080p frame)"}
    return out


def pad64(v):
    return (v + 63) // 64 * 64


def _coder_pool_workers():
    from deepvideocodec_amd.models import video_model
    return video_model._POOL._max_workers


PEAK_MFMA16_TFLOPS = 2500.0     # MI355X dense bf16 / fp16 matrix peak


def reduced_roofline(mode, probe):
    """Roofline of the 16-bit-operand modes from a probed frame pair: the dominant convolution variant and all
    convolution launches together, against both roofs that can bound them -- algorithmic bytes (fp32 tensors in
    HBM: inputs, residuals, outputs, weights, each once) per second against the HBM peak, and issued 16-bit MFMA
    FLOPs (3 products per term for the hi + lo split modes, 1 for bf16) against the dense 16-bit matrix peak.  The
    7x7 motion-estimation layers stay exact fp32 in every mode and are listed separately."""
    if probe is None:
        return None
    per = probe.summary()
    issued = 3.0 if mode in ("fp16x2", "bf16x2") else 1.0
    red = {k: v for k, v in per.items() if k[0] != 7 and k[2] != 0}     # 1x1 / 3x3 on the 16-bit MFMA
    f32 = {k: v for k, v in per.items() if k not in red}                # 7x7 layers, <= 4-output-channel layers
    if not red:
        return None
    (k, st, mt, nt), (fl, tm, cnt, nb) = max(red.items(), key=lambda kv: kv[1][1])
    rfl, rtm, rnb = (sum(v[i] for v in red.values()) for i in (0, 1, 3))
    ffl, ftm = (sum(v[i] for v in f32.values()) for i in (0, 1))
    def both(fl_, nb_, tm_):
        return {"hbm": {"achieved": nb_ / tm_ / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": nb_ / tm_ / 1e9 / PEAK_HBM_GBS},
                "mfma16": {"achieved": issued * fl_ / tm_ / 1e12, "peak": PEAK_MFMA16_TFLOPS, "unit": "TFLOP/s issued",
                           "frac": issued * fl_ / tm_ / 1e12 / PEAK_MFMA16_TFLOPS,
                           "algorithmic_tflops": fl_ / tm_ / 1e12}}
    out = {"bound": "hbm", "kernel": f"conv_mfma_f32_kernel<{k},{st},{mt},{nt}> ({mode} operands)", "launches": cnt,
           "avg_launch_ms": tm / cnt * 1e3, "algorithmic_bytes_per_launch": nb / cnt, "gflop_per_launch": fl / cnt / 1e9}
    out.update(both(fl, nb, tm))
    out["achieved"], out["peak"], out["unit"], out["frac"] = (out["hbm"]["achieved"], PEAK_HBM_GBS, "GB/s", out["hbm"]["frac"])
    out["all_reduced_convs"] = dict(both(rfl, rnb, rtm), ms_per_frame=rtm / 2 * 1e3)
    out["fp32_convs_left"] = {"ms_per_frame": ftm / 2 * 1e3, "achieved": ffl / max(ftm, 1e-12) / 1e12, "unit": "TFLOP/s",
                              "peak": PEAK_FP32_MFMA_TFLOPS, "frac": ffl / max(ftm, 1e-12) / 1e12 / PEAK_FP32_MFMA_TFLOPS}
    out["hbm_bound_kernels"] = probe.hbm_summary()
    return out


def training_leg(dev, world, rank, barrier, batch=32, size=256, frames=3, warmup=1, steps=3, precision="fp32"):
    """One GPU's share of the reference's training workload (dmc/train.sh:3-9, dmc/train.py:285-346): batch 32 of
    T=3 random 256 x 256 crops, joint stage (epoch >= 40), lambda 2048, Adam 1e-4 / aux 1e-3, clip 1.0.  With more
    than one rank the gradients are averaged by bucketed all-reduces before the optimizer steps.  `precision`: operand
    mode of the forward and data-gradient convolutions (weight gradients, everything else and all tensors stay fp32)."""
    from deepvideocodec_amd import train as TR
    from deepvideocodec_amd.models import DMC
    from deepvideocodec_amd.testing import seeded_state_dict, synthetic_gop
    torch.manual_seed(1000 + rank)   # the quantisation noise of the training forward pass: the same draws in every leg
    net = DMC()
    net.load_state_dict(seeded_state_dict(seed=0))
    net = net.to(dev).train()
    net.set_precision(precision, train=True)
    crit = TR.RateDistortionLoss(lmbda=2048.0)
    opt, aux_opt = TR.configure_optimizers(net, 1e-4, 1e-3)
    batches = [[f.to(dev) for f in synthetic_gop(frames, size, size, seed=rank * 100 + s, batch=batch)]
               for s in range(warmup + steps)]
    torch.cuda.reset_peak_memory_stats(dev)
    stats = None
    for s in range(warmup):
        stats = TR.train_step(net, crit, batches[s], opt, aux_opt, 1.0, 45)
    barrier()
    t0 = time.perf_counter()
    for s in range(warmup, warmup + steps):
        stats = TR.train_step(net, crit, batches[s], opt, aux_opt, 1.0, 45)
    barrier()
    el = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = t.item()
    # forward FLOPs of one P frame at 256 x 256: 0.2169 TFLOP; backward = data + weight gradient = 2x forward
    fwd_tflop = 0.2169 * (size * size) / (256.0 * 256.0) * (frames - 1) * batch
    tf = 3 * fwd_tflop / (el / steps)
    mem = torch.cuda.max_memory_allocated(dev) / 2 ** 30
    del net, opt, aux_opt, batches
    torch.cuda.empty_cache()
    return {"workload": f"train_step, batch {batch} per GPU x {world} GPU(s), T={frames}, {size}x{size} crops, joint stage, "
                        "lambda 2048, Adam + aux Adam, clip 1.0" + (", gradients all-reduced (RCCL)" if world > 1 else ""),
            "value": batch * world * steps / el, "unit": "samples/s", "ms_per_step": el / steps * 1e3, "steps": steps,
            "warmup": warmup, "global_batch": batch * world,
            "dtype": "f32" if precision == "fp32" else f"f32 tensors, accumulation and weight gradients; {precision} operands "
                                                       "in the forward and data-gradient convolutions",
            "roofline": {"bound": "mfma", "achieved": tf, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s per GPU",
                         "frac": tf / PEAK_FP32_MFMA_TFLOPS,
                         "flops": "3 x forward algorithmic FLOPs (forward + data gradient + weight gradient)"
                                  + ("" if precision == "fp32" else "; priced against the fp32 matrix peak although two thirds "
                                     "of them run on the 16-bit pipe: a comparison with the fp32 leg, not a utilisation")},
            "max_mem_gb": mem, "last_step": stats}


def dry_run(args, world, rank):
    """The multi-rank skeleton of main() without a GPU: same GOP numbering, barriers and timing
    reduction, gloo instead of RCCL.  Rank 0 prints the JSON line with the ranks that took part."""
    from deepvideocodec_amd import launch
    dist = launch.init_process_group(world, "gloo")
    G = max(1, args.gops_in_flight)
    mine = [(s * world + rank) * G + g for s in range(args.warmup + args.steps) for g in range(G)]
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * args.steps)
    if dist is not None:
        dist.barrier()
    elapsed = torch.tensor([time.perf_counter() - t0, float(rank)], dtype=torch.float64)
    ranks, gops = [rank], [mine]
    if dist is not None:
        tt = elapsed[:1].clone()
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed[0] = tt[0]
        ranks = [None] * world
        gops = [None] * world
        dist.all_gather_object(ranks, rank)
        dist.all_gather_object(gops, mine)
    if rank == 0:
        flat = sorted(g for gg in gops for g in gg)
        assert flat == list(range(len(flat))), "GOP numbering must cover 0..n-1 exactly once"
        n_p = args.gop - 1
        print(json.dumps({"metric": "encode frames/sec + bpp@PSNR, 1080p GOP=12, 1/2/4/8 MI355X", "dry_run": True,
                          "value": n_p * args.steps * world * G / elapsed[0].item(), "unit": "frames/s (no frame coded)",
                          "n_gpus": world, "world_size": world, "ranks": sorted(ranks), "backend": "gloo",
                          "steps": args.steps, "warmup": args.warmup, "scaling": "weak",
                          "gops": len(flat)}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    from deepvideocodec_amd import launch
    # N > 1 without a launcher: start the ranks as a child process (before anything touches the GPU)
    world = launch.resolve_world(args.gpus, os.path.abspath(__file__), sys.argv[1:], need_devices=not args.dry_run)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gops_in_flight is None:
        args.gops_in_flight = launch.default_gops_in_flight(launch.host_budget()["cores_per_rank"])
    if args.dry_run:
        return dry_run(args, world, rank)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    launch.init_process_group(world, "nccl", dev)

    from deepvideocodec_amd import ops
    from deepvideocodec_amd.models import DMC
    from deepvideocodec_amd.testing import pad_to_multiple, seeded_state_dict, synthetic_gop

    sd = seeded_state_dict(seed=0)
    net = DMC()
    net.load_state_dict(sd)
    net = net.to(dev).eval()
    net.update(force=True)
    net.set_precision(args.precision)
    net.decode_lookahead = True    # the loop below decodes every frame right after encoding it (dmc/test.py:186-188)

    def clamp(x):
        return ops.clamp(x, 0.0, 1.0)

    def make_gop(seed):
        return [pad_to_multiple(f, 64).to(dev) for f in synthetic_gop(args.gop, args.height, args.width, seed=seed)]

    n_p = args.gop - 1
    G = max(1, args.gops_in_flight)
    net.prepare()
    # step s, worker g of this rank codes GOP number (s * world + rank) * G + g
    gops = {s: [make_gop((s * world + rank) * G + g) for g in range(G)] for s in range(args.warmup + args.steps)}
    streams = [torch.cuda.Stream(device=dev) for _ in range(G)]
    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(max_workers=G)

    def run_step(s):
        """Code the step's G independent GOPs concurrently: the host-side range coding of one GOP
        overlaps the kernels of the others.  Returns (bytes, reconstructions of GOP 0)."""
        def work(g):
            torch.cuda.set_device(dev)
            with torch.cuda.stream(streams[g]):
                b, recs = codec_gop(net, gops[s][g], clamp, reuse=bool(args.reuse))
                streams[g].synchronize()
            return b, recs
        res = list(pool.map(work, range(G))) if G > 1 else [work(0)]
        return sum(r[0] for r in res), res[0][1]

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    torch.cuda.synchronize()
    for s in range(args.warmup):
        run_step(s)
    barrier()
    t0 = time.perf_counter()
    cpu0 = time.process_time()
    total_bytes = 0
    last_recs = None
    for s in range(args.warmup, args.warmup + args.steps):
        b, last_recs = run_step(s)
        total_bytes += b
    barrier()
    elapsed = time.perf_counter() - t0
    cpu_s = time.process_time() - cpu0
    import threading
    budget = launch.host_budget()
    host = {"process_cpu_s_per_frame": cpu_s / (n_p * args.steps * G), "host_cores_busy": cpu_s / elapsed,
            "threads": threading.active_count(), "coder_pool_workers": _coder_pool_workers(),
            "gop_threads": G, "cpu_count": os.cpu_count(), "cpus_allowed": budget["cpus_allowed"],
            "local_world_size": budget["local_world"], "cores_per_rank_budget": budget["cores_per_rank"],
            "note": "rank 0's process over the timed region: CPU seconds (all threads: GOP threads launching kernels "
                    "through ctypes, range-coder pool) per coded P frame, and their ratio to wall time = host cores "
                    "kept busy per rank"}
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = tt.item()

    frames_coded = n_p * args.steps * world * G
    fps = frames_coded / elapsed

    # ---- one GOP in flight (one host thread, one stream): what a single sequence sees; the decoder's host round
    # trips (range decoding between kernels) are not hidden behind other GOPs here
    single_fps = None
    if rank == 0:
        torch.cuda.synchronize()
        with torch.cuda.stream(streams[0]):
            codec_gop(net, gops[0][0][:3], clamp, reuse=bool(args.reuse))
            streams[0].synchronize()
            t2 = time.perf_counter()
            codec_gop(net, gops[0][0], clamp, reuse=bool(args.reuse))
            streams[0].synchronize()
        single_fps = n_p / (time.perf_counter() - t2)

    # ---- roofline of the convolution engine: one probed GOP frame pair (untimed), taken right after the timed
    # region, in the state the timed steps left the GPU in
    def probe_frame_pair():
        last = gops[args.warmup + args.steps - 1][0]
        from deepvideocodec_amd.models import video_model as _vm
        lookahead, overlap = net.decode_lookahead, _vm._DECODE_OVERLAP
        if not args.probe_overlap:
            net.decode_lookahead, _vm._DECODE_OVERLAP = False, False
        try:
            with ConvProbe() as pr:
                dpb = {"x_ref": last[0], "feature_ref": None, "y_ref": None, "y_mv_ref": None}
                for x in last[1:3]:
                    enc = net.encode_inter(x, dpb)
                    rec, info = net.decode_inter(enc["strings"], enc["shape"], dpb)
                    dpb = {"x_ref": clamp(rec), "feature_ref": info["feature_ref"], "y_ref": info["y_ref"],
                           "y_mv_ref": info["y_mv_ref"]}
        finally:
            net.decode_lookahead, _vm._DECODE_OVERLAP = lookahead, overlap
        return pr

    probe = probe_frame_pair() if rank == 0 else None

    # ---- the same steps once more per reduced-operand mode (reported beside the headline, never as it)
    alts = []
    alt_modes = [m for m in args.alt_precision.split(",") if m and m != "none"]
    if args.precision == "fp32":
        for mode in alt_modes:
            net.set_precision(mode)
            net.prepare()
            run_step(0)
            barrier()
            t1 = time.perf_counter()
            alt_bytes, alt_recs = 0, None
            for s in range(args.warmup, args.warmup + args.steps):
                b, alt_recs = run_step(s)
                alt_bytes += b
            barrier()
            alt_elapsed = time.perf_counter() - t1
            if world > 1:
                import torch.distributed as dist
                tt = torch.tensor([alt_elapsed], device=dev, dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                alt_elapsed = tt.item()
            alts.append((mode, alt_elapsed, alt_bytes, alt_recs, probe_frame_pair() if rank == 0 else None))
        net.set_precision("fp32")

    # ---- the training step of dmc/train.py:285-346 at the reference's stated size (batch 32 per GPU, 256 x 256 crops,
    # T = 3, joint stage): every rank steps, gradients are averaged over ranks (RCCL) when there is more than one
    training = None
    if args.training:
        training = training_leg(dev, world, rank, barrier)
        if args.training_alt != "none":
            # the same steps with split 16-bit operands in the forward / data-gradient convolutions: reported beside the
            # fp32 number, never instead of it
            training["alt"] = training_leg(dev, world, rank, barrier, precision=args.training_alt)
            a, b = training["alt"]["last_step"], training["last_step"]
            training["alt"]["d_loss_vs_fp32_rel"] = abs(a["loss"] - b["loss"]) / max(abs(b["loss"]), 1e-30)

    out = None
    if rank == 0:
        last = gops[args.warmup + args.steps - 1][0]
        n_pix = args.height * args.width
        from deepvideocodec_amd.testing import crop_to
        psnrs = []
        for x, r in zip(last[1:], last_recs):
            sse = ops.sq_err_sum(crop_to(x, (args.height, args.width)).contiguous(),
                                 crop_to(r, (args.height, args.width)).contiguous()).item()
            psnrs.append(-10 * math.log10(sse / (3 * n_pix)))
        bpp = total_bytes * 8 / (n_pix * n_p * args.steps * G)
        alt_notes = {
            "fp16x2": "1x1/3x3 convolutions multiply fp16 hi+lo split operands (22 significant bits) on the fp16 MFMA and "
                      "accumulate in fp32",
            "bf16x2": "1x1/3x3 convolutions multiply bf16 hi+lo split operands (16 significant bits) on the bf16 MFMA and "
                      "accumulate in fp32; inside the 1e-3 dB / 1e-4 bpp bar against the CPU oracle frame by frame",
            "bf16": "1x1/3x3 convolution operands rounded to bf16, fp32 accumulate; outside the parity bar frame by frame "
                    "(up to 7e-3 dB against the CPU oracle), deviations average out over a GOP",
        }
        alt_outs = []
        for mode, alt_elapsed, alt_bytes, alt_recs, alt_probe in alts:
            alt_psnrs = []
            for x, r in zip(last[1:], alt_recs):
                sse = ops.sq_err_sum(crop_to(x, (args.height, args.width)).contiguous(),
                                     crop_to(r, (args.height, args.width)).contiguous()).item()
                alt_psnrs.append(-10 * math.log10(sse / (3 * n_pix)))
            alt_bpp = alt_bytes * 8 / (n_pix * n_p * args.steps * G)
            alt_outs.append({"mode": mode, "value": frames_coded / alt_elapsed, "unit": "frames/s",
                             "ms_per_step": alt_elapsed / args.steps * 1e3, "bpp": alt_bpp,
                             "psnr_db": sum(alt_psnrs) / len(alt_psnrs),
                             "d_bpp_vs_fp32": alt_bpp - bpp,
                             "d_psnr_db_vs_fp32": sum(alt_psnrs) / len(alt_psnrs) - sum(psnrs) / len(psnrs),
                             "roofline": reduced_roofline(mode, alt_probe),
                             "note": "same steps and GOPs as the headline; " + alt_notes.get(mode, "")})

        per = probe.summary()
        dom = max(per.items(), key=lambda kv: kv[1][1])
        all_flops = sum(v[0] for v in per.values())
        all_time = sum(v[1] for v in per.values())
        (k, st, mt, nt), (fl, tm, cnt, nb) = dom
        kname = f"conv_mfma_f32_kernel<{k},{st},{mt},{nt}>"
        traffic, traffic_source = measured_traffic(kname)
        roof = {"bound": "mfma", "kernel": kname,
                "achieved": fl / tm / 1e12, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": fl / tm / 1e12 / PEAK_FP32_MFMA_TFLOPS, "traffic": traffic, "traffic_source": traffic_source,
                "launches": cnt, "avg_launch_ms": tm / cnt * 1e3, "gflop_per_launch": fl / cnt / 1e9,
                "algorithmic_bytes_per_launch": nb / cnt,
                "hbm_bound_kernels": probe.hbm_summary(),
                "all_conv_variants": {"achieved": all_flops / all_time / 1e12,
                                      "frac": all_flops / all_time / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                                      "conv_ms_per_frame": all_time / 2 * 1e3,
                                      "conv_gflop_per_frame": all_flops / 2 / 1e9}}
        if args.precision != "fp32":
            # with bf16 operands the same launches sit under the HBM roof (fp32 activations in memory)
            roof.update({"bound": "hbm", "kernel": kname + f" ({args.precision} operands)", "achieved": nb / tm / 1e9,
                         "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": nb / tm / 1e9 / PEAK_HBM_GBS,
                         "mfma_tflops": fl / tm / 1e12, "traffic": None})
            roof["all_conv_variants"].pop("frac")
        out = {
            "metric": "encode frames/sec + bpp@PSNR, 1080p GOP=12, 1/2/4/8 MI355X",
            "value": fps, "unit": "frames/s", "n_gpus": world, "world_size": world,
            "backend": "rccl" if world > 1 else "none (single process)", "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if args.precision == "fp32" else args.precision + " operands, f32 accumulate", "data": "synthetic",
            "config": {"workload": f"DMC {args.height}x{args.width} (padded {pad64(args.height)}x{pad64(args.width)}) "
                                   f"GOP={args.gop} " + ("encode_inter_with_recon (encoder-side reuse)" if args.reuse
                                                         else "encode_inter+decode_inter") + " per P frame, batch=1 per GPU, "
                                   f"seeded random-init weights (calibrated), {G} independent GOPs in flight per GPU "
                                   "(one host thread + stream each), one such group per step",
                       "gop": args.gop, "gops_per_step_per_gpu": G, "p_frames_per_step_per_gpu": n_p * G},
            "bpp": bpp, "psnr_db": sum(psnrs) / len(psnrs),
            "single_stream_fps": single_fps,
            "roofline": roof,
            "host": host,
        }
        if training is not None:
            out["training"] = training
        if alt_outs:
            out["alt_precision"] = alt_outs[0]
            if len(alt_outs) > 1:
                out["more_precisions"] = alt_outs[1:]
        if args.precision != "fp32":
            # deviation of the reduced-precision mode from the fp32 path on the first frames of a GOP
            def quality(prec):
                net.set_precision(prec)
                b, recs = codec_gop(net, last[:4], clamp)
                ps = []
                for x, r in zip(last[1:4], recs):
                    sse = ops.sq_err_sum(crop_to(x, (args.height, args.width)).contiguous(),
                                         crop_to(r, (args.height, args.width)).contiguous()).item()
                    ps.append(-10 * math.log10(sse / (3 * n_pix)))
                return b * 8 / (n_pix * 3), sum(ps) / len(ps)
            bpp32, psnr32 = quality("fp32")
            bpp16, psnr16 = quality(args.precision)
            out["vs_fp32"] = {"frames": 3, "bpp_fp32": bpp32, "bpp": bpp16, "d_bpp": bpp16 - bpp32,
                              "psnr_fp32_db": psnr32, "psnr_db": psnr16, "d_psnr_db": psnr16 - psnr32}
        if args.cpu_baseline and world == 1:   # the CPU leg runs at N = 1 only
            threads = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
            cb = cpu_baseline(sd, args, threads)
            out["cpu_baseline"] = cb
            out["gpu_over_cpu"] = fps / world / cb["value"]
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
