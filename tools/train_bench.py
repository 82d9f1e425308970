#!/usr/bin/env python3
"""Training-step benchmark of the DMC path (the reference's training workload, dmc/train.sh +
dmc/train.py:285-346): T frames of HxW random crops, batch B per GPU, joint stage (epoch >= 40),
lambda 2048, Adam 1e-4 / aux 1e-3, clip 1.0.  One process per GPU; under
`python -m torch.distributed.run --nproc-per-node N tools/train_bench.py --gpus N ...` (or plain
`python tools/train_bench.py --gpus N`, which starts that launcher itself) the gradients are
averaged with bucketed RCCL all-reduces.  Prints one JSON line from rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8, help="samples per GPU")
    ap.add_argument("--frames", type=int, default=3)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--epoch", type=int, default=45, help="selects the training stage (dmc/train.py:298-330)")
    ap.add_argument("--recompute", type=int, default=0, help="1: recompute the reconstruction / fusion blocks in backward")
    ap.add_argument("--precision", default="fp32", help="operand mode of the forward / data-gradient convolutions "
                                                        "(fp32, fp16x2, bf16x2, bf16); weight gradients stay fp32")
    ap.add_argument("--gpus", type=int, default=None,
                    help="ranks (one process per GPU); N > 1 without a launcher starts torch.distributed.run itself")
    args = ap.parse_args()
    from deepvideocodec_amd import launch
    world = launch.resolve_world(args.gpus, os.path.abspath(__file__), sys.argv[1:])
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    launch.init_process_group(world, "nccl", dev)

    from deepvideocodec_amd import train as TR
    from deepvideocodec_amd.models import DMC
    from deepvideocodec_amd.testing import seeded_state_dict, synthetic_gop
    net = DMC()
    net.load_state_dict(seeded_state_dict(seed=0))
    net = net.to(dev).train()
    net.recompute = bool(args.recompute)
    net.set_precision(args.precision, train=True)
    crit = TR.RateDistortionLoss(lmbda=2048.0)
    opt, aux_opt = TR.configure_optimizers(net, 1e-4, 1e-3)
    batches = [[f.to(dev) for f in synthetic_gop(args.frames, args.size, args.size, seed=rank * 100 + s, batch=args.batch)]
               for s in range(args.warmup + args.steps)]

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    stats = None
    for s in range(args.warmup):
        stats = TR.train_step(net, crit, batches[s], opt, aux_opt, 1.0, args.epoch)
    barrier()
    t0 = time.perf_counter()
    for s in range(args.warmup, args.warmup + args.steps):
        stats = TR.train_step(net, crit, batches[s], opt, aux_opt, 1.0, args.epoch)
    barrier()
    el = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = t.item()
    if rank == 0:
        # forward FLOPs of one P frame at 256x256: 0.2169 TFLOP; backward = data + weight gradient ~ 2x
        p_frames = args.frames - 1
        fwd_tflop = 0.2169 * (args.size * args.size) / (256.0 * 256.0) * p_frames * args.batch
        print(json.dumps({
            "metric": "training samples/s (T=%d, %dx%d crops, joint stage)" % (args.frames, args.size, args.size),
            "value": args.batch * world * args.steps / el, "unit": "samples/s", "n_gpus": world, "world_size": world,
            "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": el / args.steps * 1e3, "batch_per_gpu": args.batch,
            "recompute": bool(args.recompute),
            "approx_tflops_per_gpu": 3 * fwd_tflop / (el / args.steps), "last_step": stats,
            "max_mem_gb": torch.cuda.max_memory_allocated() / 2 ** 30,
            "dtype": "f32" if args.precision == "fp32" else "f32 (%s operands in the forward / data-gradient convolutions)" % args.precision,
            "data": "synthetic"}), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
