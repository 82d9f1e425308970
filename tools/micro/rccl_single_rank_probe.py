#!/usr/bin/env python3
"""RCCL on one rank: process-group initialisation on the device, barrier, the scalar all-reduces of the benchmark's
timing, and a 32 MiB fp32 bucket as the training step's gradient exchange sends them.  World size 1 makes every
collective the identity, but the library is loaded, the communicator is built and the calls are the ones the
multi-GPU paths issue."""
import os
import sys

import torch


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    import torch.distributed as dist
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", device_id=dev)
    dist.barrier()
    t = torch.tensor([1.5, 2.0, 3.0], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    m = torch.tensor([0.25], device=dev, dtype=torch.float64)
    dist.all_reduce(m, op=dist.ReduceOp.MAX)
    mask = torch.ones(430, device=dev, dtype=torch.int32)
    dist.all_reduce(mask, op=dist.ReduceOp.MAX)
    bucket = torch.randn((32 << 20) // 4, device=dev)
    ref = bucket.clone()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    dist.all_reduce(bucket)
    e0.record()
    for _ in range(5):
        dist.all_reduce(bucket)
    e1.record()
    torch.cuda.synchronize()
    print("sum", t.tolist(), "max", m.tolist(), "bucket unchanged:", bool(torch.equal(bucket, ref)),
          f"{e0.elapsed_time(e1) / 5 * 1e3:.1f} us per 32 MiB all-reduce on one rank", flush=True)
    dist.barrier()
    dist.destroy_process_group()
    print("rccl single-rank probe ok", flush=True)


if __name__ == "__main__":
    main()
