#!/usr/bin/env python3
"""Training driver around deepvideocodec_amd.train.fit -- the role of dmc/train.py:main (:540-655) with
its flags (`-d`, `-e`, `-lr`, `--lambda`, `--batch-size`, `--patch-size`, `--clip_max_norm`, `--save`,
`--checkpoint`, `--seed`).  One process per GPU: launch with
`python -m torch.distributed.run --nproc-per-node N tools/train.py ...`; each rank reads every N-th
sample and gradients are averaged with RCCL all-reduces (replaces nn.DataParallel, :230-237).

Datasets: `-d synthetic[:SAMPLES]` (moving smooth fields, generated on the fly), or a directory in the
layout the reference's VideoFolder reads (`<root>/{train,test}.list` naming sequence directories under
`<root>/sequences/`, frames `im1.png ...`), decoded with Pillow.  There is no intra codec in this
package: from epoch 50 on pass one with `fit(net_intra=...)` from your own script.
"""
import argparse
import logging
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class SyntheticSequences(torch.utils.data.Dataset):
    def __init__(self, samples: int, frames: int, size, seed: int):
        self.samples, self.frames, self.size, self.seed = samples, frames, size, seed

    def __len__(self):
        return self.samples

    def __getitem__(self, i):
        from deepvideocodec_amd.testing import synthetic_gop
        return [f[0] for f in synthetic_gop(self.frames, self.size[0], self.size[1], seed=self.seed + i)]


class SequenceFolder(torch.utils.data.Dataset):
    """`<root>/<split>.list` + `<root>/sequences/<entry>/im*.png`; random (train) or centre (test) crop,
    the first `frames` frames of each sequence."""

    def __init__(self, root: str, split: str, frames: int, size, seed: int):
        with open(os.path.join(root, f"{split}.list")) as f:
            self.dirs = [os.path.join(root, "sequences", ln.strip()) for ln in f if ln.strip()]
        self.frames, self.size, self.train, self.rng = frames, size, split == "train", random.Random(seed)

    def __len__(self):
        return len(self.dirs)

    def __getitem__(self, i):
        import numpy as np
        from PIL import Image
        names = sorted(n for n in os.listdir(self.dirs[i]) if n.lower().endswith(".png"))[: self.frames]
        if len(names) < self.frames:
            raise RuntimeError(f"{self.dirs[i]}: {len(names)} frames, need {self.frames}")
        imgs = [np.asarray(Image.open(os.path.join(self.dirs[i], n)).convert("RGB")) for n in names]
        h, w = imgs[0].shape[:2]
        ph, pw = self.size
        if h < ph or w < pw:
            raise RuntimeError(f"{self.dirs[i]}: {h}x{w} smaller than the {ph}x{pw} patch")
        top, left = ((self.rng.randint(0, h - ph), self.rng.randint(0, w - pw)) if self.train
                     else ((h - ph) // 2, (w - pw) // 2))
        return [torch.from_numpy(im[top:top + ph, left:left + pw].copy()).permute(2, 0, 1).float().div_(255.0)
                for im in imgs]


def make_dataset(spec: str, split: str, frames: int, size, seed: int):
    if spec.startswith("synthetic"):
        n = int(spec.split(":")[1]) if ":" in spec else 64
        return SyntheticSequences(n if split == "train" else max(1, n // 8), frames, size,
                                  seed + (0 if split == "train" else 1 << 20))
    return SequenceFolder(spec, split, frames, size, seed)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("-d", "--dataset", required=True)
    ap.add_argument("-e", "--epochs", type=int, default=100)
    ap.add_argument("-lr", "--learning-rate", type=float, default=1e-4)
    ap.add_argument("--aux-learning-rate", type=float, default=1e-3)
    ap.add_argument("-n", "--num-workers", type=int, default=4)
    ap.add_argument("--lambda", dest="lmbda", type=float, default=2048.0)
    ap.add_argument("--batch-size", type=int, default=4, help="per GPU")
    ap.add_argument("--test-batch-size", type=int, default=4)
    ap.add_argument("--frames", type=int, default=3)
    ap.add_argument("--patch-size", type=int, nargs=2, default=(256, 256))
    ap.add_argument("--clip_max_norm", type=float, default=1.0)
    ap.add_argument("--save", default=None, help="directory for checkpoint.pth.tar / checkpoint_best_loss.pth.tar")
    ap.add_argument("--checkpoint", default=None)
    ap.add_argument("--seed", type=int, default=None)
    args = ap.parse_args(argv)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
    if args.seed is not None:
        torch.manual_seed(args.seed)
        random.seed(args.seed)
    logging.basicConfig(level=logging.INFO if rank == 0 else logging.WARNING, format="%(asctime)s %(message)s")

    from deepvideocodec_amd import train as TR
    from deepvideocodec_amd.models import DMC
    size = tuple(args.patch_size)
    if size[0] % 64 or size[1] % 64:
        ap.error("--patch-size must be multiples of 64")
    seed = args.seed or 0
    train_ds = make_dataset(args.dataset, "train", args.frames, size, seed)
    test_ds = make_dataset(args.dataset, "test", args.frames, size, seed)
    sampler = lambda ds, shuffle: (torch.utils.data.distributed.DistributedSampler(ds, world, rank, shuffle=shuffle)
                                   if world > 1 else None)
    train_sampler = sampler(train_ds, True)
    train_loader = torch.utils.data.DataLoader(train_ds, batch_size=args.batch_size, num_workers=args.num_workers,
                                               shuffle=train_sampler is None, sampler=train_sampler, pin_memory=True,
                                               drop_last=True)
    test_loader = torch.utils.data.DataLoader(test_ds, batch_size=args.test_batch_size, num_workers=args.num_workers,
                                              shuffle=False, sampler=sampler(test_ds, False), pin_memory=True)
    net = DMC().to(dev)           # Xavier-normal initial weights, like the reference's constructor
    if world > 1:                 # same initial weights on every rank
        import torch.distributed as dist
        for t in list(net.parameters()) + list(net.buffers()):
            dist.broadcast(t.data, 0)
    res = TR.fit(net, train_loader, test_loader, args.epochs, lmbda=args.lmbda, learning_rate=args.learning_rate,
                 aux_learning_rate=args.aux_learning_rate, clip_max_norm=args.clip_max_norm, save_dir=args.save,
                 checkpoint=args.checkpoint, log=logging.info)
    if rank == 0:
        print({k: (round(v, 6) if isinstance(v, float) else v) for k, v in res.items()})
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
