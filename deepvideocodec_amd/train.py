"""Rate-distortion loss and the training step around DMC.forward (reference: dmc/train.py).

  RateDistortionLoss        dmc/train.py:96-211   (collect_likelihoods_list :74-93)
  configure_optimizers      dmc/train.py:251-282  (main / aux split by the `.quantiles` suffix)
  stage_of_epoch, train_step dmc/train.py:285-346 (six-stage schedule, clip, two optimizer steps)
  train_one_epoch, test_epoch, save_checkpoint, load_checkpoint, fit
                            dmc/train.py:285-403,597-655 (epoch loop, checkpoint dict, MultiStepLR)

The reductions of the loss (sum of log-likelihoods per sample, squared error per sample) are the
HIP reduction kernels with their adjoints; what is left to torch is arithmetic on (N,)-vectors.
Data parallelism replaces the reference's nn.DataParallel (dmc/train.py:230-237,598-600) by one
process per GPU: `all_reduce_gradients` averages the gradients over ranks in a few large flat
buckets (RCCL all-reduce over xGMI; 16.9 M parameters = 67.5 MB in fp32).
"""
import math
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from . import autograd as A

Tensor = torch.Tensor


def _bits_per_sample(lik: Tensor) -> Tensor:
    """(N,) sum of -log2(likelihood)."""
    if lik.is_cuda:
        return A.likelihood_bits(lik)
    return -(torch.log(lik).sum(dim=(1, 2, 3)) / math.log(2))


def collect_likelihoods_list(likelihoods_list, num_pixels: int):
    """dmc/train.py:74-93: per-sample bpp summed over frames / latents, plus the breakdown."""
    info: Dict[str, Tensor] = {}
    bpp_loss = 0
    for i, frame_likelihoods in enumerate(likelihoods_list):
        frame_bpp = 0
        for label, likelihoods in frame_likelihoods.items():
            label_bpp = 0
            for field, v in likelihoods.items():
                bpp = _bits_per_sample(v) / num_pixels
                bpp_loss = bpp_loss + bpp
                frame_bpp = frame_bpp + bpp
                label_bpp = label_bpp + bpp
                info[f"bpp_loss.{label}"] = info.get(f"bpp_loss.{label}", 0) + bpp.sum()
                info[f"bpp_loss.{label}.{i}.{field}"] = bpp.sum()
            info[f"bpp_loss.{label}.{i}"] = label_bpp.sum()
        info[f"bpp_loss.{i}"] = frame_bpp.sum()
    return bpp_loss, info


class RateDistortionLoss(nn.Module):
    """loss = mean_n(lambda * mean_frames(mse_n)) + mean_n(bpp_n) (dmc/train.py:96-211)."""

    def __init__(self, lmbda: float = 1e-2, return_details: bool = False, bitdepth: int = 8):
        super().__init__()
        self.lmbda = lmbda
        self.return_details = bool(return_details)

    @staticmethod
    def _distortion(x_hat: Tensor, target: Tensor) -> Tensor:
        """(N,) mean squared error over (C, H, W) -- the per-component means of the reference
        (dmc/train.py:114-143) averaged over the components."""
        if len(x_hat) != len(target):
            raise RuntimeError(f"len(x)={len(x_hat)} != len(target)={len(target)})")
        if x_hat.size(1) != target.size(1):
            raise RuntimeError("number of channels mismatches while computing distortion")
        n, c, h, w = x_hat.shape
        if x_hat.is_cuda:
            return A.sq_err_sum(x_hat, target) / float(c * h * w)
        return ((x_hat - target) ** 2).mean(dim=(1, 2, 3))

    def forward(self, output, target):
        if not isinstance(target, type(output["x_hat"])) or len(output["x_hat"]) != len(target):
            raise ValueError("target must be a list of frames matching output['x_hat']")
        for lst in (target, output["x_hat"]):
            if not isinstance(lst, (tuple, list)) or len(lst) < 1 or any(
                    not (isinstance(t, torch.Tensor) and t.dim() == 4) for t in lst):
                raise ValueError("Expected a list of 4D torch.Tensor (or tuples of) as input")
        _, _, H, W = target[0].size()
        num_frames = len(target)
        num_pixels = H * W * num_frames
        out = {}
        distortions = []
        for i, (x_hat, x) in enumerate(zip(output["x_hat"], target)):
            d = self._distortion(x_hat, x)
            distortions.append(d)
            if self.return_details:
                out[f"frame{i}.mse_loss"] = d
        out["mse_loss"] = torch.stack(distortions).mean()
        distortions = sum(distortions) / num_frames
        likelihoods_list = output.pop("likelihoods")
        bpp_loss, info = collect_likelihoods_list(likelihoods_list, num_pixels)
        if self.return_details:
            out.update(info)
        bpp_loss = bpp_loss.mean()
        out["loss"] = (self.lmbda * distortions).mean() + bpp_loss
        out["distortion"] = distortions.mean()
        out["bpp_loss"] = bpp_loss
        return out


def configure_optimizers(net: nn.Module, learning_rate: float = 1e-4, aux_learning_rate: float = 1e-3):
    """dmc/train.py:251-282."""
    params = dict(net.named_parameters())
    main = sorted(n for n, p in params.items() if not n.endswith(".quantiles") and p.requires_grad)
    aux = sorted(n for n, p in params.items() if n.endswith(".quantiles") and p.requires_grad)
    assert not set(main) & set(aux) and len(main) + len(aux) == len(params)
    optimizer = torch.optim.Adam((params[n] for n in main), lr=learning_rate)
    aux_optimizer = torch.optim.Adam((params[n] for n in aux), lr=aux_learning_rate)
    return optimizer, aux_optimizer


def stage_of_epoch(epoch: int) -> Tuple[bool, bool, str]:
    """(motion_pretrain, frame_pretrain, which loss to back-propagate) for an epoch
    (dmc/train.py:298-330; from epoch 50 on the intra frame is replaced by its reconstruction)."""
    if epoch < 10:
        return True, False, "mse_loss"
    if epoch < 20:
        return True, False, "loss"
    if epoch < 30:
        return False, True, "mse_loss"
    if epoch < 40:
        return False, True, "loss"
    return False, False, "loss"


def all_reduce_gradients(parameters: Iterable[Tensor], bucket_bytes: int = 32 << 20, group=None) -> int:
    """Average .grad over the ranks of the process group: gradients are packed into flat buckets of
    about `bucket_bytes`, one all-reduce (RCCL on GPUs) per bucket, then unpacked.

    Which parameters take part is decided globally: one small all-reduce (MAX) of a presence mask tells
    every rank which parameters received a gradient on at least one rank.  Those are reduced (a rank
    without a gradient for one of them contributes zeros); a parameter no rank has a gradient for keeps
    `.grad = None`, exactly as in a single-process run -- so the optimizer skips it (no Adam state, no
    step count) in the stages that train only part of the model (dmc/train.py:298-330).
    Returns the number of buckets; a no-op (0) without an initialised process group."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return 0
    world = dist.get_world_size(group)
    params = [p for p in parameters if p.requires_grad]
    if not params:
        return 0
    have = torch.tensor([0 if p.grad is None else 1 for p in params], dtype=torch.int32, device=params[0].device)
    dist.all_reduce(have, op=dist.ReduceOp.MAX, group=group)
    params = [p for p, h in zip(params, have.tolist()) if h]
    buckets: List[List[Tensor]] = [[]]
    size = 0
    for p in params:
        nb = p.numel() * p.element_size()
        if buckets[-1] and size + nb > bucket_bytes:
            buckets.append([])
            size = 0
        buckets[-1].append(p)
        size += nb
    for bucket in buckets:
        if not bucket:
            continue
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat /= world
        off = 0
        for p in bucket:
            n = p.numel()
            g = flat[off:off + n].view_as(p)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
            off += n
    return len([b for b in buckets if b])


def train_step(net, criterion: RateDistortionLoss, frames: Sequence[Tensor], optimizer, aux_optimizer,
               clip_max_norm: float = 1.0, epoch: int = 40, net_intra=None) -> Dict[str, float]:
    """One iteration of dmc/train.py:292-337 on an already-on-device batch of frames."""
    frames = list(frames)
    motion_pretrain, frame_pretrain, which = stage_of_epoch(epoch)
    optimizer.zero_grad(set_to_none=True)
    aux_optimizer.zero_grad(set_to_none=True)
    if epoch >= 50 and net_intra is not None:
        with torch.no_grad():
            frames[0] = net_intra(frames[0])["x_hat"]
    out_net = net(frames, motion_pretrain=motion_pretrain, frame_pretrain=frame_pretrain)
    out = criterion(out_net, frames[1:])
    out[which].backward()
    main_params = [p for n, p in net.named_parameters() if not n.endswith(".quantiles")]
    all_reduce_gradients(main_params)
    if clip_max_norm > 0:
        torch.nn.utils.clip_grad_norm_(net.parameters(), clip_max_norm)
    optimizer.step()
    aux_loss = sum(net.aux_loss())
    aux_loss.backward()
    all_reduce_gradients([p for n, p in net.named_parameters() if n.endswith(".quantiles")])
    aux_optimizer.step()
    return {"loss": out["loss"].item(), "mse_loss": out["mse_loss"].item(), "bpp_loss": out["bpp_loss"].item(),
            "aux_loss": aux_loss.item()}


class AverageMeter:
    """Running average (dmc/train.py:214-227)."""

    def __init__(self):
        self.val = self.avg = self.sum = 0.0
        self.count = 0

    def update(self, val, n: int = 1):
        self.val = float(val)
        self.sum += float(val) * n
        self.count += n
        self.avg = self.sum / self.count


def train_one_epoch(epoch: int, net_intra, net, criterion: RateDistortionLoss, dataloader, optimizer, aux_optimizer,
                    clip_max_norm: float = 1.0, log=None, log_every: int = 100) -> Dict[str, float]:
    """dmc/train.py:285-346: one pass over `dataloader` (each batch a list of T frame tensors) in training
    mode; the stage (which sub-networks run, which loss is back-propagated) follows from `epoch`.
    Returns the epoch's average losses."""
    net.train()
    device = next(net.parameters()).device
    meters = {k: AverageMeter() for k in ("loss", "mse_loss", "bpp_loss", "aux_loss")}
    sampler = getattr(dataloader, "sampler", None)
    if hasattr(sampler, "set_epoch"):        # per-epoch reshuffle of a DistributedSampler
        sampler.set_epoch(epoch)
    for i, batch in enumerate(dataloader):
        frames = [f.to(device, non_blocking=True) for f in batch]
        stats = train_step(net, criterion, frames, optimizer, aux_optimizer, clip_max_norm, epoch, net_intra)
        for k, m in meters.items():
            m.update(stats[k])
        if log is not None and i % log_every == 0:
            log(f"Train epoch {epoch}: [{i * len(frames[0])}/{_dataset_len(dataloader)}] "
                f"Loss: {stats['loss']:.3f} | MSE loss: {stats['mse_loss']:.5f} | "
                f"Bpp loss: {stats['bpp_loss']:.4f} | Aux loss: {stats['aux_loss']:.2f}")
    return {k: m.avg for k, m in meters.items()}


def _dataset_len(dataloader) -> int:
    ds = getattr(dataloader, "dataset", None)
    try:
        return len(ds) if ds is not None else len(dataloader)
    except TypeError:
        return -1


def test_epoch(epoch: int, dataloader, net_intra, net, criterion: RateDistortionLoss, log=None) -> float:
    """dmc/train.py:349-398: evaluation-mode pass with the epoch's stage flags; returns the average loss,
    reduced over the ranks of the process group when there is one (every rank then sees the same value,
    so `is_best` and the learning-rate schedule agree across ranks)."""
    net.eval()
    device = next(net.parameters()).device
    meters = {k: AverageMeter() for k in ("loss", "mse_loss", "bpp_loss", "aux_loss")}
    motion_pretrain, frame_pretrain, _ = stage_of_epoch(epoch)
    with torch.no_grad():
        for batch in dataloader:
            frames = [f.to(device, non_blocking=True) for f in batch]
            if epoch >= 50 and net_intra is not None:
                frames[0] = net_intra(frames[0])["x_hat"]
            out = criterion(net(frames, motion_pretrain=motion_pretrain, frame_pretrain=frame_pretrain), frames[1:])
            meters["aux_loss"].update(float(sum(net.aux_loss())))
            for k in ("loss", "mse_loss", "bpp_loss"):
                meters[k].update(float(out[k]))
    sums = torch.tensor([meters[k].sum for k in ("loss", "mse_loss", "bpp_loss", "aux_loss")] + [meters["loss"].count],
                        dtype=torch.float64)
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        sums = sums.to(device) if dist.get_backend() == "nccl" else sums
        dist.all_reduce(sums)
        sums = sums.cpu()
    avg = (sums[:4] / sums[4].clamp(min=1)).tolist()
    if log is not None:
        log(f"Test epoch {epoch}: Average losses: Loss: {avg[0]:.3f} | MSE loss: {avg[1]:.5f} | "
            f"Bpp loss: {avg[2]:.4f} | Aux loss: {avg[3]:.2f}")
    return avg[0]


def checkpoint_state(epoch: int, net, loss: float, optimizer, aux_optimizer, lr_scheduler) -> dict:
    """The dict dmc/train.py:644-652 saves (same keys)."""
    return {"epoch": epoch, "state_dict": net.state_dict(), "loss": loss, "optimizer": optimizer.state_dict(),
            "aux_optimizer": aux_optimizer.state_dict(), "lr_scheduler": lr_scheduler.state_dict()}


def save_checkpoint(state: dict, is_best: bool, base_dir: str, filename: str = "checkpoint.pth.tar") -> str:
    """dmc/train.py:400-403 (the path is joined properly rather than concatenated)."""
    import os
    import shutil
    os.makedirs(base_dir, exist_ok=True)
    path = os.path.join(base_dir, filename)
    tmp = path + ".tmp"
    torch.save(state, tmp)
    os.replace(tmp, path)           # a crash mid-write leaves the previous checkpoint intact
    if is_best:
        shutil.copyfile(path, os.path.join(base_dir, "checkpoint_best_loss.pth.tar"))
    return path


def load_checkpoint(path: str, net, optimizer=None, aux_optimizer=None, lr_scheduler=None, map_location="cpu") -> int:
    """Resume (dmc/train.py:609-616): loads the model and, when given, the optimizer / scheduler states
    (the reference leaves those three commented out); returns the epoch to continue from.  Tensors only
    (`weights_only=True`): nothing in the file is executed."""
    ck = torch.load(path, map_location=map_location, weights_only=True)
    net.load_state_dict(ck["state_dict"])
    for obj, key in ((optimizer, "optimizer"), (aux_optimizer, "aux_optimizer"), (lr_scheduler, "lr_scheduler")):
        if obj is not None and key in ck:
            obj.load_state_dict(ck[key])
    return int(ck["epoch"]) + 1


def fit(net, train_loader, test_loader, epochs: int, lmbda: float = 2048.0, learning_rate: float = 1e-4,
        aux_learning_rate: float = 1e-3, clip_max_norm: float = 1.0, net_intra=None, save_dir: Optional[str] = None,
        checkpoint: Optional[str] = None, log=None) -> Dict[str, float]:
    """The epoch loop of dmc/train.py:597-655: dual Adam, MultiStepLR(milestones=[300], gamma=0.1),
    train pass, test pass, best-loss tracking, checkpoint per epoch (written by rank 0 only)."""
    import torch.distributed as dist
    optimizer, aux_optimizer = configure_optimizers(net, learning_rate, aux_learning_rate)
    lr_scheduler = torch.optim.lr_scheduler.MultiStepLR(optimizer, milestones=[300], gamma=0.1)
    criterion = RateDistortionLoss(lmbda=lmbda, return_details=True)
    first = load_checkpoint(checkpoint, net) if checkpoint else 0
    rank0 = not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0
    best, last = float("inf"), {}
    for epoch in range(first, epochs):
        if log is not None:
            log(f"======Current epoch {epoch} ======  Learning rate: {optimizer.param_groups[0]['lr']}")
        last = train_one_epoch(epoch, net_intra, net, criterion, train_loader, optimizer, aux_optimizer,
                               clip_max_norm, log)
        loss = test_epoch(epoch, test_loader, net_intra, net, criterion, log)
        lr_scheduler.step()
        is_best, best = loss < best, min(loss, best)
        last.update({"test_loss": loss, "best_loss": best, "epoch": epoch})
        if save_dir and rank0:
            save_checkpoint(checkpoint_state(epoch, net, loss, optimizer, aux_optimizer, lr_scheduler), is_best, save_dir)
    return last
