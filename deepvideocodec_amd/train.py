"""Rate-distortion loss and the training step around DMC.forward (reference: dmc/train.py).

  RateDistortionLoss        dmc/train.py:96-211   (collect_likelihoods_list :74-93)
  configure_optimizers      dmc/train.py:251-282  (main / aux split by the `.quantiles` suffix)
  stage_of_epoch, train_step dmc/train.py:285-346 (six-stage schedule, clip, two optimizer steps)

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
    about `bucket_bytes`, one all-reduce (RCCL on GPUs) per bucket, then unpacked.  Parameters
    without a gradient on this rank contribute zeros (every rank must issue the same collectives).
    Returns the number of buckets; a no-op (0) without an initialised process group."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return 0
    world = dist.get_world_size(group)
    params = [p for p in parameters if p.requires_grad]
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
