"""GOP sharding over ranks: the only multi-GPU structure of the coding path.

A GOP starts from an intra frame with an empty decoded-picture buffer (reference:
dmc/test.py:162-172), so GOPs share no state; inside a GOP frames are sequential
(dmc/test.py:186-195).  Ranks therefore take whole GOPs round-robin and never exchange data on
the coding path; one all-reduce of three scalars merges the statistics at the end.
"""
from typing import List, Sequence, Tuple

import torch


def gop_ranges(num_frames: int, gop: int) -> List[Tuple[int, int]]:
    """[start, end) frame ranges of the GOPs of a sequence (the last one may be short)."""
    if gop < 1:
        raise ValueError("gop must be >= 1")
    return [(s, min(s + gop, num_frames)) for s in range(0, num_frames, gop)]


def gops_of_rank(num_gops: int, rank: int, world: int) -> List[int]:
    """GOP g is coded by rank g mod world."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world of {world}")
    return list(range(rank, num_gops, world))


def reduce_stats(total_bytes: float, sum_psnr: float, frames: int, device=None) -> Tuple[float, float, int]:
    """Sum (bytes, sum of per-frame PSNR, frame count) over all ranks; identity without a process group."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(total_bytes), float(sum_psnr), int(frames)
    t = torch.tensor([float(total_bytes), float(sum_psnr), float(frames)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t[0].item(), t[1].item(), int(round(t[2].item()))


def order_segments(segments: Sequence[Tuple[int, bytes]]) -> bytes:
    """Concatenate per-GOP bitstream segments (gop index, bytes) in GOP order."""
    return b"".join(b for _, b in sorted(segments, key=lambda s: s[0]))
