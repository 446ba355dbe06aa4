"""Frame-pair sharding across the GPUs of one node: one process per GPU, contiguous
blocks of the selected frame pairs, no data-path collective, ONE reduce(sum, f64) of the zero-padded
per-frame score vector to rank 0 (RCCL over xGMI on the GPUs, gloo in the CPU tests).

Each score is produced by exactly one rank and added to zeros, so rank 0 holds bit-identical values to a
single-GPU run, in frame order.
"""
from typing import Sequence, Tuple

import numpy as np


def shard_range(n_frames: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of rank `rank`: ceil(n/world) frames per rank, the tail ranks may be short/empty."""
    per = -(-n_frames // world)
    lo = min(n_frames, rank * per)
    return lo, min(n_frames, lo + per)


def reduce_scores(local_scores: Sequence[float], lo: int, n_frames: int, n_metrics: int = 1, dist=None, device="cpu"):
    """local_scores: (hi-lo, n_metrics) scores of this rank's block.  Returns the (n_frames, n_metrics) array on
    rank 0 (None elsewhere).  `dist` = torch.distributed (initialised) or None for a single process."""
    local = np.asarray(local_scores, np.float64).reshape(-1, n_metrics)
    if dist is None:
        assert lo == 0 and local.shape[0] == n_frames
        return local
    import torch
    t = torch.zeros(n_frames * n_metrics, dtype=torch.float64, device=device)
    if local.size:
        t[lo * n_metrics:(lo + local.shape[0]) * n_metrics] = torch.from_numpy(local.ravel()).to(device)
    dist.reduce(t, dst=0, op=dist.ReduceOp.SUM)
    if dist.get_rank() == 0:
        return t.cpu().numpy().reshape(n_frames, n_metrics)
    return None


def min_max_over_ranks(x: float, dist=None, device="cpu"):
    """[min, max] over the ranks of a per-rank figure (one all_reduce(MAX) of (x, -x)): bench.py's `per_rank`, which makes a
    straggler visible on the one line the driver keeps.  dist = torch.distributed (initialised) or None."""
    if dist is None:
        return [float(x), float(x)]
    import torch
    t = torch.tensor([x, -x], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [-float(t[1].item()), float(t[0].item())]
