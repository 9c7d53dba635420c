"""Multi-GPU plumbing of the hot path: frames shard across ranks, counts are all-gathered.

Frames are independent units: every rank owns one camera stream (or a
contiguous slice of a shared batch), its own vslam context and its own keypoint lists.  The
only exchange step is the all-gather of the per-rank {harris, dog} keypoint counts -- 16 bytes
per rank -- from which every rank derives the global totals and its output offset.  The
backend is whatever torch.distributed was initialised with: "nccl" (= RCCL over xGMI) on the
GPUs, "gloo" in the CPU tests.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(n_total: int, world: int, rank: int) -> range:
    """Contiguous slice of a shared batch of n_total frames owned by `rank` (sizes differ by <= 1)."""
    if not 0 <= rank < world:
        raise ValueError("rank out of range")
    base, rem = divmod(n_total, world)
    start = rank * base + min(rank, rem)
    return range(start, start + base + (1 if rank < rem else 0))


def gather_counts(local_counts: torch.Tensor, counts_all: torch.Tensor | None = None) -> torch.Tensor:
    """All-gather a [2] int64 tensor {harris_count, dog_count}; returns [world, 2].

    Works without an initialised process group (world = 1).  `counts_all` may be a
    pre-allocated [world, 2] buffer (bench.py reuses one to keep the step allocation-free).
    """
    if local_counts.dtype != torch.int64 or local_counts.numel() != 2:
        raise ValueError("local_counts must be an int64 tensor of two elements")
    world = dist.get_world_size() if dist.is_initialized() else 1
    if counts_all is None:
        counts_all = torch.empty((world, 2), dtype=torch.int64, device=local_counts.device)
    if world == 1:
        counts_all[0] = local_counts
    else:
        dist.all_gather_into_tensor(counts_all.view(-1), local_counts.contiguous())
    return counts_all


def global_offsets(counts_all: torch.Tensor, rank: int):
    """(offset of this rank's first keypoint in the global list, global total) per list kind."""
    csum = torch.cumsum(counts_all, 0)
    total = csum[-1]
    offset = csum[rank] - counts_all[rank]
    return offset, total
