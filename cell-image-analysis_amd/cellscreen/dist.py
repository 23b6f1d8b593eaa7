"""Data-parallel screening: one process per GPU, cells sharded by contiguous index ranges,
no data-path collective until the single gather of per-cell results at the end
(SURVEY.md section 8e).  torch.distributed is plumbing here: backend "nccl" (= RCCL over
xGMI) for CUDA tensors, "gloo" for the CPU tests."""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np

RESULT_FIELDS = (("mse", "float32"), ("mae", "float32"), ("cons_score", "float64"),
                 ("mod_score", "float64"), ("cons_pred", "int8"), ("mod_pred", "int8"))


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Rank r owns [r*n//W, (r+1)*n//W): contiguous, covers [0,n) exactly, sizes differ by <= 1.
    cell_id bookkeeping (improved_detection.py:217-227) is then a pure offset."""
    if world <= 0 or not (0 <= rank < world) or n < 0:
        raise ValueError("bad shard arguments")
    return (rank * n) // world, ((rank + 1) * n) // world


def shard_counts(n: int, world: int):
    return [shard_range(n, r, world)[1] - shard_range(n, r, world)[0] for r in range(world)]


def gather_results(local: Dict, n_total: int, group=None, dst: Optional[int] = None) -> Optional[Dict]:
    """All ranks call with their shard's result tensors (torch, same device type on every
    rank).  Returns, on every rank (dst=None) or only on `dst`, the global arrays in cell order.
    One all_gather per field on padded shards; the 18 B/cell payload is tiny next to the compute."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    counts = shard_counts(n_total, world)
    maxc = max(counts) if counts else 0
    out = {}
    for name, _dt in RESULT_FIELDS:
        t = local[name]
        if t.shape[0] != counts[rank]:
            raise ValueError(f"rank {rank}: field {name} has {t.shape[0]} rows, shard has {counts[rank]}")
        pad = torch.zeros((maxc,), dtype=t.dtype, device=t.device)
        pad[: t.shape[0]] = t
        bufs = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(bufs, pad, group=group)
        if dst is None or rank == dst:
            out[name] = torch.cat([b[:c] for b, c in zip(bufs, counts)])
    return out if (dst is None or rank == dst) else None


def to_torch(local_np: Dict, device="cpu") -> Dict:
    import torch
    return {k: torch.as_tensor(np.ascontiguousarray(v), device=device) for k, v in local_np.items()}


def allreduce_mean_(t, group=None):
    """In-place mean over ranks of a flat gradient tensor: each rank's loss is the mean over its
    local batch, so the gradient of the global-batch mean is the average of the per-rank
    gradients (one RCCL all-reduce of 337 KB per step; latency-bound)."""
    import torch.distributed as dist
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    t /= dist.get_world_size(group)
    return t
