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


RECORD_BYTES = 32          # one per-cell record: mse f32 | mae f32 | cons_score f64 | mod_score f64 | cons_pred i8 | mod_pred i8 | pad
_REC_LAYOUT = (("mse", 0, 4), ("mae", 4, 4), ("cons_score", 8, 8), ("mod_score", 16, 8), ("cons_pred", 24, 1), ("mod_pred", 25, 1))


def pack_records(local: Dict, rows: Optional[int] = None):
    """The six result fields of a shard as ONE uint8 tensor [rows, 32] (rows >= shard size: zero-padded tail), so
    that the gather is a single collective of 26 useful bytes per cell instead of six padded ones."""
    import torch
    n = local["mse"].shape[0]
    rows = n if rows is None else rows
    rec = torch.zeros((rows, RECORD_BYTES), dtype=torch.uint8, device=local["mse"].device)
    for name, off, width in _REC_LAYOUT:
        t = local[name]
        if t.shape[0] != n:
            raise ValueError(f"field {name} has {t.shape[0]} rows, mse has {n}")
        rec[:n, off:off + width] = t.contiguous().view(torch.uint8).view(n, width)
    return rec


def unpack_records(rec, n: Optional[int] = None) -> Dict:
    import torch
    n = rec.shape[0] if n is None else n
    dt = {"float32": torch.float32, "float64": torch.float64, "int8": torch.int8}
    out = {}
    for (name, off, width), (_, dtype) in zip(_REC_LAYOUT, RESULT_FIELDS):
        out[name] = rec[:n, off:off + width].contiguous().view(dt[dtype]).view(n)
    return out


def gather_results(local: Dict, n_total: int, group=None, dst: Optional[int] = None) -> Optional[Dict]:
    """All ranks call with their shard's result tensors (torch, same device type on every rank).  Returns, on every
    rank (dst=None) or only on `dst`, the global arrays in cell order.  ONE collective per call: the six fields are
    packed into 32-byte per-cell records (26 useful bytes; 22.5 MB per rank at 10 M cells over 8 GPUs), shards padded
    to the largest one, gathered to `dst` (or all-gathered), unpacked in shard order."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    counts = shard_counts(n_total, world)
    maxc = max(counts) if counts else 0
    if local["mse"].shape[0] != counts[rank]:
        raise ValueError(f"rank {rank}: shard has {counts[rank]} cells, results have {local['mse'].shape[0]}")
    rec = pack_records(local, maxc)
    if dst is None:
        flat = torch.empty((world * maxc, RECORD_BYTES), dtype=torch.uint8, device=rec.device)
        dist.all_gather_into_tensor(flat, rec, group=group)
        allrec = flat.view(world, maxc, RECORD_BYTES)
    else:
        allrec = torch.empty((world, maxc, RECORD_BYTES), dtype=torch.uint8, device=rec.device) if rank == dst else None
        dist.gather(rec, list(allrec.unbind(0)) if rank == dst else None, dst=dst, group=group)
        if rank != dst:
            return None
    if all(c == maxc for c in counts):
        return unpack_records(allrec.view(world * maxc, RECORD_BYTES))
    return unpack_records(torch.cat([allrec[r, :c] for r, c in enumerate(counts)]))


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
