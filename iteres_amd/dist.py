"""Multi-GPU plumbing for the record-sharded path (one process per GPU, torch.distributed: "nccl" = RCCL on ROCm,
"gloo" on CPU for the tests). The data path has no collective: each rank runs the engine on its own part of the
stream; the single exchange is a sum-all-reduce of the compact partial at end of stream."""
from __future__ import annotations

import numpy as np


def shard_bounds(n: int, rank: int, world: int, align: int = 1) -> tuple[int, int]:
    """Contiguous, disjoint, exhaustive split of n records over `world` ranks (boundaries multiples of `align`)."""
    per = -(-n // world)
    per = -(-per // align) * align
    lo = min(rank * per, n)
    return lo, min(lo + per, n)


def as_signed_view(a: np.ndarray) -> np.ndarray:
    """u64/u32 accumulators travel as int64/int32: two's-complement addition is the same sum mod 2^64 / 2^32,
    which is what the reference's unsigned counters do."""
    return a.view({np.dtype("uint64"): np.int64, np.dtype("uint32"): np.int32}[a.dtype])


def allreduce_sum_(tensors, dist=None):
    """In-place SUM all-reduce of every tensor in `tensors` (no-op for a single process)."""
    if dist is None:
        import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return tensors
    for t in tensors:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return tensors


def reduce_sum_(tensors, dist=None, dst=0):
    """In-place SUM reduce onto rank `dst` — the rank that writes the files needs the merged partial, nobody else does,
    and a ring reduce moves half the bytes of an all-reduce. Other ranks' tensors are left undefined."""
    if dist is None:
        import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return tensors
    on_gloo_device = dist.get_backend() == "gloo" and any(getattr(t, "is_cuda", False) for t in tensors)
    for t in tensors:
        if on_gloo_device:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)        # gloo has no device-tensor reduce; all_reduce gives rank dst the same sum
        else:
            dist.reduce(t, dst=dst, op=dist.ReduceOp.SUM)
    return tensors
