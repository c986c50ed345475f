"""Row-sharded search across the GPUs of one node: one process per GPU.

Each rank holds the contiguous row block  [lo, hi)  of the corpus in its own
HBM (contiguity keeps the reference's lowest-row-wins tie rule: local order is
global order and a lower rank means lower rows).  A search is
    local top-k (HIP scan)  ->  ONE all-gather of the packed (id, score)
    candidates over RCCL/xGMI  ->  k-way merge (HIP) on every rank,
so every rank ends with the same global result.  The exchange is B*k*12 bytes
per rank; it is latency-, not bandwidth-bound.

The reference is single-process and single-shard
(pipeline/utils/local_rag.py:71-78); this module is what lets the same search
cover corpora that exceed one GPU.
"""
from __future__ import annotations

import math


def shard_bounds(n_rows: int, world_size: int, rank: int):
    """Rows [lo, hi) owned by `rank` when n_rows are split into contiguous
    blocks of ceil(n/world) rows (the last blocks may be short or empty)."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad world_size/rank")
    per = math.ceil(n_rows / world_size) if n_rows > 0 else 0
    lo = min(n_rows, rank * per)
    hi = min(n_rows, lo + per)
    return lo, hi


def pack_candidates(idx, score):
    """(B, k) int64 ids + (B, k) float32 scores -> one (B*k*3,) int32 buffer
    [ids as int32 pairs | score bits], the unit the all-gather moves."""
    import torch
    b, k = idx.shape
    buf = torch.empty(b * k * 3, dtype=torch.int32, device=idx.device)
    buf[: 2 * b * k].view(torch.int64).view(b, k).copy_(idx)
    buf[2 * b * k:].view(torch.float32).view(b, k).copy_(score)
    return buf


def gather_candidates(idx, score, group=None):
    """All-gather of every rank's (B, k) candidates.  Returns (ids (W, B, k)
    int64, scores (W, B, k) float32) on the caller's device.  Works on GPU
    tensors over RCCL (backend "nccl") and on CPU tensors over gloo."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    b, k = idx.shape
    mine = pack_candidates(idx.contiguous(), score.contiguous())
    out = torch.empty((world, mine.numel()), dtype=torch.int32, device=mine.device)
    if dist.get_backend(group) == "gloo":
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine, group=group)
        out = torch.stack(parts)
    else:
        dist.all_gather_into_tensor(out, mine, group=group)
    ids = out[:, : 2 * b * k].contiguous().view(torch.int64).view(world, b, k)
    scores = out[:, 2 * b * k:].contiguous().view(torch.float32).view(world, b, k)
    return ids, scores


class ShardedIndex:
    """This rank's shard plus the process group it searches with."""

    def __init__(self, shard, group=None):
        self.shard = shard
        self.group = group

    @property
    def world_size(self):
        import torch.distributed as dist
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    def search_device(self, queries, k: int = 10, metric: str = "cosine"):
        """queries: (B, d) GPU tensor, identical on every rank.  Returns the
        global (ids (B, k) int64, scores (B, k) float32) on every rank,
        enqueued on torch's current stream."""
        from . import _native
        idx, score = self.shard.search_device(queries, k=k, metric=metric)
        if self.world_size == 1:
            return idx, score
        ids, scores = gather_candidates(idx, score, self.group)
        return _native.merge_topk_device(ids, scores, k)
