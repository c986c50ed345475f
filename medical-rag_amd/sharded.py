"""Row-sharded search across the GPUs of one node: one process per GPU.

Each rank holds the contiguous row block  [lo, hi)  of the corpus in its own
HBM (contiguity keeps the reference's lowest-row-wins tie rule: local order is
global order and a lower rank means lower rows).  A search is
    local top-k (HIP scan)  ->  ONE all-gather of the packed (id, score)
    candidates over RCCL/xGMI  ->  k-way merge (HIP) on every rank,
so every rank ends with the same global result.  The exchange is B*k*12 bytes
per rank; it is latency-, not bandwidth-bound.  With the library's own
communicator (_native.Comm, mtr_comm_create) the three steps are one C call,
mtr_sharded_search_device; torch.distributed then only carries the RCCL unique
id to the ranks.

The reference is single-process and single-shard
(pipeline/utils/local_rag.py:71-78); this module is what lets the same search
cover corpora that exceed one GPU.
"""
from __future__ import annotations



def shard_bounds(n_rows: int, world_size: int, rank: int):
    """Rows [lo, hi) owned by `rank`: contiguous blocks of n // world rows, the
    first n % world ranks one row longer, so no rank is empty unless there are
    fewer rows than ranks."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad world_size/rank")
    if n_rows < 0:
        raise ValueError("n_rows must not be negative")
    per, extra = divmod(n_rows, world_size)
    lo = rank * per + min(rank, extra)
    return lo, lo + per + (1 if rank < extra else 0)


class CandidateBlock:
    """One rank's (B, k) candidates in a single buffer, the unit the all-gather
    moves: B*k int64 ids, then B*k float32 scores, padded to whole int64 words.
    `ids` and `scores` are views, so the search kernel writes straight into the
    send buffer and the merge kernel reads straight out of the gathered one."""

    def __init__(self, b: int, k: int, device, n_blocks: int = 1):
        import torch
        self.b, self.k, self.n_blocks = b, k, n_blocks
        self.words = b * k + (b * k + 1) // 2           # int64 words per block
        self.buf = torch.zeros((n_blocks, self.words), dtype=torch.int64, device=device)
        self.ids = self.buf[:, : b * k].view(n_blocks, b, k)
        self.scores = self.buf.view(torch.float32).view(n_blocks, 2 * self.words)[:, 2 * b * k: 3 * b * k] \
            .view(n_blocks, b, k)


def gather_candidates(idx, score, group=None):
    """All-gather of every rank's (B, k) candidates.  Returns (ids (W, B, k)
    int64, scores (W, B, k) float32) views of one gathered buffer on the
    caller's device.  Works on GPU tensors over RCCL (backend "nccl") and on CPU
    tensors over gloo."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    b, k = idx.shape
    mine = CandidateBlock(b, k, idx.device)
    mine.ids[0].copy_(idx)
    mine.scores[0].copy_(score)
    everyone = CandidateBlock(b, k, idx.device, n_blocks=world)
    exchange(mine, everyone, group)
    return everyone.ids, everyone.scores


def exchange(mine: CandidateBlock, everyone: CandidateBlock, group=None):
    """All-gather of one candidate block per rank through torch.distributed:
    RCCL for device tensors under backend "nccl"; gloo for CPU tensors, and for
    device tensors staged through the host (ranks that have no RCCL
    communicator between them, e.g. several ranks sharing one GPU in a test:
    the copies synchronise the stream)."""
    import torch
    import torch.distributed as dist
    if dist.get_backend(group) == "gloo":
        if everyone.buf.is_cuda:
            host = torch.empty(everyone.buf.shape, dtype=everyone.buf.dtype)
            dist.all_gather([host[r] for r in range(everyone.n_blocks)], mine.buf[0].cpu(), group=group)
            everyone.buf.copy_(host)
            return
        parts = [everyone.buf[r] for r in range(everyone.n_blocks)]
        dist.all_gather(parts, mine.buf[0], group=group)
    else:
        dist.all_gather_into_tensor(everyone.buf, mine.buf, group=group)


class PendingSearch:
    """A search whose exchange and final merge run on the index's side stream.
    result() makes torch's current stream wait for them and returns the
    (ids, scores) tensors; they stay valid until `depth` later searches of the
    same shape have been started on the same index."""

    def __init__(self, ids, scores, done):
        self._ids, self._scores, self._done = ids, scores, done

    def result(self):
        import torch
        if self._done is not None:
            torch.cuda.current_stream(self._ids.device).wait_event(self._done)
        return self._ids, self._scores


class ShardedIndex:
    """This rank's shard plus what it exchanges candidates with.

    shard: a DeviceShard, or None when this rank holds no rows (fewer rows than
    ranks): it then contributes a block of padding (id -1, score -inf).
    comm: the library's own RCCL communicator (_native.Comm) -- local search,
    all-gather and merge are then ONE C call on one stream.  Without it the
    exchange goes through torch.distributed (`group`), which is what the CPU
    tests run over gloo, and `merge` does the final selection
    (default: the HIP merge kernel)."""

    def __init__(self, shard, group=None, always_exchange=False, comm=None, merge=None):
        self.shard = shard
        self.group = group
        self.comm = comm
        # always_exchange: run the all-gather + merge even on a single rank (a
        # rehearsal of the multi-GPU path on one GPU).
        self.always_exchange = always_exchange
        self._merge = merge
        self._bufs = {}
        self._rings = {}
        self._side = None

    @property
    def world_size(self):
        if self.comm is not None:
            return self.comm.world
        import torch.distributed as dist
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    def _buffers(self, b, k, device):
        import torch
        key = (b, k, str(device))
        hit = self._bufs.get(key)
        if hit is None:
            hit = (CandidateBlock(b, k, device), CandidateBlock(b, k, device, n_blocks=self.world_size),
                   torch.empty((b, k), dtype=torch.int64, device=device),
                   torch.empty((b, k), dtype=torch.float32, device=device))
            self._bufs[key] = hit
        return hit

    def _local_search(self, queries, k, metric, mine):
        """This rank's candidates into its block of the exchange buffer."""
        if self.shard is None:
            mine.ids[0].fill_(-1)
            mine.scores[0].fill_(float("-inf"))
        else:
            self.shard.search_device(queries, k=k, metric=metric, out=(mine.ids[0], mine.scores[0]))

    def _merge_all(self, everyone, k, out):
        if self._merge is not None:
            ids, scores = self._merge(everyone.ids, everyone.scores, k)
            out[0].copy_(ids)
            out[1].copy_(scores)
            return out
        from . import _native
        return _native.merge_topk_device(everyone.ids, everyone.scores, k, out=out)

    def search_device(self, queries, k: int = 10, metric: str = "cosine"):
        """queries: (B, d) tensor on this rank's device, identical on every rank.
        Returns the global (ids (B, k) int64, scores (B, k) float32) on every
        rank, enqueued on torch's current stream.  The returned tensors are
        reused by the next call with the same (B, k)."""
        if queries.dim() == 1:
            queries = queries[None, :]
        b = queries.shape[0]
        mine, everyone, out_i, out_s = self._buffers(b, k, queries.device)
        if self.world_size == 1 and not self.always_exchange:
            if self.shard is None:
                raise ValueError("a single rank cannot search without rows")
            return self.shard.search_device(queries, k=k, metric=metric, out=(out_i, out_s))
        if self.comm is not None:
            return self.comm.sharded_search(self.shard, queries, (out_i, out_s), k=k, metric=metric)
        self._local_search(queries, k, metric, mine)
        exchange(mine, everyone, self.group)
        return self._merge_all(everyone, k, (out_i, out_s))

    def search_device_async(self, queries, k: int = 10, metric: str = "cosine", depth: int = 2):
        """Throughput form for streams of independent searches.  The local search is
        enqueued on torch's current stream; the all-gather and the final merge
        go to a side stream, so the next search's scan overlaps with this one's
        exchange.  Returns a PendingSearch.  `depth` sets of buffers are cycled:
        at most `depth` searches of one shape may be pending (and at most three
        with the library's communicator, which cycles four exchange buffers)."""
        import torch
        if queries.dim() == 1:
            queries = queries[None, :]
        b = queries.shape[0]
        device = queries.device
        if self.world_size == 1 and not self.always_exchange:
            ids, scores = self.search_device(queries, k=k, metric=metric)
            return PendingSearch(ids, scores, None)
        if self.comm is not None and depth > 3:
            raise ValueError("at most three searches may be pending on the library's communicator")
        key = (b, k, str(device), depth)
        ring = self._rings.get(key)
        if ring is None:
            ring = {"next": 0, "slots": [
                {"mine": CandidateBlock(b, k, device), "everyone": CandidateBlock(b, k, device, n_blocks=self.world_size),
                 "ids": torch.empty((b, k), dtype=torch.int64, device=device),
                 "scores": torch.empty((b, k), dtype=torch.float32, device=device),
                 "local_done": torch.cuda.Event(), "done": None} for _ in range(depth)]}
            self._rings[key] = ring
        if self._side is None:
            self._side = torch.cuda.Stream(device=device)
        slot = ring["slots"][ring["next"]]
        ring["next"] = (ring["next"] + 1) % depth
        main = torch.cuda.current_stream(device)
        if slot["done"] is not None:
            main.wait_event(slot["done"])   # the slot's previous exchange has read `mine` and written the outputs
        mine, everyone = slot["mine"], slot["everyone"]
        self._local_search(queries, k, metric, mine)
        slot["local_done"].record(main)
        with torch.cuda.stream(self._side):
            self._side.wait_event(slot["local_done"])
            if self.comm is not None:
                local = (mine.ids[0], mine.scores[0]) if self.shard is not None else None
                self.comm.exchange_merge(local, (slot["ids"], slot["scores"]), k)
            else:
                exchange(mine, everyone, self.group)
                self._merge_all(everyone, k, (slot["ids"], slot["scores"]))
            done = torch.cuda.Event()
            done.record(self._side)
        slot["done"] = done
        return PendingSearch(slot["ids"], slot["scores"], done)
