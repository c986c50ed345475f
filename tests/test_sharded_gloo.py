"""The N>1 path on CPU: two gloo ranks exercise the shard split, the packed
candidate all-gather and the merge order.  The local search and the merge are
done by the oracle here (no GPU in this test); the GPU kernels for both are
checked in test_gpu_parity.py."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_cover_rows_exactly():
    from medical_rag_amd.sharded import shard_bounds
    for n in (0, 1, 7, 8, 9, 4032, 10_000_000):
        for world in (1, 2, 3, 4, 8):
            spans = [shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (a, b), (c, d) in zip(spans[:-1], spans[1:]):
                assert b == c and a <= b and c <= d
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1            # balanced: no rank is empty unless n < world
            assert sorted(sizes, reverse=True) == sizes
            if n >= world:
                assert min(sizes) >= 1
    assert [shard_bounds(9, 8, r) for r in range(8)] == [(0, 2)] + [(r + 1, r + 2) for r in range(1, 8)]
    assert [shard_bounds(3, 8, r)[1] - shard_bounds(3, 8, r)[0] for r in range(8)] == [1, 1, 1, 0, 0, 0, 0, 0]
    with pytest.raises(ValueError):
        shard_bounds(10, 2, 2)


def test_candidate_block_views():
    from medical_rag_amd.sharded import CandidateBlock
    for b, k in ((2, 3), (1, 1), (5, 10)):
        blk = CandidateBlock(b, k, "cpu", n_blocks=3)
        assert blk.buf.shape == (3, b * k + (b * k + 1) // 2) and blk.buf.dtype == torch.int64
        idx = torch.arange(3 * b * k, dtype=torch.int64).view(3, b, k) * (2 ** 33) - 1
        score = torch.linspace(-2, 2, 3 * b * k, dtype=torch.float32).view(3, b, k)
        score[0, 0, 0] = float("-inf")
        blk.ids.copy_(idx)
        blk.scores.copy_(score)
        # ids and scores do not overlap and survive a raw copy of the buffer
        clone = CandidateBlock(b, k, "cpu", n_blocks=3)
        clone.buf.copy_(blk.buf)
        assert torch.equal(clone.ids, idx) and torch.equal(clone.scores, score)
        assert clone.ids.stride(0) == blk.words and clone.scores.stride(0) == 2 * blk.words


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from medical_rag_amd.sharded import gather_candidates, shard_bounds
        from oracle import local_rag_oracle as orc
        rng = np.random.default_rng(3)
        n, d, b, k = 1001, 32, 5, 12
        corpus = rng.standard_normal((n, d)).astype(np.float32)
        corpus[900] = corpus[10]  # tie across the two shards
        queries = rng.standard_normal((b, d)).astype(np.float32)
        lo, hi = shard_bounds(n, world, rank)
        li, ls = orc.search(queries, corpus[lo:hi], k)
        li = np.where(li >= 0, li + lo, -1)
        ids, scores = gather_candidates(torch.from_numpy(li), torch.from_numpy(ls.astype(np.float32)))
        assert ids.shape == (world, b, k) and scores.shape == (world, b, k)
        assert torch.equal(ids[rank], torch.from_numpy(li))
        mi, ms = orc.merge_topk([(ids[r].numpy(), scores[r].numpy()) for r in range(world)], k)
        fi, fs = orc.search(queries, corpus, k)
        ok = (mi == fi).all() and np.allclose(ms, fs, atol=1e-6)
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def _spawn(worker, world, *args):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(worker, args=(world, port, ret) + args, nprocs=world, join=True)
        return dict(ret)


def test_two_rank_gather_and_merge():
    assert _spawn(_worker, 2) == {0: True, 1: True}


class _StubShard:
    """Stands in for a DeviceShard on the CPU: the oracle scores this rank's
    rows, with the call signature ShardedIndex uses (results written into the
    caller's block of the exchange buffer, global row ids)."""

    def __init__(self, rows, row_offset):
        self.rows, self.row_offset, self.calls = rows, row_offset, 0

    def search_device(self, queries, k=10, metric="cosine", out=None, stream=None):
        from oracle import local_rag_oracle as orc
        self.calls += 1
        li, ls = orc.search(queries.numpy(), self.rows, k, metric=metric)
        out[0].copy_(torch.from_numpy(np.where(li >= 0, li + self.row_offset, -1)))
        out[1].copy_(torch.from_numpy(ls.astype(np.float32)))
        return out


def _oracle_merge(ids, scores, k):
    from oracle import local_rag_oracle as orc
    mi, ms = orc.merge_topk([(ids[r].numpy(), scores[r].numpy()) for r in range(ids.shape[0])], k)
    return torch.from_numpy(mi), torch.from_numpy(ms.astype(np.float32))


def _sharded_index_worker(rank, world, port, ret, n):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from medical_rag_amd.sharded import ShardedIndex, shard_bounds
        from oracle import local_rag_oracle as orc
        rng = np.random.default_rng(17)
        d = 24
        corpus = rng.standard_normal((n, d)).astype(np.float32)
        if n > 6:
            corpus[n - 1] = corpus[0]          # a tie between the first and the last rank's rows
        lo, hi = shard_bounds(n, world, rank)
        shard = _StubShard(corpus[lo:hi], lo) if hi > lo else None   # fewer rows than ranks: no shard here
        index = ShardedIndex(shard, merge=_oracle_merge)
        ok = index.world_size == world
        # several searches in a row: changing and repeated (B, k) shapes reuse the buffers
        for b, k in ((5, 4), (1, 3), (5, 4), (2, 12), (1, 3)):
            queries = rng.standard_normal((b, d)).astype(np.float32)
            gi, gs = index.search_device(torch.from_numpy(queries), k=k)
            wi, ws = orc.search(queries, corpus, k)
            ok = ok and bool((gi.numpy() == wi).all()) and bool(np.allclose(gs.numpy(), ws, atol=1e-6))
        ok = ok and (shard is None or shard.calls == 5)
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 1001), (3, 2), (2, 9)])
def test_sharded_index_over_gloo(world, n):
    """ShardedIndex.search_device itself on several ranks: its N>1 branch, the
    buffer reuse, the order local search -> all-gather -> merge, and ranks
    without rows (fewer rows than ranks), which must still take part."""
    assert _spawn(_sharded_index_worker, world, n) == {r: True for r in range(world)}
