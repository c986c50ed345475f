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


def test_two_rank_gather_and_merge():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    world = 2
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
        assert dict(ret) == {0: True, 1: True}
