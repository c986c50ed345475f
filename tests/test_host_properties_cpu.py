"""Property tests (hypothesis) of the host-side logic that the multi-GPU path
and the oracle rest on: shard partition, candidate block layout, oracle merge."""
import numpy as np
from hypothesis import given, settings, strategies as st

from oracle import local_rag_oracle as orc


@given(n=st.integers(0, 10**9), world=st.integers(1, 64))
def test_shard_bounds_partition_the_rows(n, world):
    from medical_rag_amd.sharded import shard_bounds
    spans = [shard_bounds(n, world, r) for r in range(world)]
    assert spans[0][0] == 0 and spans[-1][1] == n
    for (lo, hi), (lo2, _) in zip(spans, spans[1:]):
        assert 0 <= lo <= hi <= n and hi == lo2          # contiguous, ordered, no gap, no overlap
    sizes = [hi - lo for lo, hi in spans]
    assert sizes == [n // world + (1 if r < n % world else 0) for r in range(world)]   # balanced, longer blocks first
    assert n < world or min(sizes) >= 1                                                # no empty rank unless n < world


@given(b=st.integers(1, 9), k=st.integers(1, 17), blocks=st.integers(1, 5))
@settings(max_examples=40, deadline=None)
def test_candidate_block_views_do_not_overlap(b, k, blocks):
    import torch
    from medical_rag_amd.sharded import CandidateBlock
    blk = CandidateBlock(b, k, "cpu", n_blocks=blocks)
    blk.ids[:] = torch.arange(blocks * b * k, dtype=torch.int64).view(blocks, b, k) + 7
    blk.scores[:] = -torch.arange(blocks * b * k, dtype=torch.float32).view(blocks, b, k) - 0.5
    assert torch.equal(blk.ids, torch.arange(blocks * b * k, dtype=torch.int64).view(blocks, b, k) + 7)   # scores did not clobber ids
    assert torch.equal(blk.scores, -torch.arange(blocks * b * k, dtype=torch.float32).view(blocks, b, k) - 0.5)
    assert blk.buf.shape == (blocks, blk.words) and blk.buf.is_contiguous()


@given(data=st.data())
@settings(max_examples=60, deadline=None)
def test_oracle_merge_equals_topk_of_the_whole(data):
    """Merging per-shard top-k lists gives the top-k of the concatenation, ties
    to the lowest row id (the rule the device merge is checked against)."""
    n_shards = data.draw(st.integers(1, 5))
    b = data.draw(st.integers(1, 4))
    k = data.draw(st.integers(1, 12))
    sizes = [data.draw(st.integers(0, 30)) for _ in range(n_shards)]
    rng = np.random.default_rng(data.draw(st.integers(0, 2**31 - 1)))
    total = sum(sizes)
    scores = rng.integers(-3, 4, size=(b, total)).astype(np.float64)   # few distinct values: many ties
    lists, lo = [], 0
    for sz in sizes:
        idx, val = orc.topk_from_scores(scores[:, lo:lo + sz], k) if sz else (np.full((b, k), -1, np.int64), np.full((b, k), -np.inf))
        idx = np.where(idx >= 0, idx + lo, -1)
        lists.append((idx, val))
        lo += sz
    got_i, got_s = orc.merge_topk(lists, k)
    want_i, want_s = orc.topk_from_scores(scores, k) if total else (np.full((b, k), -1, np.int64), np.full((b, k), -np.inf))
    assert (got_i == want_i).all() and (got_s == want_s).all()
