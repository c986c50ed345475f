"""The allocation-free entry point (mtr_index_search_device_noalloc) against the
entry point it is a variant of (mtr_index_search_device), through the C ABI:
same kernels on scratch from the library's static device array, so ids and
scores must be equal bit for bit -- on two calls in a row, the second with a
smaller input, which finds the first call's contents in the array."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def native():
    import torch
    assert torch.cuda.is_available()
    from medical_rag_amd import _native
    return _native


def _both(shard, queries, k, metric="cosine"):
    import torch
    b = queries.shape[0]
    want_i, want_s = shard.search_device(queries, k=k, metric=metric)
    got_i = torch.full((b, k), -7, dtype=torch.int64, device=queries.device)
    got_s = torch.full((b, k), float("nan"), dtype=torch.float32, device=queries.device)
    shard.search_device_noalloc(queries, (got_i, got_s), k=k, metric=metric)
    torch.cuda.synchronize()
    return (want_i, want_s), (got_i, got_s)


@pytest.mark.parametrize("store", ["bf16", "fp32"])
def test_noalloc_equals_search_device_on_consecutive_calls(native, store):
    import torch
    gen = torch.Generator(device="cuda").manual_seed(2027)
    n, d = 150_000, 256
    rows = torch.randn((n, d), generator=gen, device="cuda")
    rows[70_000] = rows[11]            # an exact tie far apart
    shard = native.DeviceShard.alloc(n, d, store_dtype=store)
    shard.load_device(0, rows)
    # matrix-core path (wide batch first, then a smaller one), then the scan (one query), then a batch again
    for b, k in ((61, 24), (5, 7), (1, 10), (1, 3), (40, 100), (2, 1)):
        queries = torch.randn((b, d), generator=gen, device="cuda")
        queries[0] = rows[11]
        (want_i, want_s), (got_i, got_s) = _both(shard, queries, k)
        assert torch.equal(got_i, want_i), (store, b, k)
        assert torch.equal(got_s, want_s), (store, b, k)
        assert got_i[0, :2].tolist() == [11, 70_000][:k]
    # the oracle's answer for the last, smaller call (float64, stored rows)
    from oracle import local_rag_oracle as orc
    from parity import check_topk
    stored = shard.read_rows().astype(np.float64)
    q = queries.cpu().numpy()
    s64 = orc.normalize_rows(q) @ stored.T
    check_topk(got_i.cpu().numpy(), got_s.cpu().numpy(), s64, 1, 1e-3 if store == "bf16" else 1e-6)
    shard.close()


def test_noalloc_with_forced_large_shard_forms(native):
    """Sampling launch + streaming ring forms (what a large shard uses) on the static scratch."""
    import torch
    gen = torch.Generator(device="cuda").manual_seed(5)
    n, d = 128 * 400 + 17, 256
    rows = torch.randn((n, d), generator=gen, device="cuda")
    shard = native.DeviceShard.alloc(n, d, store_dtype="bf16")
    shard.load_device(0, rows)
    with native.options(mfma_sample_stride=3, mfma_stream_rows=1, mfma_stages=3):
        for b, k in ((64, 20), (9, 20)):
            queries = torch.randn((b, d), generator=gen, device="cuda")
            (want_i, want_s), (got_i, got_s) = _both(shard, queries, k)
            assert torch.equal(got_i, want_i) and torch.equal(got_s, want_s)
    with native.options(scan_nontemporal=1):
        queries = torch.randn((1, d), generator=gen, device="cuda")
        (want_i, want_s), (got_i, got_s) = _both(shard, queries, 10)
        assert torch.equal(got_i, want_i) and torch.equal(got_s, want_s)
    shard.close()


def test_noalloc_two_shards_share_the_array(native):
    """Two handles on one GPU use the same static array, one call after the other on one stream."""
    import torch
    gen = torch.Generator(device="cuda").manual_seed(8)
    d = 128
    a_rows = torch.randn((30_000, d), generator=gen, device="cuda")
    b_rows = torch.randn((9_000, d), generator=gen, device="cuda")
    sa = native.DeviceShard.alloc(30_000, d, store_dtype="fp32")
    sb = native.DeviceShard.alloc(9_000, d, store_dtype="bf16", row_offset=30_000)
    sa.load_device(0, a_rows)
    sb.load_device(0, b_rows)
    queries = torch.randn((12, d), generator=gen, device="cuda")
    for shard in (sa, sb, sa):
        (want_i, want_s), (got_i, got_s) = _both(shard, queries, 9)
        assert torch.equal(got_i, want_i) and torch.equal(got_s, want_s)
    assert int(_both(sb, queries, 9)[1][0].min()) >= 30_000
    sa.close()
    sb.close()


def test_noalloc_rejects_what_does_not_fit(native):
    import torch
    gen = torch.Generator(device="cuda").manual_seed(9)
    n, d = 200_000, 256
    shard = native.DeviceShard.alloc(n, d, store_dtype="bf16")
    shard.load_device(0, torch.randn((n, d), generator=gen, device="cuda"))
    b, k = 4096, 256
    queries = torch.randn((b, d), generator=gen, device="cuda")
    out = (torch.empty((b, k), dtype=torch.int64, device="cuda"), torch.empty((b, k), dtype=torch.float32, device="cuda"))
    with pytest.raises(ValueError, match="bytes of scratch"):
        shard.search_device_noalloc(queries, out, k=k)
    # the handle and the array are still good afterwards
    q1 = queries[:3].contiguous()
    (want_i, want_s), (got_i, got_s) = _both(shard, q1, 10)
    assert torch.equal(got_i, want_i) and torch.equal(got_s, want_s)
    shard.close()


def test_noalloc_search_can_be_captured_in_a_graph(native):
    """The callers this entry point is for: a search captured once under stream
    capture and replayed, with new queries copied into the captured input."""
    import torch
    gen = torch.Generator(device="cuda").manual_seed(13)
    n, d, k = 120_000, 256, 10
    shard = native.DeviceShard.alloc(n, d, store_dtype="bf16")
    shard.load_device(0, torch.randn((n, d), generator=gen, device="cuda"))
    for b in (1, 6):
        q = torch.randn((b, d), generator=gen, device="cuda")
        out = (torch.empty((b, k), dtype=torch.int64, device="cuda"), torch.empty((b, k), dtype=torch.float32, device="cuda"))
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            shard.search_device_noalloc(q, out, k=k)      # once outside capture: kernel attributes get set here
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            shard.search_device_noalloc(q, out, k=k, stream=torch.cuda.current_stream().cuda_stream)
        for _ in range(4):          # (the second replay is the first that can see what a replay left behind)
            q.copy_(torch.randn((b, d), generator=gen, device="cuda"))
            out[0].fill_(-9)
            graph.replay()
            torch.cuda.synchronize()
            want_i, want_s = shard.search_device(q, k=k)
            torch.cuda.synchronize()
            assert torch.equal(out[0], want_i) and torch.equal(out[1], want_s)
            assert not shard.nonfinite()
        # A replay with NaN in the queries reports it, the clean replay after it does not, and a
        # search through another entry point in between (which takes a new epoch of the handle's
        # ordinary report word) changes neither: no value frozen into the graph is compared.
        clean = torch.randn((b, d), generator=gen, device="cuda")
        q.copy_(clean)
        q[b - 1, 17] = float("nan")
        graph.replay()
        torch.cuda.synchronize()
        shard.search_device(clean, k=k)
        torch.cuda.synchronize()
        assert not shard.nonfinite()                      # the last search on the handle: the clean one
        graph.replay()
        torch.cuda.synchronize()
        shard.search_device_noalloc(clean, out, k=k)      # the last call through the allocation-free entry: clean
        torch.cuda.synchronize()
        assert not shard.nonfinite()
        q[0, 0] = float("inf")
        with torch.cuda.stream(side):
            shard.search_device_noalloc(q, out, k=k)      # marks the entry point as the handle's last device call
        torch.cuda.synchronize()
        assert shard.nonfinite()
        q.copy_(clean)
        graph.replay()
        torch.cuda.synchronize()
        assert not shard.nonfinite()                      # the replay cleared the word and found nothing
        q[0, 3] = float("nan")
        graph.replay()
        torch.cuda.synchronize()
        assert shard.nonfinite()                          # ... and this one raised it again
        q.copy_(clean)
        graph.replay()
        torch.cuda.synchronize()
        want_i, want_s = shard.search_device(clean, k=k)
        torch.cuda.synchronize()
        assert torch.equal(out[0], want_i) and torch.equal(out[1], want_s)
        del graph
    shard.close()
