"""Long result lists through the row scan (k above 32): the scan keeps candidate
lists shorter than k per workgroup, a check kernel proves afterwards that no list
dropped a row of the result, and the search runs again with full lists only when
it cannot.  The answer has to be the full-list answer bit for bit either way:
with real-valued rows (lists short enough), with tie-heavy integer rows (where
ties at the list ends make the check fail and the second run decides), with
the best rows all planted in ONE workgroup's share of the rows (a short list
overflowing with rows of the result), for single queries and passes of several,
and for result lists up to the maximum of 1024 (the workgroup-wide radix merge)."""
import numpy as np
import pytest

from oracle import local_rag_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def native():
    import torch
    assert torch.cuda.is_available()
    from medical_rag_amd import _native
    return _native


@pytest.mark.parametrize("store", ["fp32", "bf16"])
def test_short_lists_give_the_full_list_answer(native, store):
    import torch
    gen = torch.Generator(device="cuda").manual_seed(11)
    n, d = 600_000, 128
    rows = torch.randn((n, d), generator=gen, device="cuda")
    queries = torch.randn((5, d), generator=gen, device="cuda")
    shard = native.DeviceShard.alloc(n, d, store_dtype=store)
    shard.load_device(0, rows)
    for form in (1, 2):                       # classic and streaming single-query scan
        for k in (33, 100, 256, 257, 1024):
            for q in (queries[:1], queries[1:4] if store == "fp32" else queries[1:2]):
                q = q.contiguous()
                # one query per pass in all three runs: a pass of four queries is other machine code than a
                # pass of two (which is all that full lists of 1024 leave room for) and rounds the last bit of
                # a few scores differently; passes of several queries are covered by the integer test below
                common = dict(scan_form=form, mfma_min_batch=1 << 30, scan_max_qb=1)
                with native.options(scan_short_lists=0, **common):
                    fi, fs = (t.clone() for t in shard.search_device(q, k=k))
                with native.options(**common):
                    si, ss = (t.clone() for t in shard.search_device(q, k=k))
                with native.options(scan_short_lists=8, **common):   # far too short: second run
                    ti, ts = (t.clone() for t in shard.search_device(q, k=k))
                torch.cuda.synchronize()
                assert torch.equal(fi, si) and torch.equal(fs, ss), (store, form, k, q.shape[0])
                assert torch.equal(fi, ti) and torch.equal(fs, ts), (store, form, k, q.shape[0])
    # against the oracle once, on the rows as stored
    back = shard.read_rows(0, n)
    q = (queries[:1] / queries[:1].norm()).contiguous()
    gi, gs = shard.search_device(q, k=1024, metric="ip")
    want_i, want_s = orc.search(q.cpu().numpy(), back, 1024, metric="ip")
    got_i = gi.cpu().numpy()
    differ = got_i != want_i
    assert np.abs(gs.cpu().numpy() - want_s).max() <= 5e-6
    assert differ.mean() < 0.02                # swaps among scores equal in fp32 only
    shard.close()


@pytest.mark.parametrize("store", ["fp32", "bf16"])
def test_integer_rows_with_ties_at_the_list_ends(native, store):
    rng = np.random.default_rng(77)
    n, d = 300_000, 64
    corpus = rng.integers(-3, 4, size=(n, d)).astype(np.float32)
    shard = native.DeviceShard.from_host(corpus, store_dtype=store, normalize=False)
    for b, k in ((1, 100), (1, 1024), (2, 300), (1, 64)):
        queries = rng.integers(-3, 4, size=(b, d)).astype(np.float32)
        want_i, want_s = orc.search(queries, corpus, k, metric="ip")
        for opts in ({}, dict(scan_short_lists=32), dict(scan_form=1), dict(mfma_min_batch=1 << 30)):
            with native.options(**opts):
                idx, score = shard.search(queries, k=k, metric="ip")
            assert (idx == want_i).all(), (store, b, k, opts)
            assert (score == want_s.astype(np.float32)).all(), (store, b, k, opts)
    shard.close()


def test_best_rows_all_in_one_workgroups_share(native):
    """The classic form deals groups of 8 rows to (workgroup, wave) pairs in turn;
    with every planted row in workgroup 0's groups its short list overflows with
    rows of the result, which the check has to notice."""
    import torch
    gen = torch.Generator(device="cuda").manual_seed(5)
    n, d, k = 500_000, 64, 256
    rows = torch.randn((n, d), generator=gen, device="cuda")
    q = torch.randn((1, d), generator=gen, device="cuda")
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    groups = torch.arange(n // 8, device="cuda")
    mine = groups[(groups % (cus * 8)) < 8]                       # groups of workgroup 0 (one per wave and round)
    planted = (mine[:, None] * 8 + torch.arange(8, device="cuda")[None, :]).reshape(-1)[:300]
    assert planted.numel() == 300
    amp = torch.linspace(3.0, 1.5, 300, device="cuda")
    rows[planted] = q * amp[:, None] + 0.05 * torch.randn((300, d), generator=gen, device="cuda")
    shard = native.DeviceShard.alloc(n, d, store_dtype="fp32", normalize=False)
    shard.load_device(0, rows)
    with native.options(scan_form=1, mfma_min_batch=1 << 30, scan_short_lists=0):
        fi, fs = (t.clone() for t in shard.search_device(q, k=k, metric="ip"))
    with native.options(scan_form=1, mfma_min_batch=1 << 30):
        si, ss = (t.clone() for t in shard.search_device(q, k=k, metric="ip"))
    torch.cuda.synchronize()
    assert torch.equal(fi, si) and torch.equal(fs, ss)
    assert set(si[0].tolist()) <= set(planted.tolist())           # the 256 best are planted rows of that one workgroup
    shard.close()
