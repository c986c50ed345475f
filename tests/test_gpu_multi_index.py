"""One index over several GPUs driven by ONE process (mtr_multi_index_*, the
form the reference's single-process caller needs: pipeline/async_main.py:190-195,
pipeline/utils/retrieval.py:70-76), through the C ABI.

This box has one GPU, so the device list names GPU 0 several times: every
block is then its own shard with its own stream and buffers on that GPU, and
the exchange goes by device copies (RCCL puts one rank on one device).  The
grouped all-gather form runs over the real RCCL with a one-device list, and
with three blocks over the tests' in-process stand-in (tests/rccl_double).
What more than one physical GPU adds -- peer copies and RCCL over xGMI -- has
not run anywhere yet."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import local_rag_oracle as orc
from parity import check_topk

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def native():
    import torch
    assert torch.cuda.is_available()
    from medical_rag_amd import _native
    return _native


@pytest.fixture(scope="module")
def rag():
    from medical_rag_amd import local_rag
    return local_rag


@pytest.mark.parametrize("store", ["fp32", "bf16"])
def test_blocks_on_one_gpu_equal_the_unsharded_oracle(native, store):
    """Tie-heavy integer data under the inner-product metric (exact in fp32 and
    bf16): ids and scores equal the oracle over ALL rows bit for bit -- uneven
    blocks, fewer rows than blocks, ties across block borders, single queries and
    batches, host and device entry, shapes changing from call to call."""
    import torch
    rng = np.random.default_rng(3030)
    for n, d, devices in ((9001, 128, [0, 0, 0]), (2, 64, [0, 0, 0]), (5, 32, [0] * 5), (30_000, 256, [0, 0]),
                          (1, 16, [0, 0]), (4097, 100, [0] * 7)):
        corpus = rng.integers(-2, 3, size=(n, d)).astype(np.float32)
        if n > 10:
            corpus[n - 1] = corpus[0]                    # a tie between the first and the last block
            corpus[n // len(devices)] = corpus[n // len(devices) - 1]   # ... and across a block border
        index = native.MultiShard.from_host(corpus, devices, store_dtype=store, normalize=False)
        info = index.info()
        assert info["exchange"] == "copy" and len(info["blocks"]) == len(devices)
        assert [blk["rows"][0] for blk in info["blocks"]] == sorted(blk["rows"][0] for blk in info["blocks"])
        assert info["blocks"][0]["rows"][0] == 0 and info["blocks"][-1]["rows"][1] == n
        for b, k in ((1, 10), (3, 10), (1, 1), (40, 25), (130, 7), (2, 300)):
            queries = rng.integers(-2, 3, size=(b, d)).astype(np.float32)
            want_i, want_s = orc.search(queries, corpus, k, metric="ip")
            got_i, got_s = index.search(queries, k=k, metric="ip")
            assert (got_i == want_i).all(), (n, d, devices, b, k)
            assert (got_s == want_s.astype(np.float32)).all(), (n, d, devices, b, k)
            q = torch.from_numpy(queries).cuda()
            dev_i, dev_s = index.search_device(q, k=k, metric="ip")
            torch.cuda.synchronize()
            assert (dev_i.cpu().numpy() == want_i).all() and (dev_s.cpu().numpy() == want_s.astype(np.float32)).all()
        np.testing.assert_array_equal(index.read_rows(), corpus)
        index.close()


def test_real_data_equals_one_shard(native):
    """Real-valued rows: the blocks' merged answer names the same rows in the same
    order as mtr_index_search on ONE shard of all rows, with the same scores, and
    agrees with the float64 oracle within the storage tolerance."""
    rng = np.random.default_rng(41)
    n, d = 50_000, 256
    corpus = rng.standard_normal((n, d)).astype(np.float32)
    corpus[40_000] = corpus[7]                            # an exact tie far apart, in different blocks
    for store, tol in (("fp32", 1e-6), ("bf16", 1e-3)):
        one = native.DeviceShard.from_host(corpus, store_dtype=store)
        many = native.MultiShard.from_host(corpus, [0, 0, 0, 0], store_dtype=store)
        for b, k in ((1, 1), (1, 10), (6, 10), (64, 33), (200, 5)):
            queries = rng.standard_normal((b, d)).astype(np.float32)
            queries[0] = corpus[7]
            want_i, want_s = one.search(queries, k=k)
            got_i, got_s = many.search(queries, k=k)
            # same rows in the same order; the scores are the same sums, bit for bit where a block
            # and the whole take the same kernel form (the single-query scan), else up to the last
            # bits (a small block's batch may be cut into query tiles of another width)
            assert (got_i == want_i).all(), (store, b, k)
            assert np.abs(got_s - want_s).max() <= 1e-6 and (b > 1 or (got_s == want_s).all()), (store, b, k)
            assert got_i[0, 0] == 7 and (k == 1 or got_i[0, 1] == 40_000)
            check_topk(got_i, got_s, orc.scores(queries, corpus), k, tol)
        one.close()
        many.close()


def test_back_to_back_device_searches_do_not_overtake_each_other(native):
    """Searches enqueued one behind the other without waiting in between: a block's
    stream must not overwrite the gather buffer while the previous merge reads it."""
    import torch
    rng = np.random.default_rng(77)
    n, d, k = 120_000, 128, 10
    corpus = rng.integers(-3, 4, size=(n, d)).astype(np.float32)
    index = native.MultiShard.from_host(corpus, [0, 0, 0], store_dtype="bf16", normalize=False)
    batches = [rng.integers(-3, 4, size=(b, d)).astype(np.float32) for b in (1, 1, 5, 1, 5, 5, 1, 1)]
    side = torch.cuda.Stream()
    outs = []
    with torch.cuda.stream(side):
        for x in batches:
            q = torch.from_numpy(x).cuda(non_blocking=False)
            outs.append(index.search_device(q, k=k, metric="ip") + (q,))
    torch.cuda.synchronize()
    for x, (gi, gs, _) in zip(batches, outs):
        want_i, want_s = orc.search(x, corpus, k, metric="ip")
        assert (gi.cpu().numpy() == want_i).all() and (gs.cpu().numpy() == want_s.astype(np.float32)).all()
    assert not index.nonfinite()
    bad = torch.from_numpy(batches[2]).cuda()
    bad[1, 3] = float("nan")
    index.search_device(bad, k=k, metric="ip")
    assert index.nonfinite()
    index.close()


def test_grouped_all_gather_over_the_real_rccl_with_one_device(native):
    """ncclCommInitAll + ncclGroupStart / ncclAllGather / ncclGroupEnd of the real
    RCCL, which a one-GPU box can run with a one-device list: same answer as the
    shard itself, bit for bit."""
    rng = np.random.default_rng(5)
    n, d = 20_000, 192
    corpus = rng.standard_normal((n, d)).astype(np.float32)
    one = native.DeviceShard.from_host(corpus, store_dtype="bf16")
    index = native.MultiShard.from_host(corpus, [0], store_dtype="bf16", exchange="rccl")
    assert index.info()["exchange"] == "rccl"
    for b, k in ((1, 10), (9, 4), (1, 10), (70, 100)):
        queries = rng.standard_normal((b, d)).astype(np.float32)
        want_i, want_s = one.search(queries, k=k)
        got_i, got_s = index.search(queries, k=k)
        assert (got_i == want_i).all() and (got_s == want_s).all(), (b, k)
    one.close()
    index.close()


_STAND_IN_SCRIPT = r"""
import sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(tests)r)
import numpy as np, torch
from medical_rag_amd import _native
from oracle import local_rag_oracle as orc
rng = np.random.default_rng(11)
ok = True
for n, d, devices in ((20011, 192, [0, 0, 0]), (2, 64, [0, 0, 0]), (9000, 128, [0] * 8)):
    corpus = rng.integers(-3, 4, size=(n, d)).astype(np.float32)
    corpus[n - 1] = corpus[0]
    index = _native.MultiShard.from_host(corpus, devices, store_dtype="bf16", normalize=False, exchange="rccl")
    ok = ok and index.info()["exchange"] == "rccl"
    for b, k in ((1, 10), (40, 7), (1, 10), (3, 33), (130, 5), (2, 300)):
        queries = rng.integers(-3, 4, size=(b, d)).astype(np.float32)
        wi, ws = orc.search(queries, corpus, k, metric="ip")
        gi, gs = index.search(queries, k=k, metric="ip")
        ok = ok and bool((gi == wi).all()) and bool((gs == ws.astype(np.float32)).all())
        di, ds = index.search_device(torch.from_numpy(queries).cuda(), k=k, metric="ip")
        torch.cuda.synchronize()
        ok = ok and bool((di.cpu().numpy() == wi).all()) and bool((ds.cpu().numpy() == ws.astype(np.float32)).all())
    index.close()
print("STAND_IN_OK" if ok else "STAND_IN_MISMATCH")
"""


def test_grouped_all_gather_with_several_blocks_over_a_stand_in():
    """The RCCL form of the exchange with MORE THAN ONE BLOCK: per-block send and
    receive buffers, the group around the all-gathers, the merge over the gathered
    blocks, a block without rows.  RCCL refuses a device list with repeats, so the
    collective underneath is the tests' in-process stand-in (tests/rccl_double,
    chosen with MTR_RCCL_LIBRARY, which is read once per process: hence the child
    process): this checks the library's code around the collective, not RCCL."""
    double = os.path.join(ROOT, "tests", "rccl_double", "librccl_double.so")
    assert os.path.exists(double), "build it with `make -C medical-rag_amd/csrc double`"
    env = dict(os.environ, MTR_RCCL_LIBRARY=double)
    script = _STAND_IN_SCRIPT % {"root": ROOT, "tests": os.path.join(ROOT, "tests")}
    run = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=300)
    assert run.returncode == 0 and "STAND_IN_OK" in run.stdout, run.stdout + run.stderr


def test_facade_over_a_device_list(rag, tmp_path):
    """load_embeddings / retrieve_top_index / get_surrounding_pages / search with
    the retriever configured for a LIST of devices (the optional config key
    local_retriever_devices): the caller's code is the reference's, the index
    behind it is row-sharded.  JSON and sidecar loads, error conventions."""
    rng = np.random.default_rng(8)
    n, d = 403, 96
    mat = rng.standard_normal((n, d)).astype(np.float32)
    mat[300] = mat[3] * 2.0                               # same direction: a cosine tie, lowest position wins
    pages = [{"index": i % 100, "text": f"page {i}", "embedding": row.tolist()} for i, row in enumerate(mat)]
    path = tmp_path / "embedded_pages_hf_multi.json"
    path.write_text(json.dumps(pages), encoding="utf-8")
    before = rag.configure()
    try:
        rag.configure_from_config({"local_retriever_devices": [0, 0, 0]})
        index = rag.load_embeddings(str(path))
        assert len(index) == n and index and len(index.shard.info()["blocks"]) == 3
        queries = rng.standard_normal((34, d)).astype(np.float32)
        queries[0] = mat[3]
        for q in queries[:6]:
            want = orc.retrieve_top_index(q, pages)
            assert rag.retrieve_top_index(q, index) == want
            assert rag.get_surrounding_pages(want, index) == orc.get_surrounding_pages(want, pages)
        assert rag.retrieve_top_index(queries[0], index) == 3
        idx, score = rag.search(index, queries, k=10)
        check_topk(idx, score, orc.scores(queries, mat), 10, 1e-6, exact_idx=True)
        # plain lists handed to retrieve_top_index go over the same device list
        assert rag.retrieve_top_index(queries[1], pages) == orc.retrieve_top_index(queries[1], pages)
        # the sidecar written from the sharded index loads into a sharded index with the same answers
        index.save_sidecar(str(path))
        again = rag.load_embeddings(str(path))
        assert len(again.shard.info()["blocks"]) == 3
        idx2, score2 = rag.search(again, queries, k=10)
        assert (idx2 == idx).all() and (score2 == score).all()
        # sklearn's conventions
        bad = queries[2].copy()
        bad[5] = np.nan
        with pytest.raises(ValueError):
            rag.retrieve_top_index(bad, index)
        assert rag.retrieve_top_index(np.array([]), index) == 0
        nan_pages = [dict(p) for p in pages]
        nan_pages[400] = dict(nan_pages[400], embedding=[float("inf")] + nan_pages[400]["embedding"][1:])
        with pytest.raises(ValueError):
            rag.retrieve_top_index(queries[0], nan_pages)
        index.close()
        again.close()
    finally:
        rag.configure(**before)
        rag.invalidate()


def test_argument_checks(native):
    import ctypes as C
    L = native.lib()
    out = C.c_void_p()
    devs = (C.c_int32 * 2)(0, 0)
    assert L.mtr_multi_index_alloc(10, 8, native.MTR_F32, 1, devs, 0, 0, C.byref(out)) == native.MTR_ERR_INVALID
    assert L.mtr_multi_index_alloc(10, 8, native.MTR_F32, 1, devs, 2, 7, C.byref(out)) == native.MTR_ERR_INVALID
    far = (C.c_int32 * 2)(0, 63)
    assert L.mtr_multi_index_alloc(10, 8, native.MTR_F32, 1, far, 2, 0, C.byref(out)) == native.MTR_ERR_INVALID
    assert "device 63" in native.last_error()
    index = native.MultiShard.alloc(10, 8, [0, 0])
    with pytest.raises(ValueError):
        index.load_host(9, np.zeros((2, 8), dtype=np.float32))
    with pytest.raises(ValueError):
        index.search(np.zeros((1, 8), dtype=np.float32), k=native.MTR_MAX_K + 1)
    index.close()
