"""GPU parity: the HIP path (through the C ABI) against the numpy oracle and the
committed golden vectors.  Tolerances: 1e-6 on fp32-stored rows, 1e-3 on
bf16-stored rows; index sets exact apart from near-ties below the tolerance."""
import logging
import zlib

import numpy as np
import pytest

from oracle import local_rag_oracle as orc
from parity import check_topk

pytestmark = pytest.mark.gpu

TOL = {"fp32": 1e-6, "bf16": 1e-3}


@pytest.fixture(scope="module")
def rag():
    import torch
    assert torch.cuda.is_available()
    from medical_rag_amd import local_rag
    return local_rag


@pytest.fixture(scope="module")
def native():
    from medical_rag_amd import _native
    return _native


@pytest.mark.parametrize("store", ["fp32", "bf16"])
def test_g1_miniature(rag, golden, store):
    a, e = golden("g1_c1_miniature")
    index = rag.PageIndex.from_matrix(a["corpus"], store_dtype=store)
    idx, score = index.search(a["queries"], k=e["k"])
    s64 = orc.scores(a["queries"], a["corpus"])
    # 34 queries on a bf16 shard take the matrix-core kernel, which also rounds the
    # queries to bf16; at d = 64 that is the bf16 bound 2^-8, not yet 1e-3.
    tol = TOL[store] if store == "fp32" else 2.0 ** -8
    nt = check_topk(idx, score, s64, e["k"], tol, exact_idx=(store == "fp32"))
    if store == "fp32":
        assert idx.tolist() == e["topk_idx"]
        np.testing.assert_allclose(score, np.array(e["topk_score"]), atol=1e-6, rtol=0)
        for q, want in zip(a["queries"], e["top1"]):
            assert rag.retrieve_top_index(q, index) == want
    print(f"g1 {store}: {nt} near-tie differences")
    index.close()


@pytest.mark.parametrize("case", ["k5", "k_eq_n", "k_gt_n"])
def test_g2_ties(rag, golden, case):
    a, e = golden("g2_ties_degenerate")
    c = e["cases"][case]
    index = rag.PageIndex.from_matrix(a["corpus"], store_dtype="fp32")
    idx, score = index.search(a["queries"], k=c["k"])
    n = a["corpus"].shape[0]
    kk = min(c["k"], n)
    s64 = orc.scores(a["queries"], a["corpus"])
    check_topk(idx, score, s64, c["k"], 1e-6)
    # exact ties (duplicates, same direction at another norm, zero rows) must
    # come out in the reference's order
    assert idx[:, :kk].tolist() == c["topk_idx"]
    index.close()


def test_g2_degenerate(rag, golden):
    a, e = golden("g2_ties_degenerate")
    zero = rag.PageIndex.from_matrix(a["zero_corpus"])
    assert [rag.retrieve_top_index(q, zero) for q in a["queries"]] == e["zero_corpus_top1"]
    idx, score = zero.search(a["queries"], k=3)
    assert idx.tolist() == [[0, 1, 2]] * len(a["queries"]) and (score == 0).all()
    single = rag.PageIndex.from_matrix(a["corpus"][:1])
    assert [rag.retrieve_top_index(q, single) for q in a["queries"]] == e["single_row_top1"]
    tiny = rag.PageIndex.from_matrix(a["tiny_corpus"])  # float64 rows, norms 1e-15 / 1e-14
    s64 = orc.scores(a["queries"].astype(np.float64), a["tiny_corpus"])
    idx, score = tiny.search(a["queries"].astype(np.float64), k=24)
    check_topk(idx, score, s64, 24, 1e-6)
    for got, row in zip(idx, e["tiny"]):
        assert got.tolist() == row["idx"]


@pytest.mark.parametrize("store", ["fp32", "bf16"])
def test_g3_near_ties(rag, golden, store):
    a, e = golden("g3_near_ties")
    index = rag.PageIndex.from_matrix(a["corpus"], store_dtype=store)
    idx, score = index.search(a["query"], k=e["k"])
    s64 = np.array(e["all_scores"])[None, :]
    nt = check_topk(idx, score, s64, e["k"], TOL[store])
    # pairs whose gap is above the tolerance must be ordered as the reference orders them
    pos = {row: j for j, row in enumerate(idx[0].tolist())}
    for p in e["pairs"]:
        if p["gap"] > 2 * TOL[store]:
            assert pos[p["higher_score_row"]] < pos[p["lower_score_row"]], p
    print(f"g3 {store}: {nt} near-tie differences")
    index.close()


def test_g5_errors(rag, golden):
    _, e = golden("g5_errors")
    d = 8
    eye = np.eye(4, d, dtype=np.float32)
    index = rag.PageIndex.from_matrix(eye)
    assert rag.retrieve_top_index(np.array([]), index) == e["empty_query"]
    assert rag.retrieve_top_index(np.ones(d, dtype=np.float32), []) == e["empty_corpus"]
    logging.disable(logging.CRITICAL)
    try:
        assert rag.load_embeddings("/nonexistent/definitely_missing.json") == e["missing_file"]
    finally:
        logging.disable(logging.NOTSET)
    assert rag.embed_queries([], "unused", "cpu") == e["embed_no_queries"]
    nan = eye.copy(); nan[2, 3] = np.nan
    with pytest.raises(ValueError):
        rag.PageIndex.from_matrix(nan)
    inf = eye.copy(); inf[1, 0] = np.inf
    with pytest.raises(ValueError):
        rag.PageIndex.from_matrix(inf)
    with pytest.raises(ValueError):
        rag.retrieve_top_index(np.array([1.0, np.nan] + [0.0] * (d - 2)), index)
    # the index is still usable after a rejected query
    assert rag.retrieve_top_index(eye[2], index) == 2


def test_g6_caller_glue_through_facade(rag, golden, tmp_path):
    import json
    a, e = golden("g6_caller_glue")
    corpus = a["corpus"]
    pages = [{"index": i % 1000, "text": f"page-{i}", "embedding": row.tolist()} for i, row in enumerate(corpus)]
    path = tmp_path / "embedded_pages_hf_test.json"
    path.write_text(json.dumps(pages))
    index = rag.load_embeddings(str(path))
    assert index and len(index) == len(pages)
    kw_map = {}
    for kw in e["unique_keywords"]:
        v = np.random.default_rng(zlib.crc32(kw.encode("utf-8"))).standard_normal(e["d"]).astype(np.float32)
        kw_map[kw] = v / np.float32(np.linalg.norm(v))
    for case, exp in zip(e["cases"], e["expected"]):
        got = {}
        for kw in case["keywords"]:  # the loop of pipeline/utils/retrieval.py:72-76
            emb = kw_map.get(kw)
            if emb is not None and emb.size > 0:
                top = rag.retrieve_top_index(emb, index)
                got[kw] = {"top_idx": top, "pages": rag.get_surrounding_pages(top, index)}
        assert got == exp["local_pages_per_keyword"]


def test_library_first_then_torch_on_the_gpu():
    """A fresh process that searches through the library BEFORE torch is imported
    (the reference's order of load_embeddings and embed_queries) can still put
    its encoder on the GPU afterwards."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r)\n"
        "from medical_rag_amd import local_rag\n"
        "ix = local_rag.PageIndex.from_matrix(np.eye(8, 16, dtype=np.float32))\n"
        "assert local_rag.retrieve_top_index(np.eye(8, 16, dtype=np.float32)[5], ix) == 5\n"
        "import torch\n"
        "t = torch.ones(4, device='cuda') * 2\n"
        "assert float(t.sum()) == 8.0\n"
        "q = torch.eye(8, 16, device='cuda')[2:3].contiguous()\n"
        "assert int(ix.search(q, k=1)[0][0, 0]) == 2\n"
        "print('ok')\n" % root)
    run = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0 and "ok" in run.stdout, run.stdout + run.stderr


def test_plain_list_is_uploaded_once(rag):
    rng = np.random.default_rng(11)
    mat = rng.standard_normal((50, 24)).astype(np.float32)
    pages = [{"index": i, "text": str(i), "embedding": r.tolist()} for i, r in enumerate(mat)]
    q = rng.standard_normal(24).astype(np.float32)
    want = orc.retrieve_top_index(q, pages)
    assert rag.retrieve_top_index(q, pages) == want
    assert rag.retrieve_top_index(q, pages) == want
    assert len(rag._UPLOADED) >= 1


SHAPES = [
    # n, d, b, k
    (1, 3, 1, 1),
    (7, 5, 2, 10),
    (63, 64, 3, 5),
    (1000, 100, 9, 10),
    (4097, 768, 1, 10),
    (4097, 768, 8, 100),
    (2500, 1024, 4, 64),
    (3001, 1024, 5, 65),
    (513, 4096, 2, 10),
    (777, 3584, 34, 10),
    (20000, 96, 1, 1024),
    (600, 40, 2, 600),
]


@pytest.mark.parametrize("store", ["fp32", "bf16"])
@pytest.mark.parametrize("n,d,b,k", SHAPES)
def test_random_shapes(rag, store, n, d, b, k):
    rng = np.random.default_rng(n * 131 + d * 7 + b)
    corpus = rng.standard_normal((n, d)).astype(np.float32)
    queries = rng.standard_normal((b, d)).astype(np.float32) * np.float32(3.0)
    if n > 10:
        corpus[n // 2] = corpus[1]          # exact duplicate far apart
        corpus[n - 1] = corpus[1] * 2.0     # same direction, other norm
    index = rag.PageIndex.from_matrix(corpus, store_dtype=store)
    idx, score = index.search(queries, k=k)
    s64 = orc.scores(queries, corpus)
    tol = TOL[store]
    if store == "bf16" and d < 256:
        # bf16 keeps 8 significant bits: |score error| <= 2^-9 * sum|p_i q_i| <= 2^-9 for unit
        # vectors (2^-8 when the queries are rounded too, as on the matrix-core path for
        # b >= 2), which only averages down below 1e-3 once d is a few hundred.
        tol = 2.0 ** -8
    check_topk(idx, score, s64, k, tol, exact_idx=False)
    if store == "fp32" and n > 10:
        # the duplicate pair ties bit-exactly: the lower row must come first
        for r in range(b):
            row = idx[r].tolist()
            if 1 in row and n // 2 in row:
                assert row.index(1) < row.index(n // 2)
    index.close()


def test_inner_product_metric(rag):
    rng = np.random.default_rng(5)
    corpus = rng.standard_normal((900, 72)).astype(np.float32)
    queries = rng.standard_normal((3, 72)).astype(np.float32)
    index = rag.PageIndex.from_matrix(corpus, store_dtype="fp32", normalize=False)
    idx, score = index.search(queries, k=7, metric="ip")
    s64 = orc.scores(queries, corpus, metric="ip")
    check_topk(idx, score, s64, 7, 1e-4)  # |scores| ~ 10: 1e-6 relative
    with pytest.raises(ValueError):
        index.search(queries, k=7, metric="cosine")
    index.close()


@pytest.mark.parametrize("store", ["fp32", "bf16"])
def test_integer_data_is_exact_with_many_ties(native, store):
    """Small integer vectors under the inner-product metric: every product and
    every partial sum is exact in fp32 and bf16, so ids AND scores must equal
    the oracle bit for bit, and ties are everywhere (scores are small integers).
    Random shapes across both kernels, all tile widths, k below and above n."""
    rng = np.random.default_rng(2024)
    widths = [1, 7, 33, 64, 100, 128, 192, 256, 320]
    for case in range(60):
        n = int(rng.integers(1, 4000))
        d = int(rng.choice(widths))
        b = int(rng.choice([1, 2, 3, 5, 31, 32, 33, 64, 65, 129, 140]))
        k = int(rng.choice([1, 2, 10, 33, 100, 256, 300]))
        corpus = rng.integers(-3, 4, size=(n, d)).astype(np.float32)
        queries = rng.integers(-3, 4, size=(b, d)).astype(np.float32)
        shard = native.DeviceShard.from_host(corpus, store_dtype=store, normalize=False)
        idx, score = shard.search(queries, k=k, metric="ip")
        want_i, want_s = orc.search(queries, corpus, k, metric="ip")
        assert (idx == want_i).all(), (case, n, d, b, k)
        assert (score == want_s.astype(np.float32)).all(), (case, n, d, b, k)
        shard.close()
    # one large tie-heavy case through the sampling launch (forced: the shard is below its size threshold)
    import os
    n, d, b, k = 200_000, 128, 40, 50
    corpus = rng.integers(-2, 3, size=(n, d)).astype(np.float32)
    queries = rng.integers(-2, 3, size=(b, d)).astype(np.float32)
    shard = native.DeviceShard.from_host(corpus, store_dtype=store, normalize=False)
    idx, score = _search_with_env(shard, queries, k, metric="ip", MTR_MFMA_SAMPLE_STRIDE=12)
    want_i, want_s = orc.search(queries, corpus, k, metric="ip")
    assert (idx == want_i).all() and (score == want_s.astype(np.float32)).all()
    shard.close()


def test_fp32_sixteen_query_tiles(native):
    """fp32 shards: the query tiles of 16, 48, 80, 96 and 112 on v_mfma_f32_16x16x4_f32
    (batches of 5..16, 33..48, 65..112, 129.. in several such tiles, and small shards,
    which take narrower tiles; a sampling launch may cut the batch into tiles of
    another width than the full launch) -- plain, with streaming row loads,
    behind a sampling launch (its 16-query maxima form), ragged last tiles, k
    below and above the rows of a tile.  Integer data: ids and scores equal the
    oracle bit for bit."""
    rng = np.random.default_rng(1616)
    for n, d, b, k, opts in ((1000, 64, 5, 10, {}), (4032, 96, 34, 10, {}), (777, 32, 16, 100, {}),
                             (50_000, 128, 9, 7, dict(mfma_stream_rows=1)), (50_000, 128, 16, 33, dict(mfma_sample_stride=3)),
                             (20_001, 64, 12, 1, dict(mfma_sample_stride=2, mfma_stream_rows=1)), (129, 32, 16, 256, {}),
                             (3, 32, 7, 5, {}), (60_000, 256, 2, 10, dict(mfma_min_batch=2)),
                             (70_000, 128, 34, 10, {}), (70_000, 64, 48, 40, dict(mfma_sample_stride=3, mfma_stream_rows=1)),
                             (40_001, 96, 41, 1, dict(mfma_stream_rows=1)), (70_000, 32, 33, 200, {}),
                             (131_072, 64, 65, 10, dict(mfma_sample_stride=5)), (60_001, 128, 90, 33, dict(mfma_sample_stride=2)),
                             (131_072, 32, 112, 5, dict(mfma_sample_stride=5, mfma_stream_rows=1)), (40_000, 64, 200, 10, {})):
        corpus = rng.integers(-3, 4, size=(n, d)).astype(np.float32)
        queries = rng.integers(-3, 4, size=(b, d)).astype(np.float32)
        shard = native.DeviceShard.from_host(corpus, store_dtype="fp32", normalize=False)
        with native.options(**opts):
            idx, score = shard.search(queries, k=k, metric="ip")
        want_i, want_s = orc.search(queries, corpus, k, metric="ip")
        assert (idx == want_i).all(), (n, d, b, k, opts)
        assert (score == want_s.astype(np.float32)).all(), (n, d, b, k, opts)
        shard.close()
    # real-valued data: within the fp32 tolerance of the float64 oracle (every 16 x 16 width: 16, 48, 80, 96, 112)
    n, d, k = 40_000, 256, 10
    corpus = rng.standard_normal((n, d)).astype(np.float32)
    shard = native.DeviceShard.from_host(corpus, store_dtype="fp32")
    for b in (13, 40, 70, 96, 112, 130, 200):
        queries = rng.standard_normal((b, d)).astype(np.float32)
        idx, score = shard.search(queries, k=k)
        check_topk(idx, score, orc.scores(queries, corpus), k, 1e-6)
    shard.close()


def test_rows_of_one_k_step(native):
    """Rows of exactly 128 bytes (64 bf16 or 32 fp32 values): one K step per tile on the
    matrix-core path, plain and behind a sampling launch.  Integer data, bit for bit."""
    rng = np.random.default_rng(128)
    for store, n, d, b, k, opts in (("bf16", 5000, 64, 40, 10, {}), ("fp32", 5000, 32, 9, 10, {}), ("fp32", 777, 32, 34, 3, {}),
                                    ("bf16", 131_072, 64, 130, 33, dict(mfma_sample_stride=5)),
                                    ("fp32", 70_000, 32, 70, 5, dict(mfma_sample_stride=2)), ("bf16", 300, 64, 2, 256, {})):
        corpus = rng.integers(-3, 4, size=(n, d)).astype(np.float32)
        queries = rng.integers(-3, 4, size=(b, d)).astype(np.float32)
        shard = native.DeviceShard.from_host(corpus, store_dtype=store, normalize=False)
        with native.options(**opts):
            idx, score = shard.search(queries, k=k, metric="ip")
        want_i, want_s = orc.search(queries, corpus, k, metric="ip")
        assert (idx == want_i).all() and (score == want_s.astype(np.float32)).all(), (store, n, d, b, k, opts)
        shard.close()


@pytest.mark.parametrize("store", ["fp32", "bf16"])
def test_long_result_lists_on_a_full_grid(native, store):
    """k above 256 (the scan with long per-wave lists, then the streaming merge of
    every workgroup's list) on a shard large enough for one workgroup per CU, ids
    beyond 2^32.  Integer data, so ids and scores must equal the oracle bit for
    bit, ties included; single queries and a pass of several, k up to the maximum."""
    rng = np.random.default_rng(4242)
    n, d = 150_000, 48
    corpus = rng.integers(-3, 4, size=(n, d)).astype(np.float32)
    shard = native.DeviceShard.from_host(corpus, store_dtype=store, normalize=False, row_offset=7_000_000_000)
    for b, k in ((1, 1024), (1, 257), (3, 700), (6, 300)):
        queries = rng.integers(-3, 4, size=(b, d)).astype(np.float32)
        idx, score = shard.search(queries, k=k, metric="ip")
        want_i, want_s = orc.search(queries, corpus, k, metric="ip")
        assert (idx == want_i + 7_000_000_000).all(), (b, k)
        assert (score == want_s.astype(np.float32)).all(), (b, k)
    shard.close()


@pytest.mark.parametrize("store", ["fp32", "bf16"])
def test_stored_rows_are_normalised(rag, store):
    rng = np.random.default_rng(9)
    corpus = (rng.standard_normal((300, 130)) * 5).astype(np.float32)
    corpus[4] = 0
    index = rag.PageIndex.from_matrix(corpus, store_dtype=store)
    back = index.shard.read_rows()
    want = orc.normalize_rows(corpus)
    tol = 1e-7 if store == "fp32" else 2.0 ** -9
    np.testing.assert_allclose(back, want, atol=tol, rtol=tol)
    assert (back[4] == 0).all()
    index.close()


@pytest.mark.parametrize("store", ["fp32", "bf16"])
def test_sharded_merge_is_exact_on_integer_data(native, store):
    """Uneven shards (one of a single row, one shorter than k) of tie-heavy
    integer data, searched shard by shard and merged on the device: ids and
    scores equal the unsharded oracle bit for bit, for small and large batches."""
    import torch
    rng = np.random.default_rng(808)
    n, d = 9000, 128
    corpus = rng.integers(-2, 3, size=(n, d)).astype(np.float32)
    bounds = [0, 1, 8, 2500, 2501, 7000, n]
    shards = [native.DeviceShard.from_host(corpus[lo:hi], store_dtype=store, normalize=False, row_offset=lo)
              for lo, hi in zip(bounds[:-1], bounds[1:])]
    for b, k in ((1, 10), (3, 10), (40, 25), (130, 7)):
        queries = rng.integers(-2, 3, size=(b, d)).astype(np.float32)
        q = torch.from_numpy(queries).cuda()
        parts = [sh.search_device(q, k=k, metric="ip") for sh in shards]
        gi, gs = native.merge_topk_device(torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts]), k)
        torch.cuda.synchronize()
        want_i, want_s = orc.search(queries, corpus, k, metric="ip")
        assert (gi.cpu().numpy() == want_i).all(), (b, k)
        assert (gs.cpu().numpy() == want_s.astype(np.float32)).all(), (b, k)
    for sh in shards:
        sh.close()


def test_device_search_and_merge(native):
    import torch
    rng = np.random.default_rng(21)
    n, d, b, k = 6000, 256, 6, 12
    corpus = rng.standard_normal((n, d)).astype(np.float32)
    corpus[100] = corpus[5100]  # tie across shards
    queries = rng.standard_normal((b, d)).astype(np.float32)
    bounds = [0, 1500, 1501, 4000, n]
    shards = [native.DeviceShard.from_host(corpus[lo:hi], store_dtype="fp32", row_offset=lo)
              for lo, hi in zip(bounds[:-1], bounds[1:])]
    q = torch.from_numpy(queries).cuda()
    parts_i, parts_s = [], []
    for sh in shards:
        i, s = sh.search_device(q, k=k)
        parts_i.append(i)
        parts_s.append(s)
    torch.cuda.synchronize()
    host_i, host_s = shards[2].search(queries, k=k)
    assert (parts_i[2].cpu().numpy() == host_i).all() and (parts_s[2].cpu().numpy() == host_s).all()
    gi, gs = native.merge_topk_device(torch.stack(parts_i), torch.stack(parts_s), k)
    torch.cuda.synchronize()
    s64 = orc.scores(queries, corpus)
    check_topk(gi.cpu().numpy(), gs.cpu().numpy(), s64, k, 1e-6, exact_idx=True)
    # merge against the oracle's merge on the same candidate lists
    want_i, want_s = orc.merge_topk([(i.cpu().numpy(), s.cpu().numpy()) for i, s in zip(parts_i, parts_s)], k)
    assert (gi.cpu().numpy() == want_i).all()
    np.testing.assert_array_equal(gs.cpu().numpy(), want_s.astype(np.float32))
    for sh in shards:
        sh.close()


def test_full_size_properties(native):
    """Size-independent checks at a bench-like size (1M x 256 bf16): planted
    needles are found, results are sorted, idempotent, and a sharded search
    merged on the GPU equals the single-shard search."""
    import torch
    n, d, b, k = 1_000_000, 256, 4, 10
    g = torch.Generator(device="cuda").manual_seed(1234)
    corpus = torch.randn((n, d), generator=g, device="cuda", dtype=torch.float32)
    queries = torch.randn((b, d), generator=g, device="cuda", dtype=torch.float32)
    needles = torch.tensor([[17, 999_999, 500_000], [3, 4, 5], [123_456, 654_321, 1], [777_777, 0, 64]])
    for r in range(b):
        for j, row in enumerate(needles[r].tolist()):
            # cosine to the query ~ 1/sqrt(1 + 0.01 (j+1)^2): 0.995, 0.981, 0.958
            corpus[row] = queries[r] * 2.0 + 0.2 * (j + 1) * torch.randn(d, generator=g, device="cuda")
    whole = native.DeviceShard.alloc(n, d, store_dtype="bf16")
    whole.load_device(0, corpus)
    i1, s1 = whole.search_device(queries, k=k)
    i2, s2 = whole.search_device(queries, k=k)
    torch.cuda.synchronize()
    assert torch.equal(i1, i2) and torch.equal(s1, s2)
    assert (s1[:, :-1] >= s1[:, 1:]).all()
    for r in range(b):
        assert i1[r, :3].tolist() == needles[r].tolist()
    half = n // 2
    parts = []
    for lo, hi in ((0, half), (half, n)):
        sh = native.DeviceShard.alloc(hi - lo, d, store_dtype="bf16", row_offset=lo)
        sh.load_device(0, corpus[lo:hi])
        parts.append(sh.search_device(queries, k=k))
    torch.cuda.synchronize()
    gi, gs = native.merge_topk_device(torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts]), k)
    torch.cuda.synchronize()
    assert torch.equal(gi, i1) and torch.equal(gs, s1)
    # re-score the reported rows on the CPU in float64 from the stored rows
    rows = i1.flatten().cpu().numpy()
    stored = np.stack([whole.read_rows(int(r), 1)[0] for r in rows]).astype(np.float64)
    qn = orc.normalize_rows(queries.cpu().numpy())
    resc = np.einsum("ij,ij->i", stored, np.repeat(qn, k, axis=0)).reshape(b, k)
    # four queries on bf16 rows run on the matrix cores, which round the queries to bf16 too
    np.testing.assert_allclose(s1.cpu().numpy(), resc, atol=1e-3)
    whole.close()


MFMA_SHAPES = [
    # n, d, b, k   (rows that are a multiple of 128 bytes and b >= 2 (bf16) / 9 (fp32) take the matrix-core kernel;
    #               b <= 32 / <= 64 / more select its 32- / 64- / 128-query tiles, k <= 32 the small buffers)
    (5000, 512, 5, 10),
    (3000, 320, 31, 40),
    (4097, 768, 40, 10),
    (2500, 1024, 130, 100),
    (1000, 64, 17, 256),
    (129, 128, 200, 5),
    (70000, 256, 64, 32),
    (300, 192, 257, 300),   # k > 256: served by the scan instead
]


@pytest.mark.parametrize("store", ["bf16", "fp32"])
@pytest.mark.parametrize("n,d,b,k", MFMA_SHAPES)
def test_matrix_core_path(rag, n, d, b, k, store):
    rng = np.random.default_rng(n + d + b + k)
    corpus = rng.standard_normal((n, d)).astype(np.float32)
    queries = rng.standard_normal((b, d)).astype(np.float32)
    corpus[n // 2] = corpus[1]
    corpus[n - 1] = corpus[1] * 2.0
    queries[0] = corpus[1] * 3.0  # hits the tie set exactly
    index = rag.PageIndex.from_matrix(corpus, store_dtype=store)
    idx, score = index.search(queries, k=k)
    s64 = orc.scores(queries, corpus)
    # bf16: rows AND queries are rounded, 1e-3 holds from d of a few hundred; below, the bf16 bound.
    # fp32: the fp32 matrix instruction with blocked summation keeps the scan's 1e-6.
    tol = 1e-6 if store == "fp32" else 1e-3 if d >= 256 else 2.0 ** -8
    nt = check_topk(idx, score, s64, k, tol)
    # bit-equal stored rows tie exactly: lowest row first
    row = idx[0].tolist()
    assert row[:3] == [1, n // 2, n - 1], row[:5]
    print(f"mfma {store} {n}x{d} b={b} k={k}: {nt} near-tie differences")
    index.close()


def test_matrix_core_matches_scan(native):
    """Both kernels on the same bf16 shard: identical row sets apart from
    near-ties (the scan keeps fp32 queries, the matrix-core path rounds them)."""
    import os
    import torch
    rng = np.random.default_rng(77)
    n, d, b, k = 30000, 512, 48, 20
    corpus = rng.standard_normal((n, d)).astype(np.float32)
    queries = rng.standard_normal((b, d)).astype(np.float32)
    shard = native.DeviceShard.from_host(corpus, store_dtype="bf16")
    big_i, big_s = shard.search(queries, k=k)                       # b >= 2: matrix cores
    parts = [shard.search(queries[i:i + 1], k=k) for i in range(b)]  # b = 1: scan
    scan_i = np.concatenate([p[0] for p in parts]); scan_s = np.concatenate([p[1] for p in parts])
    stored = shard.read_rows().astype(np.float64)
    s64 = orc.normalize_rows(queries) @ stored.T   # exact scores of the STORED rows
    check_topk(scan_i, scan_s, s64, k, 1e-5)
    check_topk(big_i, big_s, s64, k, 1e-3)
    shard.close()


@pytest.mark.parametrize("store", ["bf16", "fp32"])
@pytest.mark.parametrize("stride", [2, 5, 16])
def test_sampled_two_launch_search_equals_single_launch(native, store, stride):
    """Large shards get a sampling launch first (every stride-th row tile; the
    k-th largest per-tile best score becomes each query's starting floor).
    Forced here on a small shard: the result must be identical to the single
    launch, ties and a ragged, sampled last tile included."""
    import os
    rng = np.random.default_rng(500 + stride)
    n, d, b, k = 128 * 300 + 45, 256, 37, 24   # 301 row tiles: strides up to 301 // (2 k) = 6 are honoured
    corpus = rng.standard_normal((n, d)).astype(np.float32)
    queries = rng.standard_normal((b, d)).astype(np.float32)
    # exact ties across sampled and other tiles, one of them in the ragged last tile
    for row in (128 * stride + 3, 128 * (stride + 1) + 9, n - 2):
        corpus[row] = corpus[5]
    queries[0] = corpus[5]
    shard = native.DeviceShard.from_host(corpus, store_dtype=store)
    one_i, one_s = _search_with_env(shard, queries, k, MTR_MFMA_SAMPLE_STRIDE=0)
    two_i, two_s = _search_with_env(shard, queries, k, MTR_MFMA_SAMPLE_STRIDE=stride)
    assert (one_i == two_i).all() and (one_s == two_s).all()
    assert one_i[0, :4].tolist() == [5, 128 * stride + 3, 128 * (stride + 1) + 9, n - 2]
    check_topk(two_i, two_s, orc.scores(queries, corpus), k, 1e-6 if store == "fp32" else 2.0 ** -8)
    shard.close()


def _search_with_env(shard, queries, k, metric="cosine", **switches):
    """One search with the given launch-shape switches set (MTR_MFMA_STAGES=3 sets
    the library's process-wide switch "mfma_stages"; restored afterwards)."""
    from medical_rag_amd import _native
    with _native.options(**{name[4:].lower(): int(value) for name, value in switches.items()}):
        return shard.search(queries, k=k, metric=metric)


@pytest.mark.parametrize("store,d", [("bf16", 256), ("bf16", 128), ("fp32", 192), ("fp32", 64)])
@pytest.mark.parametrize("b", [2, 37, 64, 100])
@pytest.mark.parametrize("stride", [0, 3])
def test_streaming_ring_form_equals_two_buffer_form(native, store, d, b, stride):
    """Single-query-tile searches of large shards load the rows with the
    streaming cache policy and, for up to 64 queries, through a three-buffer LDS
    ring with counted waits.  Forced here on a small shard: bit-identical to the
    two-buffer form (same K order), ties, a ragged last tile and row widths of
    two K steps included; with and without the sampling launch."""
    rng = np.random.default_rng(1300 + b + d + stride)
    n, k = 128 * 300 + 51, 24   # 301 row tiles: a sampling stride of 3 is honoured
    corpus = rng.standard_normal((n, d)).astype(np.float32)
    queries = rng.standard_normal((b, d)).astype(np.float32)
    for row in (128 * 3 + 3, 128 * 40 + 9, n - 2):
        corpus[row] = corpus[5]
    queries[0] = corpus[5]
    shard = native.DeviceShard.from_host(corpus, store_dtype=store)
    base_i, base_s = _search_with_env(shard, queries, k, MTR_MFMA_SAMPLE_STRIDE=stride, MTR_MFMA_STREAM_ROWS=0, MTR_MFMA_STAGES=2)
    nt_i, nt_s = _search_with_env(shard, queries, k, MTR_MFMA_SAMPLE_STRIDE=stride, MTR_MFMA_STREAM_ROWS=1, MTR_MFMA_STAGES=2)
    ring_i, ring_s = _search_with_env(shard, queries, k, MTR_MFMA_SAMPLE_STRIDE=stride, MTR_MFMA_STREAM_ROWS=1, MTR_MFMA_STAGES=3)
    assert (base_i == nt_i).all() and (base_s == nt_s).all()
    assert (base_i == ring_i).all() and (base_s == ring_s).all()
    # bf16 tiles of 64 queries take 256-row tiles (eight waves) in the ring form; the 128-row ring stays selectable
    r128_i, r128_s = _search_with_env(shard, queries, k, MTR_MFMA_SAMPLE_STRIDE=stride, MTR_MFMA_STREAM_ROWS=1, MTR_MFMA_STAGES=3,
                                      MTR_MFMA_TILE_ROWS=128)
    assert (base_i == r128_i).all() and (base_s == r128_s).all()
    assert ring_i[0, :4].tolist() == [5, 128 * 3 + 3, 128 * 40 + 9, n - 2]
    check_topk(ring_i, ring_s, orc.scores(queries, corpus), k, 1e-6 if store == "fp32" else 2.0 ** -8)
    # the ring on a shard of a single row tile, and of fewer tiles than workgroups
    for rows in (100, 128 * 5):
        small = native.DeviceShard.from_host(corpus[:rows], store_dtype=store)
        a_i, a_s = _search_with_env(small, queries, k, MTR_MFMA_STREAM_ROWS=0, MTR_MFMA_STAGES=2)
        r_i, r_s = _search_with_env(small, queries, k, MTR_MFMA_STREAM_ROWS=1, MTR_MFMA_STAGES=3)
        assert (a_i == r_i).all() and (a_s == r_s).all()
        small.close()
    shard.close()


@pytest.mark.parametrize("store", ["bf16", "fp32"])
def test_forced_kernel_forms_are_exact_on_random_integer_shapes(native, store):
    """Random shapes (ragged row counts, fewer rows than a tile, k above n, all
    tile widths) with the large-shard forms forced on: streaming row loads, the
    three-buffer ring, the sampling launch and, for bf16, the 256 x 256-tile
    kernel.  Integer data under the inner-product metric: ids and scores must
    equal the oracle bit for bit, ties everywhere."""
    rng = np.random.default_rng(4242 if store == "bf16" else 4343)
    for case in range(36):
        n = int(rng.choice([1, 5, 127, 128, 129, 255, 257, 1000, 4097, 20000, 60001]))
        d = int(rng.choice([64, 128, 192, 256, 320]))
        b = int(rng.choice([2, 3, 31, 32, 33, 64, 65, 128, 129, 300]))
        k = int(rng.choice([1, 5, 10, 33, 100, 256]))
        stride = int(rng.choice([0, 2, 7]))
        env = dict(MTR_MFMA_STREAM_ROWS=1, MTR_MFMA_STAGES=int(rng.choice([2, 3])), MTR_MFMA_SAMPLE_STRIDE=stride,
                   MTR_MFMA_VIA_REGS=1 + case % 2)   # rows through registers, on either matrix instruction, wherever the shape has that form (rows of whole 512 bytes)
        if store == "bf16" and case % 2 == 0:
            env["MTR_WIDE_MIN_BATCH"] = 1
        corpus = rng.integers(-3, 4, size=(n, d)).astype(np.float32)
        queries = rng.integers(-3, 4, size=(b, d)).astype(np.float32)
        shard = native.DeviceShard.from_host(corpus, store_dtype=store, normalize=False)
        idx, score = _search_with_env(shard, queries, k, metric="ip", **env)
        want_i, want_s = orc.search(queries, corpus, k, metric="ip")
        assert (idx == want_i).all(), (case, n, d, b, k, env)
        assert (score == want_s.astype(np.float32)).all(), (case, n, d, b, k, env)
        shard.close()


@pytest.mark.parametrize("b", [3, 37, 256, 300])
@pytest.mark.parametrize("stride", [2, 5])
def test_wide_tile_kernel_equals_general_kernel(native, b, stride):
    """Large batches on large bf16 shards go through the 256 x 256-tile kernel
    (floors from its own sampling form, every score that reaches its floor
    appended, selection afterwards).  Forced here on a small shard and small
    batches.  It multiplies with another matrix instruction than the general
    kernel, so on real-valued data the two may round a sum differently in the
    last bit: rows must agree except where the scores tie to 1e-6, ties, a ragged
    last tile and padding queries included; on integer data (every sum exact)
    ids and scores must be equal bit for bit."""
    rng = np.random.default_rng(900 + b + stride)
    n, d, k = 256 * 150 + 77, 256, 24
    corpus = rng.standard_normal((n, d)).astype(np.float32)
    queries = rng.standard_normal((b, d)).astype(np.float32)
    for row in (256 * stride + 3, 256 * (stride + 1) + 9, n - 2):
        corpus[row] = corpus[5]
    queries[0] = corpus[5]
    shard = native.DeviceShard.from_host(corpus, store_dtype="bf16")
    one_i, one_s = _search_with_env(shard, queries, k, MTR_MFMA_SAMPLE_STRIDE=0)
    wide_i, wide_s = _search_with_env(shard, queries, k, MTR_MFMA_SAMPLE_STRIDE=stride, MTR_WIDE_MIN_BATCH=1)
    assert np.abs(one_s - wide_s).max() <= 1e-6
    differ = one_i != wide_i
    if differ.any():   # a swap inside a run of scores equal to the last bit or two
        assert np.abs(one_s[differ] - wide_s[differ]).max() <= 1e-6
        assert differ.mean() < 0.01
    # bit-equal stored rows tie exactly in either arithmetic: lowest row first
    assert wide_i[0, :4].tolist() == [5, 256 * stride + 3, 256 * (stride + 1) + 9, n - 2]
    check_topk(wide_i, wide_s, orc.scores(queries, corpus), k, 2.0 ** -8)
    shard.close()
    # integer data, inner product: exact in both kernels
    icorpus = rng.integers(-3, 4, size=(n, d)).astype(np.float32)
    iqueries = rng.integers(-3, 4, size=(b, d)).astype(np.float32)
    ishard = native.DeviceShard.from_host(icorpus, store_dtype="bf16", normalize=False)
    a_i, a_s = _search_with_env(ishard, iqueries, k, MTR_MFMA_SAMPLE_STRIDE=0, metric="ip")
    w_i, w_s = _search_with_env(ishard, iqueries, k, MTR_MFMA_SAMPLE_STRIDE=stride, MTR_WIDE_MIN_BATCH=1, metric="ip")
    assert (a_i == w_i).all() and (a_s == w_s).all()
    want_i, want_s = orc.search(iqueries, icorpus, k, metric="ip")
    assert (w_i == want_i).all() and (w_s == want_s.astype(np.float32)).all()
    ishard.close()


@pytest.mark.parametrize("b", [3, 129, 150, 192, 200, 257, 400, 577])
def test_wide_tile_kernel_query_tile_widths(native, b):
    """The 256 x 256-tile kernel also comes with query tiles of 192 and 128 (three and
    two 16-query column tiles per wave instead of four), so that a batch a little above
    128 or 256 does not pay for a whole 256-wide tile.  Every width, forced and as
    planned, on integer rows: ids and scores equal the oracle bit for bit (ties
    everywhere, ragged last row tile, padding queries); on real-valued rows all widths
    return the same bits (same instruction, same order of a sum's terms)."""
    rng = np.random.default_rng(5100 + b)
    n, d, k = 256 * 90 + 131, 192, 10
    icorpus = rng.integers(-3, 4, size=(n, d)).astype(np.float32)
    iqueries = rng.integers(-3, 4, size=(b, d)).astype(np.float32)
    ishard = native.DeviceShard.from_host(icorpus, store_dtype="bf16", normalize=False)
    want_i, want_s = orc.search(iqueries, icorpus, k, metric="ip")
    for width in (0, 128, 192, 256):
        w_i, w_s = _search_with_env(ishard, iqueries, k, MTR_MFMA_SAMPLE_STRIDE=3, MTR_WIDE_MIN_BATCH=1,
                                    MTR_WIDE_TILE_QUERIES=width, metric="ip")
        assert ishard.last_kernel() == "score_filter_wide_kernel"
        assert (w_i == want_i).all() and (w_s == want_s.astype(np.float32)).all(), (b, width)
    ishard.close()
    corpus = rng.standard_normal((n, d)).astype(np.float32)
    queries = rng.standard_normal((b, d)).astype(np.float32)
    shard = native.DeviceShard.from_host(corpus, store_dtype="bf16")
    ref = None
    for width in (256, 192, 128, 0):
        got = _search_with_env(shard, queries, k, MTR_MFMA_SAMPLE_STRIDE=3, MTR_WIDE_MIN_BATCH=1, MTR_WIDE_TILE_QUERIES=width)
        if ref is None:
            ref = got
            check_topk(got[0], got[1], orc.scores(queries, corpus), k, 2.0 ** -8)
        assert (got[0] == ref[0]).all() and (got[1] == ref[1]).all(), (b, width)
    shard.close()


def test_wide_tile_kernel_overflow_falls_back(native):
    """When more scores reach a floor than a key list holds (here: thousands of
    bit-equal rows, which all tie with the floor), the general kernel recomputes
    the batch: same result as without the wide kernel, lowest rows first."""
    rng = np.random.default_rng(77)
    n, d, b, k = 256 * 120, 128, 40, 10
    corpus = rng.standard_normal((n, d)).astype(np.float32)
    corpus[1000:9000] = corpus[7]          # 8000 copies: every one of them reaches query 0's floor
    queries = rng.standard_normal((b, d)).astype(np.float32)
    queries[0] = corpus[7]
    shard = native.DeviceShard.from_host(corpus, store_dtype="bf16")
    one_i, one_s = _search_with_env(shard, queries, k, MTR_MFMA_SAMPLE_STRIDE=0)
    wide_i, wide_s = _search_with_env(shard, queries, k, MTR_MFMA_SAMPLE_STRIDE=4, MTR_WIDE_MIN_BATCH=1)
    # the recomputation IS the general kernel (started without floors): identical
    assert (one_i == wide_i).all() and (one_s == wide_s).all()
    assert one_i[0].tolist() == [7] + list(range(1000, 1009))
    # and a batch without such ties afterwards, on the same handle, takes the wide kernel alone again
    queries2 = rng.standard_normal((b, d)).astype(np.float32)
    a_i, a_s = _search_with_env(shard, queries2, k, MTR_MFMA_SAMPLE_STRIDE=0)
    b_i, b_s = _search_with_env(shard, queries2, k, MTR_MFMA_SAMPLE_STRIDE=4, MTR_WIDE_MIN_BATCH=1)
    assert np.abs(a_s - b_s).max() <= 1e-6
    assert (a_i == b_i).mean() > 0.99 and np.abs(a_s[a_i != b_i] - b_s[a_i != b_i]).max(initial=0.0) <= 1e-6
    shard.close()


@pytest.mark.parametrize("store", ["bf16", "fp32"])
def test_batches_of_changing_size_on_one_shard(native, store):
    """The shard keeps its query staging buffer between searches and clears
    only what an earlier, larger batch left behind: shrinking, growing and
    tile-width-changing batch sizes in a row must each match the oracle."""
    rng = np.random.default_rng(61)
    n, d, k = 3000, 256, 12
    corpus = rng.standard_normal((n, d)).astype(np.float32)
    shard = native.DeviceShard.from_host(corpus, store_dtype=store)
    tol = 1e-6 if store == "fp32" else 1e-3
    for b in (40, 5, 33, 130, 3, 64, 129, 2, 200):
        queries = rng.standard_normal((b, d)).astype(np.float32)
        idx, score = shard.search(queries, k=k)
        check_topk(idx, score, orc.scores(queries, corpus), k, tol)
    shard.close()


def test_repeated_searches_are_identical(native):
    """Row ranges share floors through atomics, so which candidates a range keeps
    depends on timing; the merged result must not.  Same search, ten times."""
    import torch
    gen = torch.Generator(device="cuda").manual_seed(5)
    n, d, b, k = 150_000, 256, 300, 50
    rows = torch.randn((n, d), generator=gen, device="cuda")
    queries = torch.randn((b, d), generator=gen, device="cuda")
    shard = native.DeviceShard.alloc(n, d, store_dtype="bf16")
    shard.load_device(0, rows)
    first_i, first_s = (t.clone() for t in shard.search_device(queries, k=k))
    for _ in range(9):
        idx, score = shard.search_device(queries, k=k)
        assert torch.equal(idx, first_i) and torch.equal(score, first_s)
    # and the result is right: float64 re-scoring of the stored rows
    stored = torch.from_numpy(shard.read_rows()).cuda().double()
    qn = torch.nn.functional.normalize(queries.double(), dim=1)
    want = (qn[:8] @ stored.T).topk(k, dim=1)
    assert torch.equal(want.indices, first_i[:8]) or (want.values - first_s[:8].double()).abs().max() < 1e-3
    shard.close()


def test_concurrent_calls_on_one_handle(native):
    """ctypes drops the GIL during a call, so threads really do enter the
    library together; calls on one handle are serialised inside it and every
    thread must get its own right answer (host-buffer entry point)."""
    import threading
    rng = np.random.default_rng(77)
    n, d, k = 20000, 128, 8
    corpus = rng.standard_normal((n, d)).astype(np.float32)
    shard = native.DeviceShard.from_host(corpus, store_dtype="fp32")
    batches = [rng.standard_normal((b, d)).astype(np.float32) for b in (1, 2, 5, 40, 1, 3, 70, 9)]
    want = [orc.search(q, corpus, k) for q in batches]
    errors = []

    def worker(i):
        try:
            for _ in range(5):
                idx, score = shard.search(batches[i], k=k)
                if not (idx == want[i][0]).all() or np.abs(score - want[i][1]).max() > 1e-6:
                    errors.append(i)
        except Exception as exc:  # noqa: BLE001 -- reported below
            errors.append((i, repr(exc)))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(len(batches))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    shard.close()


def test_async_sharded_search_matches_sync(native):
    """The pipelined form of the row-sharded search (exchange + final merge on a
    side stream) on a one-rank RCCL group: a stream of searches must give what
    the synchronous form gives, whatever the order of result() calls."""
    import torch
    import torch.distributed as dist
    from medical_rag_amd.sharded import ShardedIndex
    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29577", rank=0, world_size=1,
                                device_id=torch.device("cuda:0"))
    try:
        rng = np.random.default_rng(3)
        n, d, k = 30000, 256, 10
        corpus = rng.standard_normal((n, d)).astype(np.float32)
        shard = native.DeviceShard.from_host(corpus, store_dtype="fp32", row_offset=1000)
        index = ShardedIndex(shard, always_exchange=True)
        batches = [torch.from_numpy(rng.standard_normal((b, d)).astype(np.float32)).cuda() for b in (1, 1, 4, 1, 4, 4, 1, 1)]
        want = [tuple(t.clone() for t in index.search_device(q, k=k)) for q in batches]
        pending = [index.search_device_async(q, k=k, depth=2) for q in batches[:2]]
        got = []
        for q in batches[2:]:
            got.append(tuple(t.clone() for t in pending.pop(0).result()))
            pending.append(index.search_device_async(q, k=k, depth=2))
        while pending:
            got.append(tuple(t.clone() for t in pending.pop(0).result()))
        torch.cuda.synchronize()
        for (wi, ws), (gi, gs) in zip(want, got):
            assert torch.equal(wi, gi) and torch.equal(ws, gs)
        assert int(want[0][0].min()) >= 1000   # global ids carry the shard's row offset
        shard.close()
    finally:
        dist.destroy_process_group()


def test_library_communicator_on_one_rank(native):
    """The RCCL exchange behind the C ABI (mtr_comm_create, mtr_sharded_search_device,
    mtr_comm_exchange_merge_device) on a one-rank communicator: local search ->
    all-gather -> merge must give exactly what mtr_index_search_device gives, for
    the fused call, the split (pipelined) form and changing shapes; a rank without
    rows contributes padding only."""
    import torch
    from medical_rag_amd.sharded import ShardedIndex
    comm = native.Comm.create(native.Comm.unique_id(), 0, 1, 0)
    assert (comm.rank, comm.world, comm.device) == (0, 1, 0)
    rng = np.random.default_rng(12)
    n, d = 40000, 256
    corpus = rng.standard_normal((n, d)).astype(np.float32)
    shard = native.DeviceShard.from_host(corpus, store_dtype="bf16", row_offset=5000)
    index = ShardedIndex(shard, comm=comm, always_exchange=True)
    for b, k in ((1, 10), (7, 10), (1, 10), (40, 33), (3, 1), (130, 100)):
        q = torch.from_numpy(rng.standard_normal((b, d)).astype(np.float32)).cuda()
        want_i, want_s = (t.clone() for t in shard.search_device(q, k=k))
        got_i, got_s = index.search_device(q, k=k)
        torch.cuda.synchronize()
        assert torch.equal(got_i, want_i) and torch.equal(got_s, want_s), (b, k)
        pend = [index.search_device_async(q, k=k, depth=2) for _ in range(2)]
        for p in pend:
            pi, ps = p.result()
            torch.cuda.synchronize()
            assert torch.equal(pi, want_i) and torch.equal(ps, want_s), (b, k)
    assert int(got_i.min()) >= 5000
    # no rows on this rank: padding in, padding out
    empty = ShardedIndex(None, comm=comm, always_exchange=True)
    q = torch.from_numpy(rng.standard_normal((2, d)).astype(np.float32)).cuda()
    ei, es = empty.search_device(q, k=4)
    torch.cuda.synchronize()
    assert (ei == -1).all() and torch.isneginf(es).all()
    assert native.lib().mtr_comm_exchange_merge_device(comm._h, None, None, 0, 1, 1, None, None, None) == native.MTR_ERR_INVALID
    assert native.lib().mtr_sharded_search_device(None, None, None, 0, 1, 1, 0, None, None, None) == native.MTR_ERR_INVALID
    shard.close()
    comm.close()


def _two_rank_worker(rank, world, port, ret):
    import os
    import sys
    import torch
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from medical_rag_amd import _native
        from medical_rag_amd.sharded import ShardedIndex, shard_bounds
        from oracle import local_rag_oracle as orc
        torch.cuda.set_device(0)
        rng = np.random.default_rng(99)
        n, d = 30011, 256
        corpus = rng.integers(-3, 4, size=(n, d)).astype(np.float32)      # integer data: exact scores, ties everywhere
        corpus[n - 1] = corpus[0]                                          # a tie between the first and the last rank
        lo, hi = shard_bounds(n, world, rank)
        shard = _native.DeviceShard.from_host(corpus[lo:hi], store_dtype="bf16", normalize=False, row_offset=lo)
        index = ShardedIndex(shard)
        ok = index.world_size == world
        for b, k in ((1, 10), (40, 7), (1, 10), (3, 33)):
            queries = rng.integers(-3, 4, size=(b, d)).astype(np.float32)
            q = torch.from_numpy(queries).cuda()
            gi, gs = index.search_device(q, k=k, metric="ip")
            torch.cuda.synchronize()
            wi, ws = orc.search(queries, corpus, k, metric="ip")
            ok = ok and bool((gi.cpu().numpy() == wi).all()) and bool((gs.cpu().numpy() == ws.astype(np.float32)).all())
        ret[rank] = bool(ok)
        shard.close()
    finally:
        dist.destroy_process_group()


def test_two_processes_search_their_shards_and_merge():
    """The row-sharded search with MORE THAN ONE RANK on hardware: two processes, each
    with its half of the corpus in HBM (both on this box's one GPU, so the exchange goes
    through gloo with host staging: RCCL refuses two ranks on one device), local search
    and final merge in the HIP kernels.  Every rank must end with the unsharded
    oracle's answer, bit for bit on integer data."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_two_rank_worker, args=(2, port, ret), nprocs=2, join=True)
        assert dict(ret) == {0: True, 1: True}


def _double_rank_worker(rank, world, port, ret, n):
    import os
    import sys
    import torch
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      MTR_RCCL_LIBRARY=os.path.join(root, "tests", "rccl_double", "librccl_double.so"))
    dist.init_process_group("gloo", rank=rank, world_size=world)    # carries the communicator id only
    try:
        from medical_rag_amd import _native
        from medical_rag_amd.sharded import ShardedIndex, shard_bounds
        from oracle import local_rag_oracle as orc
        torch.cuda.set_device(0)
        rng = np.random.default_rng(123)
        d = 192
        corpus = rng.integers(-3, 4, size=(n, d)).astype(np.float32)      # integer data: exact scores, ties everywhere
        corpus[n - 1] = corpus[0]                                          # a tie between the first and the last rank
        lo, hi = shard_bounds(n, world, rank)
        shard = None
        if hi > lo:
            shard = _native.DeviceShard.from_host(corpus[lo:hi], store_dtype="bf16", normalize=False, row_offset=lo)
        comm = _native.Comm.from_process_group(0)
        index = ShardedIndex(shard, comm=comm)
        ok = index.world_size == world and comm.world == world and comm.rank == rank
        # one C call per search: local top-k into the send block, all-gather, merge
        for b, k in ((1, 10), (40, 7), (1, 10), (3, 33), (130, 5), (2, 300)):
            queries = rng.integers(-3, 4, size=(b, d)).astype(np.float32)
            q = torch.from_numpy(queries).cuda()
            gi, gs = index.search_device(q, k=k, metric="ip")
            torch.cuda.synchronize()
            wi, ws = orc.search(queries, corpus, k, metric="ip")
            ok = ok and bool((gi.cpu().numpy() == wi).all()) and bool((gs.cpu().numpy() == ws.astype(np.float32)).all())
        # the pipelined form: exchange and merge on a side stream, several searches pending
        for depth in (2, 3):
            batches = [rng.integers(-3, 4, size=(4, d)).astype(np.float32) for _ in range(7)]
            pending = [index.search_device_async(torch.from_numpy(x).cuda(), k=9, metric="ip", depth=depth) for x in batches[:depth - 1]]
            got = []
            for x in batches[depth - 1:]:
                pending.append(index.search_device_async(torch.from_numpy(x).cuda(), k=9, metric="ip", depth=depth))
                gi, gs = pending.pop(0).result()
                torch.cuda.synchronize()
                got.append((gi.cpu().numpy().copy(), gs.cpu().numpy().copy()))
            while pending:
                gi, gs = pending.pop(0).result()
                torch.cuda.synchronize()
                got.append((gi.cpu().numpy().copy(), gs.cpu().numpy().copy()))
            for x, (gi, gs) in zip(batches, got):
                wi, ws = orc.search(x, corpus, 9, metric="ip")
                ok = ok and bool((gi == wi).all()) and bool((gs == ws.astype(np.float32)).all())
        # One rank's own search fails (cosine on rows stored without normalisation is refused on
        # that rank only): it must still enter the all-gather, with padding, so that the others
        # finish -- with the failing rank's rows missing -- and the communicator stays usable.
        normed = None
        if hi > lo and rank != 1:
            normed = _native.DeviceShard.from_host(corpus[lo:hi], store_dtype="fp32", normalize=True, row_offset=lo)
        failing = ShardedIndex(shard if rank == 1 else normed, comm=comm)
        queries = rng.standard_normal((5, d)).astype(np.float32)
        q = torch.from_numpy(queries).cuda()
        lo1, hi1 = shard_bounds(n, world, 1)
        try:
            gi, gs = failing.search_device(q, k=6, metric="cosine")
            torch.cuda.synchronize()
            ok = ok and rank != 1
            s64 = orc.scores(queries, corpus)
            s64[:, lo1:hi1] = -np.inf                                      # the failing rank's rows are missing
            have = n - (hi1 - lo1)
            gi, gs = gi.cpu().numpy(), gs.cpu().numpy()
            ok = ok and not ((gi >= lo1) & (gi < hi1)).any() and bool((gi[:, have:] == -1).all())
            if have >= 6:
                from parity import check_topk
                check_topk(gi, gs, s64, 6, 1e-6)
        except ValueError as exc:
            ok = ok and rank == 1 and "normalize" in str(exc)
        # ... and the next search is a matched collective again
        queries = rng.integers(-3, 4, size=(2, d)).astype(np.float32)
        gi, gs = index.search_device(torch.from_numpy(queries).cuda(), k=5, metric="ip")
        torch.cuda.synchronize()
        wi, ws = orc.search(queries, corpus, 5, metric="ip")
        ok = ok and bool((gi.cpu().numpy() == wi).all()) and bool((gs.cpu().numpy() == ws.astype(np.float32)).all())
        ret[rank] = bool(ok)
        if shard is not None:
            shard.close()
        if normed is not None:
            normed.close()
        comm.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 30011), (3, 20000), (3, 2)])
def test_library_communicator_with_several_ranks_over_a_stand_in(world, n):
    """csrc/comm_rccl.hip with MORE THAN ONE RANK: block layout, buffer rotation, the
    merge over every rank's gathered block, ranks without rows (2 rows over 3 ranks),
    the fused call and the pipelined form, and a search that fails on one rank only
    (that rank still takes part in the all-gather; nobody is left waiting).  RCCL itself refuses two ranks on one
    GPU, so the collective underneath is the tests' shared-memory stand-in
    (tests/rccl_double, chosen with MTR_RCCL_LIBRARY): this checks the library's
    code around the all-gather, not RCCL.  Every rank must end with the unsharded
    oracle's answer, bit for bit on integer data."""
    import os
    import socket
    import torch.multiprocessing as mp
    double = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rccl_double", "librccl_double.so")
    assert os.path.exists(double), "build it with `make -C medical-rag_amd/csrc double`"
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_double_rank_worker, args=(world, port, ret, n), nprocs=world, join=True)
        assert dict(ret) == {r: True for r in range(world)}


def test_merge_reads_packed_gathered_buffer(native):
    """The exchange buffer layout of the multi-GPU path: every rank's ids and
    scores in one block, merged through strided views without copies."""
    import torch
    from medical_rag_amd.sharded import CandidateBlock
    rng = np.random.default_rng(31)
    n, d, b, k, world = 9000, 128, 3, 7, 4
    corpus = rng.standard_normal((n, d)).astype(np.float32)
    queries = rng.standard_normal((b, d)).astype(np.float32)
    q = torch.from_numpy(queries).cuda()
    everyone = CandidateBlock(b, k, q.device, n_blocks=world)
    per = -(-n // world)
    shards = []
    for r in range(world):
        lo, hi = r * per, min(n, (r + 1) * per)
        sh = native.DeviceShard.from_host(corpus[lo:hi], store_dtype="fp32", row_offset=lo)
        sh.search_device(q, k=k, out=(everyone.ids[r], everyone.scores[r]))
        shards.append(sh)
    assert not everyone.ids.is_contiguous()
    gi, gs = native.merge_topk_device(everyone.ids, everyone.scores, k)
    torch.cuda.synchronize()
    check_topk(gi.cpu().numpy(), gs.cpu().numpy(), orc.scores(queries, corpus), k, 1e-6, exact_idx=True)
    for sh in shards:
        sh.close()


def test_sidecar_load_matches_json_load(rag, tmp_path):
    import json
    from medical_rag_amd import index_format
    rng = np.random.default_rng(41)
    mat = rng.standard_normal((500, 96)).astype(np.float32)
    pages = [{"index": i % 7, "text": f"p{i}", "embedding": r.tolist()} for i, r in enumerate(mat)]
    path = tmp_path / "embedded_pages_hf_x.json"
    path.write_text(json.dumps(pages))
    queries = rng.standard_normal((6, 96)).astype(np.float32)
    from_json = rag.load_embeddings(str(path))
    want = from_json.search(queries, k=9)
    index_format.convert(str(path))
    from_sidecar = rag.load_embeddings(str(path))
    got = from_sidecar.search(queries, k=9)
    assert (got[0] == want[0]).all() and (got[1] == want[1]).all()
    assert list(from_sidecar) == list(from_json)
    assert rag.get_surrounding_pages(3, from_sidecar) == rag.get_surrounding_pages(3, from_json)


def test_bf16_sidecar_reproduces_the_shard(rag, tmp_path):
    """A bf16 index file written from a bf16 shard holds its stored rows; loading
    it (no second normalisation) gives the same shard bit for bit, at half the
    bytes of the fp32 file."""
    import json
    import os
    from medical_rag_amd import index_format
    rng = np.random.default_rng(43)
    mat = (rng.standard_normal((700, 192)) * 3).astype(np.float32)
    mat[5] = 0
    pages = [{"index": i % 9, "text": f"p{i}", "embedding": r.tolist()} for i, r in enumerate(mat)]
    path = tmp_path / "embedded_pages_hf_z.json"
    path.write_text(json.dumps(pages))
    queries = rng.standard_normal((40, 192)).astype(np.float32)
    before = rag.configure()
    try:
        rag.configure(store_dtype="bf16")
        from_json = rag.load_embeddings(str(path))
        want = from_json.search(queries, k=7)
        mtr, _ = from_json.save_sidecar(str(path))
        hdr = index_format.read_header(mtr)
        assert hdr["dtype"] == index_format.DTYPE_BF16 and hdr["as_stored"]
        assert os.path.getsize(mtr) == index_format.HEADER.size + mat.size * 2
        from_file = rag.load_embeddings(str(path))
        assert (from_file.shard.read_rows() == from_json.shard.read_rows()).all()
        got = from_file.search(queries, k=7)
        assert (got[0] == want[0]).all() and (got[1] == want[1]).all()
        assert list(from_file) == list(from_json)
        # index_format.convert(..., "bf16") writes the same file
        os.remove(mtr)
        index_format.convert(str(path), store_dtype="bf16")
        again = rag.load_embeddings(str(path))
        assert (again.shard.read_rows() == from_json.shard.read_rows()).all()
        # the bf16 file into an fp32 shard: the same values, normalised in fp32
        rag.configure(store_dtype="fp32")
        as_f32 = rag.load_embeddings(str(path))
        assert as_f32.shard.info()["store_dtype"] == "fp32"
        np.testing.assert_allclose(as_f32.shard.read_rows(), from_json.shard.read_rows(), atol=2e-3, rtol=0)
    finally:
        rag.configure(**before)


def test_plain_list_edited_in_place_is_uploaded_again(rag):
    rng = np.random.default_rng(12)
    mat = rng.standard_normal((60, 16)).astype(np.float32)
    pages = [{"index": i, "text": str(i), "embedding": r.tolist()} for i, r in enumerate(mat)]
    q = mat[7] * 2.0
    assert rag.retrieve_top_index(q, pages) == 7
    pages[0]["embedding"] = (mat[7] * 3.0).tolist()      # page 0 now points the same way: lowest index wins
    assert rag.retrieve_top_index(q, pages) == 0
    pages[0]["embedding"][0] += 50.0                     # edited in place
    assert rag.retrieve_top_index(q, pages) == 7
    rag.invalidate(pages)
    assert rag.retrieve_top_index(q, pages) == 7


def test_batched_caller_matches_reference_loop(rag, golden):
    from medical_rag_amd import caller
    a, e = golden("g6_caller_glue")
    pages = [{"index": i % 1000, "text": f"page-{i}"} for i in range(a["corpus"].shape[0])]
    index = rag.PageIndex.from_matrix(a["corpus"], pages)
    cases = [{"case_id": c["case_id"], "original_keywords": c["keywords"]} for c in e["cases"]]
    assert caller.unique_keywords(cases) == e["unique_keywords"]
    kw_map = {}
    for kw in e["unique_keywords"]:
        v = np.random.default_rng(zlib.crc32(kw.encode("utf-8"))).standard_normal(e["d"]).astype(np.float32)
        kw_map[kw] = v / np.float32(np.linalg.norm(v))
    kw_map["never used"] = np.array([])          # empty vector: skipped, as at retrieval.py:74
    got = caller.local_pages_for_cases(cases, kw_map, index)
    want = [{kw: v["pages"] for kw, v in c["local_pages_per_keyword"].items()} for c in e["expected"]]
    assert got == want


@pytest.mark.parametrize("dtype", ["float32", "bfloat16", "float16"])
def test_pool_normalize_matches_torch(native, dtype):
    import torch
    torch.manual_seed(3)
    for b, t, d in ((5, 17, 1024), (3, 40, 4096), (2, 9, 100), (1, 1, 7)):
        h = torch.randn(b, t, d, device="cuda").to(getattr(torch, dtype))
        mask = (torch.rand(b, t, device="cuda") > 0.3).long()
        mask[:, 0] = 1
        if b > 1:
            mask[1] = 0                                  # an all-padding row: zero vector, no NaN
        got = native.pool_normalize(h, mask)
        hf = h.float()
        m = mask.unsqueeze(-1).expand(hf.size()).float()  # the reference's expression, local_rag.py:61-65
        want = torch.nn.functional.normalize((hf * m).sum(1) / torch.clamp(m.sum(1), min=1e-9), p=2, dim=1)
        assert got.dtype == torch.float32 and got.shape == (b, d)
        assert torch.isfinite(got).all()
        torch.testing.assert_close(got, want, atol=2e-6, rtol=1e-5)


@pytest.mark.parametrize("dtype", ["float32", "bfloat16", "float16"])
def test_pool_normalize_long_sequences_take_the_sliced_form(native, dtype):
    """Index pages run to thousands of tokens: the pooling is then cut over token
    slices and column blocks (two kernels).  Against the reference's torch
    expression, against the one-kernel form, on widths beyond its 8192 columns
    and on rows that are not a multiple of 16 bytes."""
    import torch
    torch.manual_seed(11)
    for b, t, d in ((3, 2048, 1024), (2, 700, 4096), (1, 300, 9000), (2, 260, 1001), (4, 40, 10000)):
        h = torch.randn(b, t, d, device="cuda").to(getattr(torch, dtype))
        lengths = torch.randint(1, t + 1, (b,), device="cuda")
        lengths[0] = t
        mask = (torch.arange(t, device="cuda")[None, :] < lengths[:, None]).long()   # right padding, as the tokenizer pads
        if b > 1:
            mask[1, ::3] = 0                                                        # holes as well
        got = native.pool_normalize(h, mask)
        hf = h.float()
        m = mask.unsqueeze(-1).expand(hf.size()).float()
        want = torch.nn.functional.normalize((hf * m).sum(1) / torch.clamp(m.sum(1), min=1e-9), p=2, dim=1)
        assert torch.isfinite(got).all()
        torch.testing.assert_close(got, want, atol=2e-6, rtol=1e-5)
        if d <= 8192:
            with native.options(pool_sliced=0):
                one = native.pool_normalize(h, mask)
            with native.options(pool_sliced=1):
                two = native.pool_normalize(h, mask)
            torch.testing.assert_close(one, two, atol=2e-6, rtol=1e-5)
            assert torch.equal(two, got) or torch.equal(one, got)


def test_index_builder_on_gpu(rag, golden, tmp_path):
    """The index builder with the encoder on the GPU (pooling in the HIP kernel):
    same vectors as the reference's builder produced for the fixture, the JSON it
    writes is the reference's schema, and the sidecar loads and searches."""
    import json
    import os
    from medical_rag_amd import index_builder
    a, e = golden("g7_encoder")
    tiny = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tiny_encoder")
    splits = tmp_path / "ocr_batches"
    splits.mkdir()
    (splits / "split_1.json").write_text(json.dumps({"pages": e["pages_in"][:3]}))
    (splits / "split_2.json").write_text(json.dumps({"pages": e["pages_in"][3:]}))
    report = index_builder.build(str(splits / "split_*.json"), tiny, "cuda", str(tmp_path), batch_size=2)
    assert report["pages"] == len(e["pages_in"]) and report["embedded"] == len(e["pages_in"]) and report["failed"] == []
    written = json.load(open(report["path"]))
    assert [{k: v for k, v in p.items() if k != "embedding"} for p in written] == e["pages_out"]
    vectors = np.array([p["embedding"] for p in written])
    np.testing.assert_allclose(vectors, a["page_vectors"], atol=2e-5, rtol=0)
    index = rag.load_embeddings(report["path"])           # the sidecar the builder wrote
    assert len(index) == len(written) and list(index) == e["pages_out"]
    idx, score = index.search(a["page_vectors"], k=1)
    assert idx[:, 0].tolist() == list(range(len(written)))
    np.testing.assert_allclose(score[:, 0], 1.0, atol=1e-5)
    # the same pages through the two-kernel (token-sliced) pooling form that long pages take
    from medical_rag_amd import _native
    with _native.options(pool_sliced=1):
        sliced = index_builder.generate_embeddings_hf(e["pages_in"], tiny, "cuda", batch_size=2)
    np.testing.assert_allclose(np.array([p["embedding"] for p in sliced]), a["page_vectors"], atol=2e-5, rtol=0)


def test_embed_queries_on_gpu_matches_reference(rag, golden):
    """Encoder forward in PyTorch-ROCm, pooling + normalisation in the HIP kernel."""
    import os
    a, e = golden("g7_encoder")
    tiny = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tiny_encoder")
    got = rag.embed_queries(e["queries"], tiny, "cuda", batch_size=4)
    np.testing.assert_allclose(np.stack(got), a["queries_bs64"], atol=2e-5, rtol=0)


def test_texts_to_pages_without_leaving_the_gpu(rag, golden):
    """Keyword texts -> encoder -> pooled vectors -> search, all on the device;
    same hits as the host route through embed_queries + numpy search."""
    import os
    from medical_rag_amd import caller
    a, e = golden("g7_encoder")
    tiny = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tiny_encoder")
    texts = list(e["queries"])
    d = a["queries_bs64"].shape[1]
    corpus = np.random.default_rng(8).standard_normal((700, d)).astype(np.float32)
    corpus[41] = a["queries_bs64"][0]           # a page that is exactly the first keyword's vector
    index = rag.PageIndex.from_matrix(corpus)
    dev = rag.embed_queries_device(texts, tiny, "cuda", batch_size=4)
    assert dev.is_cuda and tuple(dev.shape) == (len(texts), d)
    np.testing.assert_allclose(dev.cpu().numpy(), a["queries_bs64"], atol=2e-5, rtol=0)
    di, ds = index.search(dev, k=5)
    assert di.is_cuda and ds.is_cuda
    hi, hs = index.search(np.stack(rag.embed_queries(texts, tiny, "cuda", batch_size=4)), k=5)
    assert (di.cpu().numpy() == hi).all()
    np.testing.assert_allclose(ds.cpu().numpy(), hs, atol=1e-6, rtol=0)
    hits = caller.top_pages_for_texts(texts, tiny, "cuda", index, top_k=1, batch_size=4)
    assert list(hits) == texts and int(hits[texts[0]][0][0]) == 41
    assert [int(hits[t][0][0]) for t in texts] == hi[:, 0].tolist()
    assert rag.embed_queries_device([], tiny, "cuda") is None
    index.close()


@pytest.mark.parametrize("store", ["fp32", "bf16"])
def test_near_duplicate_rows_at_full_width(rag, store):
    """d = 4096 with rows almost parallel to the queries (cosines near 1, so the
    products do not cancel): the long accumulation must still meet the tolerance."""
    rng = np.random.default_rng(99)
    n, d, b, k = 600, 4096, 8, 10
    corpus = rng.standard_normal((n, d)).astype(np.float32)
    queries = rng.standard_normal((b, d)).astype(np.float32)
    for r in range(b):
        for j in range(k):
            corpus[10 + r * 50 + j * 3] = queries[r] * 2.0 + (0.05 + 0.1 * j) * rng.standard_normal(d).astype(np.float32)
    index = rag.PageIndex.from_matrix(corpus, store_dtype=store)
    idx, score = index.search(queries, k=k)
    s64 = orc.scores(queries, corpus)
    check_topk(idx, score, s64, k, TOL[store], exact_idx=(store == "fp32"))
    for r in range(b):
        assert sorted(idx[r].tolist()) == [10 + r * 50 + j * 3 for j in range(k)]
    index.close()
