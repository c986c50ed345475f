"""Pins the numpy oracle to golden vectors produced by the reference itself
(tests/golden/make_golden.py).  CPU only."""
import logging
import zlib

import numpy as np
import pytest

from oracle import local_rag_oracle as orc


def test_g1_topk_matches_reference(golden):
    a, e = golden("g1_c1_miniature")
    idx, val = orc.search(a["queries"], a["corpus"], e["k"])
    assert idx.tolist() == e["topk_idx"]
    np.testing.assert_allclose(val, np.array(e["topk_score"]), rtol=0, atol=1e-12)
    pages = [{"embedding": r.tolist()} for r in a["corpus"]]
    got = [orc.retrieve_top_index(q, pages) for q in a["queries"][:5]]
    assert got == e["top1"][:5]
    assert idx[:, 0].tolist() == e["top1"]


@pytest.mark.parametrize("case", ["k5", "k_eq_n", "k_gt_n"])
def test_g2_ties(golden, case):
    a, e = golden("g2_ties_degenerate")
    c = e["cases"][case]
    idx, val = orc.search(a["queries"], a["corpus"], c["k"])
    n = a["corpus"].shape[0]
    kk = min(c["k"], n)
    assert idx[:, :kk].tolist() == c["topk_idx"]
    np.testing.assert_allclose(val[:, :kk], np.array(c["topk_score"]), rtol=0, atol=1e-12)
    assert (idx[:, kk:] == -1).all() and np.isneginf(val[:, kk:]).all()


def test_g2_degenerate(golden):
    a, e = golden("g2_ties_degenerate")
    assert [orc.retrieve_top_index(q, a["zero_corpus"]) for q in a["queries"]] == e["zero_corpus_top1"]
    assert [orc.retrieve_top_index(q, a["corpus"][:1]) for q in a["queries"]] == e["single_row_top1"]
    for q, row in zip(a["queries"], e["tiny"]):
        idx, val = orc.search(q.astype(np.float64), a["tiny_corpus"], 24)
        assert idx[0].tolist() == row["idx"]
        np.testing.assert_allclose(val[0], np.array(row["score"]), rtol=0, atol=1e-12)


def test_g3_near_ties(golden):
    a, e = golden("g3_near_ties")
    s = orc.scores(a["query"], a["corpus"])[0]
    np.testing.assert_allclose(s, np.array(e["all_scores"]), rtol=0, atol=1e-12)
    idx, val = orc.topk_from_scores(s, e["k"])
    assert idx[0].tolist() == e["topk_idx"]
    for p in e["pairs"]:
        gap = s[p["higher_score_row"]] - s[p["lower_score_row"]]
        assert 0.5 * p["gap"] < gap < 2.0 * p["gap"]


def test_g4_windows(golden):
    _, e = golden("g4_windows")
    pages = [{"index": i % 4, "text": f"t{i}", "embedding": [float(i), 1.0]} for i in range(e["n_pages"])]
    for c in e["cases"]:
        src = [] if c.get("empty_pages") else pages
        assert orc.get_surrounding_pages(c["target"], src, c["window"]) == c["result"]


def test_g5_error_conventions(golden, tmp_path):
    _, e = golden("g5_errors")
    logging.disable(logging.CRITICAL)
    try:
        d = 8
        eye = np.eye(4, d, dtype=np.float32)
        assert orc.retrieve_top_index(np.array([]), eye) == e["empty_query"]
        assert orc.retrieve_top_index(np.ones(d, dtype=np.float32), []) == e["empty_corpus"]
        assert orc.load_embeddings(str(tmp_path / "missing.json")) == e["missing_file"]
        bad = tmp_path / "bad.json"
        bad.write_text("{ not json")
        assert orc.load_embeddings(str(bad)) == e["bad_json"]
        nan = eye.copy(); nan[2, 3] = np.nan
        with pytest.raises(ValueError, match="NaN") as ei:
            orc.retrieve_top_index(np.ones(d, dtype=np.float32), nan)
        assert str(ei.value) == e["nan_row"]["message"]
        with pytest.raises(ValueError) as ei:
            orc.retrieve_top_index(np.array([1.0, np.nan] + [0.0] * (d - 2)), eye)
        assert str(ei.value) == e["nan_query"]["message"]
        inf = eye.copy(); inf[1, 0] = np.inf
        with pytest.raises(ValueError) as ei:
            orc.retrieve_top_index(np.ones(d, dtype=np.float32), inf)
        assert str(ei.value) == e["inf_row"]["message"]
    finally:
        logging.disable(logging.NOTSET)


def keyword_vector(kw, d):
    v = np.random.default_rng(zlib.crc32(kw.encode("utf-8"))).standard_normal(d).astype(np.float32)
    return v / np.float32(np.linalg.norm(v))


def test_g6_caller_glue(golden):
    a, e = golden("g6_caller_glue")
    cases = e["cases"]
    assert orc.unique_keywords(cases) == e["unique_keywords"]
    pages = [{"index": i % 1000, "text": f"page-{i}", "embedding": None} for i in range(a["corpus"].shape[0])]
    kw_map = {kw: keyword_vector(kw, e["d"]) for kw in e["unique_keywords"]}
    for case, exp in zip(cases, e["expected"]):
        got = orc.local_pages_for_case(case, kw_map, pages, corpus=a["corpus"])
        want = {kw: v["pages"] for kw, v in exp["local_pages_per_keyword"].items()}
        assert got == want


def test_merge_matches_global_topk():
    rng = np.random.default_rng(7)
    p = rng.standard_normal((300, 16)); q = rng.standard_normal((5, 16))
    p[17] = p[250]  # cross-shard duplicate -> tie resolved to the lower index
    full_i, full_v = orc.search(q, p, 12)
    parts = []
    for lo, hi in ((0, 100), (100, 200), (200, 300)):
        i, v = orc.search(q, p[lo:hi], 12)
        parts.append((np.where(i >= 0, i + lo, -1), v))
    # shard-local cosine == global cosine because normalisation is per row
    mi, mv = orc.merge_topk(parts, 12)
    assert mi.tolist() == full_i.tolist()
    np.testing.assert_allclose(mv, full_v, atol=1e-12)
