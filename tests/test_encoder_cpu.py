"""Encoder side of the path (embed_queries, index builder) on CPU against
vectors the reference itself produced with the committed tiny encoder
(tests/golden/tiny_encoder, fixture g7)."""
import json
import logging
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TINY = os.path.join(ROOT, "tests", "golden", "tiny_encoder")


def test_embed_queries_matches_reference(golden):
    from medical_rag_amd import local_rag as rag
    a, e = golden("g7_encoder")
    got = rag.embed_queries(e["queries"], TINY, "cpu")
    assert len(got) == len(e["queries"]) and all(v.dtype == np.float32 and v.shape == (32,) for v in got)
    np.testing.assert_allclose(np.stack(got), a["queries_bs64"], atol=1e-6, rtol=0)
    got2 = rag.embed_queries(e["queries"], TINY, "cpu", batch_size=2)
    np.testing.assert_allclose(np.stack(got2), a["queries_bs2"], atol=1e-6, rtol=0)


def test_embed_queries_device_is_the_same_vectors_as_one_tensor(golden):
    from medical_rag_amd import local_rag as rag
    a, e = golden("g7_encoder")
    t = rag.embed_queries_device(e["queries"], TINY, "cpu", batch_size=2)
    assert tuple(t.shape) == a["queries_bs2"].shape and t.is_contiguous()
    np.testing.assert_allclose(t.numpy(), a["queries_bs2"], atol=1e-6, rtol=0)
    assert rag.embed_queries_device([], TINY, "cpu") is None


def test_embed_queries_soft_fail_on_bad_model():
    from medical_rag_amd import local_rag as rag
    logging.disable(logging.CRITICAL)
    try:
        out = rag.embed_queries(["a", "b"], "/nonexistent/model/dir", "cpu")
    finally:
        logging.disable(logging.NOTSET)
    assert len(out) == 2 and all(v.size == 0 for v in out)  # local_rag.py:45-46


def test_index_builder_matches_reference(golden, tmp_path):
    from medical_rag_amd import index_builder, index_format
    from oracle import local_rag_oracle as orc
    a, e = golden("g7_encoder")
    failed = []
    pages = index_builder.generate_embeddings_hf(e["pages_in"], TINY, "cpu", batch_size=2, failed=failed)
    assert failed == []
    assert [{k: v for k, v in p.items() if k != "embedding"} for p in pages] == e["pages_out"]
    np.testing.assert_allclose(np.array([p["embedding"] for p in pages]), a["page_vectors"], atol=1e-6, rtol=0)
    path = index_builder.index_file_name(str(tmp_path), "org/model-x")
    assert os.path.basename(path) == "embedded_pages_hf_org_model-x.json"
    index_builder.write_index(pages, path)
    # the JSON is the reference's schema (readable by its loader) and the sidecar is fresh
    assert orc.load_embeddings(path) == json.load(open(path))
    matrix, meta = index_format.open_sidecar(path)
    np.testing.assert_array_equal(np.asarray(matrix), np.array([p["embedding"] for p in pages], dtype=np.float32))
    assert meta == e["pages_out"]


def test_index_builder_reports_failed_batches(tmp_path):
    from medical_rag_amd import index_builder
    pages = [{"index": 0, "markdown": "lung"}, {"index": 1, "markdown": 5}, {"index": 2, "markdown": "heart"}]
    failed = []
    logging.disable(logging.CRITICAL)
    try:
        out = index_builder.generate_embeddings_hf(pages, TINY, "cpu", batch_size=1, failed=failed)
    finally:
        logging.disable(logging.NOTSET)
    assert failed == [1] and [p["index"] for p in out] == [0, 2]


def test_load_all_pages_sorted(tmp_path):
    from medical_rag_amd import index_builder
    for name, idx in (("split_2.json", [3]), ("split_1.json", [1, 2])):
        (tmp_path / name).write_text(json.dumps({"pages": [{"index": i, "markdown": str(i)} for i in idx]}))
    pages = index_builder.load_all_pages(str(tmp_path / "split_*.json"))
    assert [p["index"] for p in pages] == [1, 2, 3]
    logging.disable(logging.CRITICAL)
    try:
        assert index_builder.load_all_pages(str(tmp_path / "nothing_*.json")) == []
    finally:
        logging.disable(logging.NOTSET)
