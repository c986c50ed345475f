"""numpy float64 restatement of the reference's local dense retriever.

TEST INFRASTRUCTURE ONLY -- never imported by the shipped package.

Parity status: PINNED.  Every function here is checked in
tests/test_oracle_golden.py against golden vectors that were produced by
importing and running the reference's own pipeline/utils/local_rag.py in the
build container (generator: tests/golden/make_golden.py; the reference source
itself never ships).  The reference repository holds no tests or known-answer
vectors of its own for this path.

Reference lines restated (paths relative to the reference checkout):
  pipeline/utils/local_rag.py:71-78   retrieve_top_index
  pipeline/utils/local_rag.py:80-88   get_surrounding_pages
  pipeline/utils/local_rag.py:90-101  load_embeddings
  pipeline/utils/retrieval.py:70-76   per-case keyword loop
  pipeline/async_main.py:187-195      keyword de-duplication + embedding map
Third-party arithmetic restated: scikit-learn (reference pins 1.6.1,
requirements.txt:186) sklearn.metrics.pairwise.cosine_similarity, dense path:
both operands validated finite and promoted to float64 unless both are
float32, each row divided by its L2 norm where norms below 10*eps count as 1
(sklearn.preprocessing.normalize + _handle_zeros_in_scale), then X @ Y.T.
"""
from __future__ import annotations

import json
import logging
from pathlib import Path

import numpy as np

_TINY = 10.0 * np.finfo(np.float64).eps  # sklearn _handle_zeros_in_scale cut-off


def _check_finite(a: np.ndarray) -> None:
    # sklearn check_array(force_all_finite=True) message, as raised through
    # cosine_similarity at local_rag.py:77.
    if not np.isfinite(a).all():
        if np.isnan(a).any():
            raise ValueError("Input contains NaN.")
        raise ValueError("Input contains infinity or a value too large for dtype('float64').")


def normalize_rows(x: np.ndarray) -> np.ndarray:
    """Row L2 normalisation in float64; norms < 10*eps are treated as 1."""
    x = np.array(x, dtype=np.float64, copy=True)
    if x.ndim == 1:
        x = x[None, :]
    norms = np.sqrt(np.einsum("ij,ij->i", x, x))
    norms[norms < _TINY] = 1.0
    x /= norms[:, None]
    return x


def as_matrix(page_embeddings) -> np.ndarray:
    """local_rag.py:76 -- np.array over the pages' "embedding" lists (float64),
    or an already-built (N, d) array."""
    if isinstance(page_embeddings, np.ndarray):
        return page_embeddings.astype(np.float64, copy=False)
    return np.array([p["embedding"] for p in page_embeddings], dtype=np.float64)


def scores(queries: np.ndarray, corpus: np.ndarray, metric: str = "cosine") -> np.ndarray:
    """(B, N) float64 score matrix.  metric "cosine" is local_rag.py:77;
    "ip" is the raw inner product (no normalisation on either side)."""
    q = np.asarray(queries, dtype=np.float64)
    if q.ndim == 1:
        q = q[None, :]
    p = np.asarray(corpus, dtype=np.float64)
    _check_finite(q)
    _check_finite(p)
    if metric == "cosine":
        q = normalize_rows(q)
        p = normalize_rows(p)
    elif metric != "ip":
        raise ValueError(f"unknown metric {metric!r}")
    return q @ p.T


def topk_from_scores(s: np.ndarray, k: int):
    """Stable descending top-k of each row: ties go to the lowest index, which
    is the only extension of np.argmax (local_rag.py:78) to k > 1.
    Returns (idx int64 (B, k), score float64 (B, k)); columns >= N are padded
    with idx -1 / score -inf."""
    s = np.atleast_2d(s)
    b, n = s.shape
    kk = min(k, n)
    order = np.argsort(-s, axis=1, kind="stable")[:, :kk]
    idx = np.full((b, k), -1, dtype=np.int64)
    val = np.full((b, k), -np.inf, dtype=np.float64)
    idx[:, :kk] = order
    val[:, :kk] = np.take_along_axis(s, order, axis=1)
    return idx, val


def search(queries, corpus, k: int, metric: str = "cosine"):
    return topk_from_scores(scores(queries, corpus, metric), k)


def retrieve_top_index(query_embedding: np.ndarray, page_embeddings) -> int:
    """local_rag.py:71-78 including the two guards at :73-74."""
    if query_embedding.size == 0 or len(page_embeddings) == 0:
        return 0
    sim = scores(query_embedding.reshape(1, -1), as_matrix(page_embeddings)).ravel()
    return int(np.argmax(sim))


def get_surrounding_pages(target_index: int, all_pages, window_size: int = 1) -> list:
    """local_rag.py:80-88."""
    if not all_pages or not (0 <= target_index < len(all_pages)):
        return []
    lo = max(0, target_index - window_size)
    hi = min(len(all_pages), target_index + window_size + 1)
    return [{k: v for k, v in page.items() if k != "embedding"} for page in all_pages[lo:hi]]


def load_embeddings(path: str) -> list:
    """local_rag.py:90-101: missing file or undecodable JSON -> [] + error log."""
    p = Path(path)
    if not p.exists():
        logging.error(f"Embedding file not found: {p}")
        return []
    try:
        with open(p, "r", encoding="utf-8") as f:
            return json.load(f)
    except json.JSONDecodeError:
        logging.error(f"Error decoding JSON from embedding file: {p}")
        return []


def merge_topk(parts, k: int):
    """k-way merge of per-shard candidate lists.  `parts` is a list of
    (idx int64 (B, k_i) global row ids, score (B, k_i)); padded entries have
    idx -1.  Order key: score descending, then index ascending."""
    idx = np.concatenate([p[0] for p in parts], axis=1)
    val = np.concatenate([p[1] for p in parts], axis=1).astype(np.float64)
    b = idx.shape[0]
    out_i = np.full((b, k), -1, dtype=np.int64)
    out_v = np.full((b, k), -np.inf, dtype=np.float64)
    for r in range(b):
        keep = idx[r] >= 0
        ii, vv = idx[r][keep], val[r][keep]
        order = np.lexsort((ii, -vv))[:k]
        out_i[r, : order.size] = ii[order]
        out_v[r, : order.size] = vv[order]
    return out_i, out_v


def unique_keywords(cases) -> list:
    """async_main.py:193 -- stripped, non-empty string keywords, de-duplicated.
    The reference iterates a set (arbitrary order); sorted here so the result
    is reproducible -- per-keyword results do not depend on the order."""
    return sorted(
        {kw.strip() for case in cases for kw in case.get("keywords", []) if isinstance(kw, str) and kw.strip()}
    )


def local_pages_for_case(case, kw_embeddings_map, pages, corpus=None) -> dict:
    """retrieval.py:70-76 -- raw (unstripped) keyword lookup, skip on miss or
    empty vector, top-1 row, +-1 page window."""
    out = {}
    kws = case.get("keywords", [])
    if not (len(pages) and kws):
        return out
    mat = as_matrix(pages) if corpus is None else corpus
    for kw in kws:
        emb = kw_embeddings_map.get(kw)
        if emb is not None and emb.size > 0:
            top = retrieve_top_index(emb, mat)
            out[kw] = get_surrounding_pages(top, pages)
    return out
