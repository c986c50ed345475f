"""Batched form of the reference's per-case retrieval loop.

The reference runs one full scan per (case, keyword)
(pipeline/utils/retrieval.py:70-76) although every unique keyword vector exists
before stage 1 starts (pipeline/async_main.py:193-195).  Here all distinct
keyword vectors go through ONE search call; the per-case loop becomes a
dictionary lookup with exactly the reference's skipping rules: keywords are
looked up unstripped, and a missing or empty vector skips the keyword.
"""
from __future__ import annotations

import numpy as np

from .local_rag import _SETTINGS, get_surrounding_pages, search


def unique_keywords(cases) -> list:
    """pipeline/async_main.py:193 -- stripped, non-empty string keywords of all
    cases, de-duplicated (sorted for reproducibility; the reference iterates a set)."""
    return sorted({kw.strip() for case in cases for kw in case.get("original_keywords", case.get("keywords", []))
                   if isinstance(kw, str) and kw.strip()})


def top_pages_for_keywords(kw_embeddings_map: dict, pages, top_k: int | None = None) -> dict:
    """One batched search for every keyword that has a non-empty vector:
    keyword -> (idx (top_k,), score (top_k,)).  top_k None: the configured
    local_top_k (default 1)."""
    if top_k is None:
        top_k = _SETTINGS["top_k"]
    usable = [(kw, np.asarray(v)) for kw, v in kw_embeddings_map.items() if v is not None and np.asarray(v).size > 0]
    if not usable or len(pages) == 0:
        return {}
    idx, score = search(pages, np.stack([v.reshape(-1) for _, v in usable]), k=top_k)
    return {kw: (idx[i], score[i]) for i, (kw, _) in enumerate(usable)}


def top_pages_for_texts(keywords: list, model_name: str, device: str, pages, top_k: int = 1, batch_size: int = 64) -> dict:
    """Encoder and retriever on the same GPU: embeds the keyword texts
    (pipeline/async_main.py:195), scores them while still in HBM and brings
    back only the hits.  keyword -> (idx (top_k,), score (top_k,)); {} when
    there is nothing to embed, the model cannot load or there are no pages."""
    from .local_rag import embed_queries_device
    if not keywords or len(pages) == 0:
        return {}
    vectors = embed_queries_device(list(keywords), model_name, device, batch_size)
    if vectors is None:
        return {}
    idx, score = search(pages, vectors, k=top_k)
    idx, score = idx.cpu().numpy(), score.cpu().numpy()
    # the device path cannot fail a call it does not wait for; after the copy above it can be asked
    shard = getattr(pages, "shard", None)
    if shard is not None and shard.nonfinite():
        raise ValueError("Input contains NaN or infinity.")   # what sklearn raises at pipeline/utils/local_rag.py:77
    return {kw: (idx[i], score[i]) for i, kw in enumerate(keywords)}


def local_pages_for_cases(cases, kw_embeddings_map: dict, pages, window_size: int | None = None,
                          hits: dict | None = None) -> list:
    """For every case the dict the reference stores as "local_pages_per_keyword"
    (pipeline/utils/retrieval.py:70-76, 82): keyword -> hit page with its
    neighbours, vectors stripped.  window_size None: the configured window
    (local_window, default 1 as in the reference)."""
    if window_size is None:
        window_size = _SETTINGS["window"]
    if hits is None:
        hits = top_pages_for_keywords(kw_embeddings_map, pages, top_k=1)
    out = []
    for case in cases:
        per_kw = {}
        kws = case.get("original_keywords", case.get("keywords", []))
        if len(pages) and kws:
            for kw in kws:
                hit = hits.get(kw)
                if hit is not None:
                    per_kw[kw] = get_surrounding_pages(int(hit[0][0]), pages, window_size)
        out.append(per_kw)
    return out
