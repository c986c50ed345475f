#!/usr/bin/env python3
"""Generates tests/golden/*.npz + *.json by running the reference's own
pipeline/utils/local_rag.py on seeded synthetic embeddings.

Runs only in the build container (needs /root/reference); the committed
outputs are data: inputs and the reference's answers.  Usage:
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
import json
import logging
import os
import sys
import tempfile

import numpy as np

REF = os.environ.get("MEDTUTOR_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(REF, "pipeline"))
from utils import local_rag as ref  # noqa: E402  (the reference module itself)
from sklearn.metrics.pairwise import cosine_similarity  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
logging.disable(logging.CRITICAL)


def unit_rows(rng, n, d):
    x = rng.standard_normal((n, d)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True).astype(np.float32)
    return x


def pages_from(mat, with_text=True):
    return [
        {"index": i % 1000, "text": f"page-{i}", "embedding": row.tolist()} if with_text else {"embedding": row.tolist()}
        for i, row in enumerate(mat)
    ]


def ref_topk(q, pages, k):
    """Reference scoring row (local_rag.py:76-77) + stable descending order;
    column 0 is asserted equal to the reference's own argmax."""
    mat = np.array([p["embedding"] for p in pages])
    sim = cosine_similarity(q.reshape(1, -1), mat).flatten()
    order = np.argsort(-sim, kind="stable")[:k]
    assert int(order[0]) == ref.retrieve_top_index(q, pages)
    return order.astype(np.int64), sim[order], sim


def save(name, arrays, expected):
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **arrays)
    with open(os.path.join(HERE, name + ".json"), "w") as f:
        json.dump(expected, f, indent=1)
    print("wrote", name)


def g1():
    rng = np.random.default_rng(0)
    corpus = unit_rows(rng, 4032, 64)
    queries = unit_rows(np.random.default_rng(1), 34, 64)
    pages = pages_from(corpus, with_text=False)
    idx, val, top1 = [], [], []
    for q in queries:
        i, v, _ = ref_topk(q, pages, 10)
        idx.append(i.tolist())
        val.append(v.tolist())
        top1.append(ref.retrieve_top_index(q, pages))
    save("g1_c1_miniature", {"corpus": corpus, "queries": queries},
         {"k": 10, "topk_idx": idx, "topk_score": val, "top1": top1})


def g2():
    """Ties and degenerate inputs."""
    rng = np.random.default_rng(2)
    d = 32
    base = rng.standard_normal((24, d)).astype(np.float32)
    corpus = base.copy()
    corpus[3] = corpus[1] * np.float32(4.0)      # same direction, other norm: exact cosine tie
    corpus[7] = corpus[1] * np.float32(0.25)
    corpus[9] = corpus[5]                        # exact duplicate
    corpus[12] = 0.0                             # zero row -> score 0
    corpus[13] = 0.0
    corpus[20] = -corpus[1]                      # anti-parallel
    queries = np.stack([
        base[1] * np.float32(3.0),               # un-normalised query hitting the tie set
        base[5],
        rng.standard_normal(d).astype(np.float32) * np.float32(100.0),
        -base[1],
    ])
    pages = pages_from(corpus, with_text=False)
    cases = {}
    for name, k in (("k5", 5), ("k_eq_n", 24), ("k_gt_n", 30)):
        idx, val = [], []
        for q in queries:
            i, v, _ = ref_topk(q, pages, k)
            idx.append(i.tolist())
            val.append(v.tolist())
        cases[name] = {"k": k, "topk_idx": idx, "topk_score": val}
    zero_corpus = np.zeros((6, d), dtype=np.float32)
    zero_top1 = [ref.retrieve_top_index(q, pages_from(zero_corpus, False)) for q in queries]
    single = corpus[:1]
    single_top1 = [ref.retrieve_top_index(q, pages_from(single, False)) for q in queries]
    tiny = corpus.copy().astype(np.float64)
    tiny[2] = tiny[2] / np.linalg.norm(tiny[2]) * 1e-15   # below sklearn's 10*eps cut-off: left unscaled
    tiny[4] = tiny[4] / np.linalg.norm(tiny[4]) * 1e-14   # above it: normalised
    tiny_pages = pages_from(tiny, False)
    tiny_rows = []
    for q in queries:
        i, v, sim = ref_topk(q.astype(np.float64), tiny_pages, 24)
        tiny_rows.append({"idx": i.tolist(), "score": v.tolist()})
    save("g2_ties_degenerate",
         {"corpus": corpus, "queries": queries, "zero_corpus": zero_corpus, "tiny_corpus": tiny},
         {"cases": cases, "zero_corpus_top1": zero_top1, "single_row_top1": single_top1, "tiny": tiny_rows})


def g3():
    """Near ties: for each gap g a pair of rows whose cosines to the query
    differ by about g, the better one at the HIGHER index."""
    rng = np.random.default_rng(3)
    d = 128
    gaps = [1e-2, 1e-3, 1e-4, 1e-6, 1e-8]
    q = unit_rows(rng, 1, d)[0].astype(np.float64)
    noise = rng.standard_normal((64, d))
    corpus = np.empty((64, d), dtype=np.float64)
    for i in range(64):
        n = noise[i] - (noise[i] @ q) * q
        n /= np.linalg.norm(n)
        c = 0.1 + 0.002 * i
        corpus[i] = c * q + np.sqrt(1 - c * c) * n
    pairs = []
    for j, g in enumerate(gaps):
        lo, hi = 2 * j, 2 * j + 1
        c = 0.9 - 0.05 * j
        for row, cc in ((lo, c), (hi, c + g)):
            n = noise[row] - (noise[row] @ q) * q
            n /= np.linalg.norm(n)
            corpus[row] = cc * q + np.sqrt(1 - cc * cc) * n
        pairs.append({"gap": g, "lower_score_row": lo, "higher_score_row": hi})
    pages = pages_from(corpus, False)
    i, v, sim = ref_topk(q, pages, 10)
    save("g3_near_ties", {"corpus": corpus, "query": q},
         {"k": 10, "topk_idx": i.tolist(), "topk_score": v.tolist(), "pairs": pairs,
          "all_scores": sim.tolist()})


def g4():
    pages = [{"index": i % 4, "text": f"t{i}", "embedding": [float(i), 1.0]} for i in range(7)]
    out = []
    for w in (0, 1, 2):
        for t in (-1, 0, 1, 5, 6, 7):
            out.append({"target": t, "window": w, "result": ref.get_surrounding_pages(t, pages, w)})
    out.append({"target": 0, "window": 1, "empty_pages": True, "result": ref.get_surrounding_pages(0, [], 1)})
    with open(os.path.join(HERE, "g4_windows.json"), "w") as f:
        json.dump({"n_pages": 7, "cases": out}, f, indent=1)
    print("wrote g4_windows")


def g5():
    d = 8
    pages = pages_from(np.eye(4, d, dtype=np.float32), False)
    res = {
        "empty_query": ref.retrieve_top_index(np.array([]), pages),
        "empty_corpus": ref.retrieve_top_index(np.ones(d, dtype=np.float32), []),
        "missing_file": ref.load_embeddings("/nonexistent/definitely_missing.json"),
        "embed_no_queries": ref.embed_queries([], "unused", "cpu"),
    }
    with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as f:
        f.write("{ not json")
        bad = f.name
    res["bad_json"] = ref.load_embeddings(bad)
    os.unlink(bad)
    bad_pages = pages_from(np.eye(4, d, dtype=np.float32), False)
    bad_pages[2]["embedding"][3] = float("nan")
    try:
        ref.retrieve_top_index(np.ones(d, dtype=np.float32), bad_pages)
        res["nan_row"] = "no error"
    except ValueError as e:
        res["nan_row"] = {"raises": "ValueError", "message": str(e)}
    try:
        ref.retrieve_top_index(np.array([1.0, np.nan] + [0.0] * (d - 2)), pages)
        res["nan_query"] = "no error"
    except ValueError as e:
        res["nan_query"] = {"raises": "ValueError", "message": str(e)}
    inf_pages = pages_from(np.eye(4, d, dtype=np.float32), False)
    inf_pages[1]["embedding"][0] = float("inf")
    try:
        ref.retrieve_top_index(np.ones(d, dtype=np.float32), inf_pages)
        res["inf_row"] = "no error"
    except ValueError as e:
        res["inf_row"] = {"raises": "ValueError", "message": str(e)}
    with open(os.path.join(HERE, "g5_errors.json"), "w") as f:
        json.dump(res, f, indent=1)
    print("wrote g5_errors")


def g6():
    """Caller glue on the reference's sample keyword file: keywords per case
    (data copied from pipeline/data/keywords_sample/), one seeded vector per
    unique stripped keyword, per-case loop as pipeline/utils/retrieval.py:70-76
    driven through the reference's two functions."""
    import zlib
    path = os.path.join(REF, "pipeline/data/keywords_sample/keyword_generator_output_sample.jsonl")
    cases = []
    with open(path, encoding="utf-8") as f:
        for line in f:
            if line.strip():
                rec = json.loads(line)
                cases.append({"case_id": rec["case_id"], "keywords": rec.get("keywords", [])})
    d, n = 96, 512
    corpus = unit_rows(np.random.default_rng(6), n, d)
    pages = pages_from(corpus, True)
    uniq = sorted({kw.strip() for c in cases for kw in c["keywords"] if isinstance(kw, str) and kw.strip()})
    kw_map = {}
    for kw in uniq:
        seed = zlib.crc32(kw.encode("utf-8"))
        v = np.random.default_rng(seed).standard_normal(d).astype(np.float32)
        kw_map[kw] = v / np.float32(np.linalg.norm(v))
    expected = []
    for c in cases:
        per_kw = {}
        for kw in c["keywords"]:
            emb = kw_map.get(kw)
            if emb is not None and emb.size > 0:
                top = ref.retrieve_top_index(emb, pages)
                per_kw[kw] = {"top_idx": top, "pages": ref.get_surrounding_pages(top, pages)}
        expected.append({"case_id": c["case_id"], "local_pages_per_keyword": per_kw})
    save("g6_caller_glue", {"corpus": corpus},
         {"cases": cases, "unique_keywords": uniq, "d": d, "seed_rule": "crc32(utf8(keyword)) -> default_rng -> standard_normal(d) f32, /norm",
          "expected": expected})


TINY_VOCAB = (["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"]
              + "lung edema pleural effusion cardiac pneumonia normal heart chest left right lobe opacity mass "
                "the of and with no is".split() + ["##s", "##ing", "##al"])
G7_QUERIES = ["pleural effusion", "normal heart", "left lung mass with edema", "pneumonia", "cardiac", "unknownword"]
G7_PAGES = [{"index": i % 3, "markdown": t} for i, t in enumerate([
    "the left lung is normal", "pleural effusion with opacity of the right lobe", "", "cardiac mass",
    "no pneumonia and no edema the heart is normal the chest is normal"])]


def g7():
    """Encoder side (a2, a6): a tiny random-weight BERT saved under
    tests/golden/tiny_encoder (so no download is needed), embedded with the
    reference's own embed_queries and generate_embeddings_hf."""
    import io
    import contextlib
    import torch
    from transformers import BertConfig, BertModel, BertTokenizer
    tiny = os.path.join(HERE, "tiny_encoder")
    if not os.path.exists(os.path.join(tiny, "model.safetensors")):
        os.makedirs(tiny, exist_ok=True)
        with open(os.path.join(tiny, "vocab.txt"), "w") as f:
            f.write("\n".join(TINY_VOCAB) + "\n")
        BertTokenizer(os.path.join(tiny, "vocab.txt")).save_pretrained(tiny)
        torch.manual_seed(0)
        cfg = BertConfig(vocab_size=len(TINY_VOCAB), hidden_size=32, num_hidden_layers=2, num_attention_heads=4,
                         intermediate_size=64, max_position_embeddings=64)
        BertModel(cfg).save_pretrained(tiny)
    q_all = ref.embed_queries(G7_QUERIES, tiny, "cpu")
    q_bs2 = ref.embed_queries(G7_QUERIES, tiny, "cpu", batch_size=2)
    sys.path.insert(0, os.path.join(REF, "pipeline", "utils"))
    import embed as ref_embed  # the reference's index builder
    with contextlib.redirect_stdout(io.StringIO()):
        pages = ref_embed.generate_embeddings_hf(G7_PAGES, tiny, "cpu", batch_size=2)
    save("g7_encoder", {"queries_bs64": np.stack(q_all), "queries_bs2": np.stack(q_bs2),
                        "page_vectors": np.array([p["embedding"] for p in pages], dtype=np.float64)},
         {"queries": G7_QUERIES, "pages_in": G7_PAGES,
          "pages_out": [{k: v for k, v in p.items() if k != "embedding"} for p in pages],
          "model_dir": "tests/golden/tiny_encoder"})


if __name__ == "__main__":
    g1(); g2(); g3(); g4(); g5(); g6(); g7()
