#!/usr/bin/env python3
"""Times the reference's own retrieve_top_index on config C1 in the build
container (needs /root/reference; never runs on the GPU box).

C1: N = 4032 pages (the shipped OCR splits), d = 4096 (Qwen3-Embedding-8B), the
34 unique keywords of the sample keyword file, synthetic unit-norm vectors
serialised as pipeline/utils/embed.py:74-79 does (Python lists of floats).
Environment: C1_DIM (4096; 3584 is the reference's fallback encoder,
pipeline/utils/embed.py:91), C1_TAG (prefix of the output file, default r02);
BLAS threads follow OMP_NUM_THREADS / OPENBLAS_NUM_THREADS as set by the caller
and are recorded.  Writes profiles/<tag>_c1_reference_cpu_d<dim>[_t<threads>].json.
"""
import json
import os
import sys
import time

import numpy as np

REF = os.environ.get("MEDTUTOR_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(REF, "pipeline"))
from utils import local_rag as ref  # noqa: E402
from sklearn.metrics.pairwise import cosine_similarity  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    n, d, b = 4032, int(os.environ.get("C1_DIM", 4096)), 34
    rng = np.random.default_rng(0)
    mat = rng.standard_normal((n, d)).astype(np.float32)
    mat /= np.linalg.norm(mat, axis=1, keepdims=True)
    pages = [{"index": i % 1000, "text": "", "embedding": row.tolist()} for i, row in enumerate(mat)]
    queries = rng.standard_normal((b, d)).astype(np.float32)
    queries /= np.linalg.norm(queries, axis=1, keepdims=True)
    ref.retrieve_top_index(queries[0], pages)  # warm-up
    t0 = time.perf_counter()
    tops = [ref.retrieve_top_index(q, pages) for q in queries]
    as_written = (time.perf_counter() - t0) / b
    cached = np.array([p["embedding"] for p in pages])
    t0 = time.perf_counter()
    tops2 = [int(np.argmax(cosine_similarity(q.reshape(1, -1), cached).flatten())) for q in queries]
    scoring_only = (time.perf_counter() - t0) / b
    assert tops == tops2
    out = {
        "config": f"C1: N={n}, d={d}, {b} queries, top-1 (reference as written)",
        "host": f"{os.cpu_count()} vCPU build container, numpy {np.__version__}",
        "blas_threads": os.environ.get("OMP_NUM_THREADS") or os.environ.get("OPENBLAS_NUM_THREADS") or "default (all cores)",
        "retrieve_top_index_ms_per_query": as_written * 1e3,
        "queries_per_s_as_written": 1.0 / as_written,
        "cosine_similarity_only_ms_per_query": scoring_only * 1e3,
        "queries_per_s_matrix_cached": 1.0 / scoring_only,
        "top1": tops,
    }
    threads = os.environ.get("OMP_NUM_THREADS") or os.environ.get("OPENBLAS_NUM_THREADS")
    tag = os.environ.get("C1_TAG", "r02")
    path = os.path.join(ROOT, "profiles", f"{tag}_c1_reference_cpu_d{d}" + (f"_t{threads}" if threads else "") + ".json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "top1"}, indent=1))


if __name__ == "__main__":
    main()
