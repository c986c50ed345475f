"""Wall time from the reference's JSON index file to a searchable index, at the
reference's own size (4032 pages x 4096, the file pipeline/utils/embed.py:110-113
writes): json.load + numpy.array as pipeline/utils/local_rag.py:96-98 and :76 do,
the library's reader, load_embeddings end to end (file -> HBM), and the binary
sidecar.  One JSON line on stdout."""
import json
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from medical_rag_amd import index_format, local_rag  # noqa: E402


def main():
    n = int(os.environ.get("JSON_LOAD_N", 4032))
    d = int(os.environ.get("JSON_LOAD_D", 4096))
    rng = np.random.default_rng(0)
    mat = rng.standard_normal((n, d)).astype(np.float32)
    mat /= np.linalg.norm(mat, axis=1, keepdims=True)
    tmp = tempfile.mkdtemp(prefix="json_load_")
    path = os.path.join(tmp, "embedded_pages_hf_synthetic.json")
    t = time.perf_counter()
    with open(path, "w", encoding="utf-8") as f:
        json.dump([{"index": i, "text": "page %d " % i * 200, "embedding": row.tolist()} for i, row in enumerate(mat)], f, indent=2)
    out = {"n": n, "d": d, "file_MB": round(os.path.getsize(path) / 1e6, 1), "write_s": round(time.perf_counter() - t, 2)}
    print(out, file=sys.stderr, flush=True)

    t = time.perf_counter()
    with open(path, "r", encoding="utf-8") as f:
        pages = json.load(f)
    out["json_load_s"] = round(time.perf_counter() - t, 3)
    t = time.perf_counter()
    want = np.array([p["embedding"] for p in pages], dtype=np.float64)
    out["numpy_array_s"] = round(time.perf_counter() - t, 3)
    del pages

    for threads in (1, 4, 16):
        from medical_rag_amd import _native
        t = time.perf_counter()
        got, rest = _native.read_index_json(path, threads=threads)
        out[f"reader_{threads}_threads_s"] = round(time.perf_counter() - t, 3)
        assert np.array_equal(got, want)
    t = time.perf_counter()
    got, meta = index_format.read_json_index(path)
    out["reader_with_page_dicts_s"] = round(time.perf_counter() - t, 3)
    assert np.array_equal(got, want) and len(meta) == n

    import torch
    torch.cuda.init()
    q = mat[[5, 1000, n - 1]]
    for label in ("json_to_hbm_first_s", "json_to_hbm_s"):
        t = time.perf_counter()
        index = local_rag.load_embeddings(path)
        idx, _ = index.search(q, k=1)
        out[label] = round(time.perf_counter() - t, 3)
        assert idx[:, 0].tolist() == [5, 1000, n - 1]
        index.close()
    index_format.convert(path)
    t = time.perf_counter()
    index = local_rag.load_embeddings(path)
    idx, _ = index.search(q, k=1)
    out["sidecar_to_hbm_s"] = round(time.perf_counter() - t, 3)
    assert idx[:, 0].tolist() == [5, 1000, n - 1]
    index.close()
    for name in os.listdir(tmp):
        os.remove(os.path.join(tmp, name))
    os.rmdir(tmp)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
