"""Binary index sidecar (host logic only)."""
import json
import os

import numpy as np
import pytest

from medical_rag_amd import index_format as fmt


def make_index(tmp_path, n=37, d=24, seed=0):
    rng = np.random.default_rng(seed)
    mat = rng.standard_normal((n, d)).astype(np.float32)
    pages = [{"index": i % 10, "text": f"page {i} é", "embedding": row.tolist()} for i, row in enumerate(mat)]
    path = tmp_path / "embedded_pages_hf_test.json"
    path.write_text(json.dumps(pages, indent=2), encoding="utf-8")
    return str(path), mat, pages


def test_convert_round_trip(tmp_path):
    path, mat, pages = make_index(tmp_path)
    assert fmt.open_sidecar(path) is None
    mtr, meta_path = fmt.convert(path)
    hdr = fmt.read_header(mtr)
    assert (hdr["n"], hdr["d"], hdr["dtype"]) == (37, 24, 0)
    assert hdr["source_size"] == os.path.getsize(path)
    got = fmt.open_sidecar(path)
    assert got is not None
    matrix, meta = got
    assert matrix.dtype == np.float32 and matrix.shape == mat.shape
    np.testing.assert_array_equal(np.asarray(matrix), mat)   # fp32 values survive the JSON round trip exactly
    assert meta == [{k: v for k, v in p.items() if k != "embedding"} for p in pages]


def test_stale_or_damaged_sidecar_is_ignored(tmp_path):
    path, mat, pages = make_index(tmp_path)
    mtr, meta_path = fmt.convert(path)
    # JSON rewritten afterwards -> stale
    pages[0]["text"] = "changed"
    with open(path, "w", encoding="utf-8") as f:
        json.dump(pages, f)
    os.utime(path, ns=(1, 1))
    assert fmt.open_sidecar(path) is None
    fmt.convert(path)
    assert fmt.open_sidecar(path) is not None
    # truncated matrix
    with open(mtr, "r+b") as f:
        f.truncate(os.path.getsize(mtr) - 4)
    assert fmt.open_sidecar(path) is None
    # wrong magic
    with open(mtr, "wb") as f:
        f.write(b"NOTANIDX" + b"\0" * 56)
    assert fmt.open_sidecar(path) is None
    with pytest.raises(ValueError):
        fmt.read_header(mtr)


def test_write_sidecar_without_source(tmp_path):
    mat = np.arange(12, dtype=np.float32).reshape(3, 4)
    meta = [{"index": i, "text": str(i)} for i in range(3)]
    base = str(tmp_path / "built_on_gpu.json")
    fmt.write_sidecar(base, mat, meta)
    matrix, got_meta = fmt.open_sidecar(base)
    np.testing.assert_array_equal(np.asarray(matrix), mat)
    assert got_meta == meta
    with pytest.raises(ValueError):
        fmt.write_sidecar(base, mat, meta[:2])


def test_bf16_sidecar_round_trip(tmp_path):
    """A uint16 matrix is stored as bfloat16 bits with the as-stored flag and comes back as it went in."""
    import json
    import numpy as np
    from medical_rag_amd import index_format
    rng = np.random.default_rng(5)
    bits = (rng.standard_normal((37, 24)).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
    meta = [{"index": i, "text": f"p{i}"} for i in range(37)]
    path = str(tmp_path / "embedded_pages_hf_y.json")
    mtr, _ = index_format.write_sidecar(path, bits, meta, bind_to_source=False, as_stored=True)
    hdr = index_format.read_header(mtr)
    assert (hdr["n"], hdr["d"], hdr["dtype"], hdr["as_stored"]) == (37, 24, index_format.DTYPE_BF16, True)
    matrix, got_meta, hdr2 = index_format.open_sidecar(path, with_header=True)
    assert matrix.dtype == np.uint16 and (np.asarray(matrix) == bits).all() and got_meta == meta and hdr2 == hdr
    assert os.path.getsize(mtr) == index_format.HEADER.size + bits.size * 2
    # fp32 sidecars keep reading as before, flag clear
    index_format.write_sidecar(path, bits.astype(np.float32), meta, bind_to_source=False)
    assert index_format.read_header(mtr)["as_stored"] is False and index_format.open_sidecar(path)[0].dtype == np.float32
