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


@pytest.mark.parametrize("normalized", [True, False])
def test_save_sidecar_marks_only_normalised_rows_as_stored(tmp_path, normalized):
    """Rows of a shard built without normalisation must not skip it on the way back in."""
    from medical_rag_amd import local_rag

    rows = np.arange(12, dtype=np.float32).reshape(3, 4) + 1.0

    class StubShard:
        def info(self):
            return {"store_dtype": "fp32", "normalized": normalized, "n": 3, "d": 4, "device": 0, "row_offset": 0}

        def read_rows(self):
            return rows

    index = local_rag.PageIndex([{"index": i, "text": str(i)} for i in range(3)], StubShard())
    path = str(tmp_path / "embedded_pages_hf_z.json")
    mtr, _ = index.save_sidecar(path, bind_to_source=False)
    assert fmt.read_header(mtr)["as_stored"] is normalized
    np.testing.assert_array_equal(np.asarray(fmt.open_sidecar(path)[0]), rows)


# ---------------------------------------------------------------------------
# The library's reader of the reference's JSON index (host code, no GPU).
# ---------------------------------------------------------------------------
def _by_json_module(path):
    with open(path, "r", encoding="utf-8") as f:
        pages = json.load(f)
    matrix = np.array([p["embedding"] for p in pages], dtype=np.float64)
    return matrix, [{k: v for k, v in p.items() if k != "embedding"} for p in pages]


def test_json_reader_equals_json_module(tmp_path):
    path, mat, pages = make_index(tmp_path, n=53, d=40, seed=3)
    got = fmt.read_json_index(path)
    assert got is not None
    want_matrix, want_meta = _by_json_module(path)
    assert got[0].dtype == np.float64 and np.array_equal(got[0], want_matrix)
    assert got[1] == want_meta
    as_f32 = fmt.read_json_index(path, dtype=np.float32)
    assert as_f32[0].dtype == np.float32 and np.array_equal(as_f32[0], mat)


def test_json_reader_awkward_but_valid_files(tmp_path):
    rng = np.random.default_rng(5)
    pages = []
    for i in range(17):
        pages.append({
            "text": 'quote " bracket ] [ brace } { backslash \\ \\" tab \t nl \n é   "embedding": [1, 2]',
            "nested": {"embedding": ["not", "numbers"], "list": [1, [2, {"a": None}], "]"], "t": True, "f": False},
            "embedding": rng.standard_normal(6).tolist(),
            "index": i,
            "tail\\": "\\",
        })
    for name, kwargs in (("compact", {"separators": (",", ":")}), ("pretty", {"indent": 4}), ("ascii", {"ensure_ascii": True}),
                         ("utf8", {"ensure_ascii": False})):
        path = tmp_path / f"{name}.json"
        path.write_text(json.dumps(pages, **kwargs), encoding="utf-8")
        got = fmt.read_json_index(str(path))
        assert got is not None, name
        want_matrix, want_meta = _by_json_module(path)
        assert np.array_equal(got[0], want_matrix) and got[1] == want_meta, name
    # key order: the vector first, last, and alone
    for j, page in enumerate(({"embedding": [1.5, -2], "index": 0}, {"index": 0, "embedding": [1.5, -2]}, {"embedding": [1.5, -2]})):
        path = tmp_path / f"order{j}.json"
        path.write_text(" \n" + json.dumps([page, page]) + "\n\t ", encoding="utf-8")
        got = fmt.read_json_index(str(path))
        assert np.array_equal(got[0], [[1.5, -2.0], [1.5, -2.0]])
        assert got[1] == [{k: v for k, v in page.items() if k != "embedding"}] * 2


def test_json_reader_numbers_follow_python(tmp_path):
    """Every spelling json.dump can produce, and the grammar's corners, give the
    double float() gives (correct rounding on both sides)."""
    rng = np.random.default_rng(11)
    texts = ["0", "-0", "0.0", "-0.0", "1", "-1", "10", "123456789012345678901234567890", "0.1", "0.30000000000000004",
             "1e0", "1E5", "1e+5", "1e-5", "2.5e-324", "5e-324", "1.7976931348623157e308", "1e309", "-1e309", "1e-400",
             "9007199254740993", "9007199254740992", "0.000001", "123.456e-7", "1.0000000000000002", "4.35", "0.5e1",
             "8.5e22", "8.5e23", "1e22", "1e23", "72057594037927933", "0.00000000000000000000000000000000000001"]
    for _ in range(4000):
        kind = rng.integers(0, 4)
        if kind == 0:
            texts.append(repr(float(np.float32(rng.standard_normal()))))          # what embed.py's tolist() writes
        elif kind == 1:
            texts.append(repr(float(rng.standard_normal() * 10.0 ** rng.integers(-30, 30))))
        elif kind == 2:
            digits = "".join(str(x) for x in rng.integers(0, 10, rng.integers(1, 25)))
            texts.append(f"{rng.integers(0, 1000)}.{digits}e{rng.integers(-40, 40)}")
        else:
            texts.append(str(int(rng.integers(-10 ** 17, 10 ** 17))))
    path = tmp_path / "numbers.json"
    path.write_text('[{"embedding": [' + ", ".join(texts) + "]}]", encoding="utf-8")
    got = fmt.read_json_index(str(path))
    assert got is not None
    want = np.array([float(t) for t in texts])
    assert got[0].shape == (1, len(texts))
    assert np.array_equal(got[0][0].view(np.uint64), want.view(np.uint64))      # bit for bit, signed zeros included
    # Python's three extra literals
    path.write_text('[{"embedding": [NaN, Infinity, -Infinity, 1]}]', encoding="utf-8")
    got = fmt.read_json_index(str(path))
    assert np.isnan(got[0][0, 0]) and got[0][0, 1] == np.inf and got[0][0, 2] == -np.inf and got[0][0, 3] == 1.0


@pytest.mark.parametrize("text", [
    "", "   ", "{}", '{"embedding": [1]}', "[1, 2]", '[{"index": 1}]', "[{}]",
    '[{"embedding": []}]', '[{"embedding": null}]', '[{"embedding": 1.5}]', '[{"embedding": "x"}]',
    '[{"embedding": [1, 2]}, {"embedding": [1]}]', '[{"embedding": [1]}, {"embedding": [1, 2]}]',
    '[{"embedding": [1, "2"]}]', '[{"embedding": [1, [2]]}]', '[{"embedding": [1, null]}]',
    '[{"embedding": [1,]}]', '[{"embedding": [,1]}]', '[{"embedding": [1 2]}]', '[{"embedding": [1,,2]}]',
    '[{"embedding": [01]}]', '[{"embedding": [+1]}]', '[{"embedding": [.5]}]', '[{"embedding": [1.]}]',
    '[{"embedding": [1e]}]', '[{"embedding": [0x10]}]', '[{"embedding": [inf]}]', '[{"embedding": [nan]}]',
    '[{"embedding": [-NaN]}]', '[{"embedding": [1.5x]}]', '[{"embedding": [--1]}]', '[{"embedding": [1e+]}]',
    '[{"embedding": [1], "embedding": [2]}]', '[{"embedding": [1]}] x', '[{"embedding": [1]},]',
    '[{"embedding": [1]}', '[{"embedding": [1]', '[{"embedding": [1}]', '[{"embedding" [1]}]', '[{"text": "open}]',
    '﻿[{"embedding": [1]}]',
])
def test_json_reader_declines_what_it_does_not_know(tmp_path, text):
    """Anything but a list of page objects with one flat numeric vector each, all
    of one length: the reader steps aside, and the json module's verdict (a
    value or JSONDecodeError) stands for the file."""
    path = tmp_path / "odd.json"
    path.write_text(text, encoding="utf-8")
    assert fmt.read_json_index(str(path)) is None


def test_json_reader_bad_text_outside_the_vectors_raises_like_json(tmp_path):
    path = tmp_path / "bad.json"
    path.write_text('[{"index": tru, "embedding": [1, 2]}]', encoding="utf-8")
    with pytest.raises(json.JSONDecodeError):
        fmt.read_json_index(str(path))
    path.write_text('[{"text": "raw \n newline", "embedding": [1, 2]}]', encoding="utf-8")
    with pytest.raises(json.JSONDecodeError):
        fmt.read_json_index(str(path))


def test_json_reader_empty_list_and_threads(tmp_path):
    from medical_rag_amd import _native
    path = tmp_path / "empty.json"
    path.write_text(" [ ] ", encoding="utf-8")
    matrix, meta = fmt.read_json_index(str(path))
    assert matrix.shape == (0, 0) and meta == []
    path, mat, pages = make_index(tmp_path, n=9, d=5)
    want = _by_json_module(path)[0]
    for threads in (1, 2, 3, 9, 64):
        matrix, rest = _native.read_index_json(path, threads=threads)
        assert np.array_equal(matrix, want)
