"""Binary sidecar for the page-embedding index.

The reference keeps the index as one JSON list of {"index", "text",
"embedding": [d floats]} (written at pipeline/utils/embed.py:74-79,110-113,
parsed whole by pipeline/utils/local_rag.py:90-101): 470 MB of text and ten
seconds of json.load for 4032 x 4096, and no path to 10^6+ rows.  The sidecar
stores the same rows as raw little-endian fp32 behind a fixed header, plus the
page dicts (everything but the vectors) as JSON; the matrix is memory-mapped
and streamed to the GPU in chunks.

    <index>.json          the reference's file (source of truth)
    <index>.json.mtr      header (64 bytes) + n*d float32, C order
    <index>.json.pages    JSON list of the page dicts without "embedding"

Header: magic "MTRIDX1\\0", u64 n, u32 d, u32 dtype (0 = fp32, 2 = bfloat16
bits), u64 size and i64 mtime_ns of the JSON the sidecar was made from (0/0
when built without one), u32 flags (bit 0: the rows are stored as a shard
holds them, L2-normalised and rounded, and load without a second
normalisation), 20 bytes reserved.  A sidecar whose recorded size/mtime do not
match its JSON is stale and ignored.

A bf16 sidecar is half the bytes on disk and over PCIe (100M x 768: 154 GB
instead of 307 GB); it is written from a device shard (the loader's own
normalisation and rounding), so loading it reproduces that shard bit for bit.
"""
from __future__ import annotations

import json
import os
import struct

import numpy as np

MAGIC = b"MTRIDX1\0"
HEADER = struct.Struct("<8sQIIQqI20x")
DTYPE_F32, DTYPE_BF16 = 0, 2
FLAG_AS_STORED = 1
assert HEADER.size == 64


def sidecar_paths(json_path: str):
    return json_path + ".mtr", json_path + ".pages"


def write_sidecar(json_path: str, matrix: np.ndarray, pages_meta: list, bind_to_source: bool = True,
                  as_stored: bool = False):
    """Writes <json_path>.mtr / .pages for an (n, d) matrix and its page dicts.
    float matrices are stored as fp32; a uint16 matrix is bfloat16 bits (as
    DeviceShard.read_rows_bf16 returns them).  as_stored: the rows are a shard's
    stored rows and must not be normalised again on load."""
    if isinstance(matrix, np.ndarray) and matrix.dtype == np.uint16:
        matrix, dtype = np.ascontiguousarray(matrix), DTYPE_BF16
    else:
        matrix, dtype = np.ascontiguousarray(matrix, dtype=np.float32), DTYPE_F32
    if matrix.ndim != 2 or len(pages_meta) != matrix.shape[0]:
        raise ValueError("matrix must be (n, d) with one page dict per row")
    size = mtime = 0
    if bind_to_source and os.path.exists(json_path):
        st = os.stat(json_path)
        size, mtime = st.st_size, st.st_mtime_ns
    mtr, pages = sidecar_paths(json_path)
    tmp = mtr + ".tmp"
    with open(tmp, "wb") as f:
        f.write(HEADER.pack(MAGIC, matrix.shape[0], matrix.shape[1], dtype, size, mtime,
                            FLAG_AS_STORED if as_stored else 0))
        matrix.tofile(f)
    os.replace(tmp, mtr)
    tmp = pages + ".tmp"
    with open(tmp, "w", encoding="utf-8") as f:
        json.dump(pages_meta, f)
    os.replace(tmp, pages)
    return mtr, pages


def read_json_index(json_path: str, dtype=np.float64):
    """(matrix (n, d), page dicts without "embedding") of the reference's JSON
    index through the library's reader -- the numbers go from the file's text
    straight into the matrix, the page dicts come from the json module applied
    to the few KB a page that are left -- or None when that reader declines the
    file or the library is not built; json.load then decides what the file is.
    Same values as numpy.array([p["embedding"] ...]) over json.load's result
    (pipeline/utils/local_rag.py:96-98), an order of magnitude sooner."""
    from . import _native
    try:
        got = _native.read_index_json(json_path, dtype=dtype)
    except _native.NativeLibraryError:
        return None
    if got is None:
        return None
    matrix, rest = got
    meta = json.loads(rest.decode("utf-8"))     # raises what json.load would for bad text outside the vectors
    if not isinstance(meta, list) or len(meta) != matrix.shape[0]:
        return None
    for page in meta:
        del page["embedding"]
    return matrix, meta


def convert(json_path: str, store_dtype: str = "fp32", device: int | None = None):
    """Parses the reference's JSON index once and writes its sidecar.
    store_dtype "bf16": the rows go through a bf16 shard on the GPU first and
    the sidecar holds exactly what that shard stores (needs the GPU)."""
    fast = read_json_index(json_path, dtype=np.float32)
    if fast is not None:
        matrix, meta = fast
        if not meta:
            raise ValueError("empty index")
    else:
        with open(json_path, "r", encoding="utf-8") as f:
            pages = json.load(f)
        if not pages:
            raise ValueError("empty index")
        matrix = np.array([p["embedding"] for p in pages], dtype=np.float32)
        meta = [{k: v for k, v in p.items() if k != "embedding"} for p in pages]
    if store_dtype in ("bf16", "bfloat16"):
        from . import _native
        shard = _native.DeviceShard.from_host(matrix, store_dtype="bf16", normalize=True, device=device or 0)
        try:
            return write_sidecar(json_path, shard.read_rows_bf16(), meta, as_stored=True)
        finally:
            shard.close()
    return write_sidecar(json_path, matrix, meta)


def read_header(mtr_path: str):
    with open(mtr_path, "rb") as f:
        raw = f.read(HEADER.size)
    if len(raw) != HEADER.size:
        raise ValueError("truncated sidecar header")
    magic, n, d, dtype, size, mtime, flags = HEADER.unpack(raw)
    if magic != MAGIC:
        raise ValueError("not an MTRIDX1 sidecar")
    if dtype not in (DTYPE_F32, DTYPE_BF16):
        raise ValueError(f"unsupported sidecar dtype {dtype}")
    return {"n": n, "d": d, "dtype": dtype, "source_size": size, "source_mtime_ns": mtime,
            "as_stored": bool(flags & FLAG_AS_STORED)}


def open_sidecar(json_path: str, with_header: bool = False):
    """(memory-mapped (n, d) matrix, page dicts) when a fresh sidecar exists for
    json_path, else None.  The matrix is float32, or uint16 bfloat16 bits for a
    bf16 sidecar; with_header adds the parsed header as a third element."""
    mtr, pages = sidecar_paths(json_path)
    if not (os.path.exists(mtr) and os.path.exists(pages)):
        return None
    try:
        hdr = read_header(mtr)
    except (OSError, ValueError):
        return None
    if os.path.exists(json_path):
        st = os.stat(json_path)
        if (hdr["source_size"], hdr["source_mtime_ns"]) not in ((st.st_size, st.st_mtime_ns), (0, 0)):
            return None  # the JSON changed after the sidecar was written
    esize = 2 if hdr["dtype"] == DTYPE_BF16 else 4
    expect = HEADER.size + hdr["n"] * hdr["d"] * esize
    if os.path.getsize(mtr) != expect:
        return None
    matrix = np.memmap(mtr, dtype="<u2" if esize == 2 else "<f4", mode="r", offset=HEADER.size,
                       shape=(hdr["n"], hdr["d"]))
    with open(pages, "r", encoding="utf-8") as f:
        meta = json.load(f)
    if len(meta) != hdr["n"]:
        return None
    return (matrix, meta, hdr) if with_header else (matrix, meta)
