"""CPU-side checks of the boundary: the shared library loads, exports every
symbol the header declares, the ctypes table covers them all, and the host
logic that needs no GPU behaves as the reference's does."""
import ctypes
import logging
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "medtutor_retrieval.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mtr_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    from medical_rag_amd import _native
    names = declared_symbols()
    assert len(names) >= 12
    assert os.path.exists(_native.LIB_PATH), "build the library first (make -C medical-rag_amd/csrc)"
    handle = ctypes.CDLL(_native.LIB_PATH)
    for name in names:
        assert hasattr(handle, name), f"{name} declared in the header but not exported"
        assert name in _native.SIGNATURES, f"{name} has no ctypes signature"
    assert sorted(_native.SIGNATURES) == names
    assert b"gfx950" in _native.lib().mtr_version()


def test_error_codes_match_header():
    from medical_rag_amd import _native
    text = open(HEADER).read()
    for name in ("MTR_OK", "MTR_ERR_INVALID", "MTR_ERR_NONFINITE", "MTR_ERR_HIP", "MTR_ERR_COMM", "MTR_ERR_OOM"):
        val = int(re.search(name + r"\s*=\s*(-?\d+)", text).group(1))
        assert getattr(_native, name) == val
    assert int(re.search(r"#define MTR_MAX_K (\d+)", text).group(1)) == _native.MTR_MAX_K


def test_argument_errors_without_gpu():
    """Argument validation happens before any HIP call."""
    from medical_rag_amd import _native
    lib = _native.lib()
    out = ctypes.c_void_p()
    assert lib.mtr_index_alloc(0, 8, 0, 1, 0, 0, ctypes.byref(out)) == _native.MTR_ERR_INVALID
    assert "positive" in _native.last_error()
    assert lib.mtr_index_alloc(10, 8, 7, 1, 0, 0, ctypes.byref(out)) == _native.MTR_ERR_INVALID
    assert lib.mtr_index_search(None, None, 0, 0, 1, 1, 0, None, None) == _native.MTR_ERR_INVALID
    assert lib.mtr_index_info(None, None, None, None, None, None, None) == _native.MTR_ERR_INVALID
    lib.mtr_index_destroy(None)  # no-op


def test_facade_guards_and_windows_need_no_gpu(golden, tmp_path):
    from medical_rag_amd import local_rag as rag
    _, g4 = golden("g4_windows")
    pages = [{"index": i % 4, "text": f"t{i}", "embedding": [float(i), 1.0]} for i in range(g4["n_pages"])]
    for c in g4["cases"]:
        src = [] if c.get("empty_pages") else pages
        assert rag.get_surrounding_pages(c["target"], src, c["window"]) == c["result"]
    _, g5 = golden("g5_errors")
    assert rag.retrieve_top_index(np.array([]), pages) == g5["empty_query"]
    assert rag.retrieve_top_index(np.ones(2), []) == g5["empty_corpus"]
    assert rag.embed_queries([], "unused", "cpu") == g5["embed_no_queries"]
    logging.disable(logging.CRITICAL)
    try:
        assert rag.load_embeddings(str(tmp_path / "nope.json")) == g5["missing_file"]
        bad = tmp_path / "bad.json"
        bad.write_text("{ not json")
        assert rag.load_embeddings(str(bad)) == g5["bad_json"]
    finally:
        logging.disable(logging.NOTSET)


def test_no_cpu_scoring_fallback():
    """Without a GPU a real search must raise, not quietly score on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from medical_rag_amd import local_rag as rag
    with pytest.raises(RuntimeError):
        rag.retrieve_top_index(np.ones(4, dtype=np.float32), np.eye(4, dtype=np.float32))


def test_package_does_not_import_oracle():
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); import medical_rag_amd, medical_rag_amd.sharded; "
            "bad=[m for m in sys.modules if m.split('.')[0]=='oracle']; assert not bad, bad" % ROOT)
    subprocess.run([sys.executable, "-c", code], check=True)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "medical-rag_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f), encoding="utf-8").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
