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


def test_sanitizer_build_of_the_abi_layer():
    """make asan: the two host-logic files of the library compiled with
    AddressSanitizer (host side only) and driven by csrc/abi_check.cpp through
    every entry point's argument checks (plus a small search when a GPU is there)."""
    import subprocess
    csrc = os.path.join(ROOT, "medical-rag_amd", "csrc")
    subprocess.run(["make", "-C", csrc, "asan", "-j4"], check=True, capture_output=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")   # the HIP runtime keeps its own allocations until exit
    run = subprocess.run([os.path.join(ROOT, "medical-rag_amd", "abi_check_asan")], capture_output=True, text=True, env=env,
                         timeout=300)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "abi_check: 0 failure(s)" in run.stdout
    assert "AddressSanitizer" not in run.stderr


def test_configure_from_config_reads_optional_keys():
    from medical_rag_amd import local_rag as rag
    before = rag.configure()
    try:
        # the reference's own config has none of the keys: nothing changes
        assert rag.configure_from_config({"retrievers": ["local"], "local_index_path": "x.json"}) == before
        got = rag.configure_from_config({"local_index_dtype": "bf16", "local_retriever_devices": [3, 4],
                                         "local_top_k": 5, "local_window": 2})
        assert (got["store_dtype"], got["device"], got["top_k"], got["window"]) == ("bf16", 3, 5, 2)
        assert got["devices"] == [3, 4]          # the whole list: the index is row-sharded over it
        assert rag.configure_from_config({"local_retriever_devices": 1})["devices"] == [1]
        assert rag.configure_from_config({"local_retriever_devices": "2,0"})["devices"] == [2, 0]
        assert rag.configure_from_config({"local_retriever_devices": []})["devices"] == [2, 0]   # empty: unchanged
        with pytest.raises(ValueError):
            rag.configure_from_config({"local_retriever_devices": [0, -1]})
        with pytest.raises(ValueError):
            rag.configure_from_config({"local_index_dtype": "int4"})
        with pytest.raises(ValueError):
            rag.configure_from_config({"local_top_k": 0})
    finally:
        rag.configure(**before)


def test_uploaded_list_fingerprint_sees_edits():
    from medical_rag_amd import local_rag as rag
    pages = [{"index": i, "text": str(i), "embedding": [float(i), 1.0, 2.0]} for i in range(200)]
    a = rag._fingerprint(pages)
    assert rag._fingerprint(pages) == a
    pages[0]["embedding"] = [9.0, 9.0, 9.0]              # vector replaced
    b = rag._fingerprint(pages)
    assert b != a
    pages[0]["embedding"][0] = -1.0                       # vector edited in place
    assert rag._fingerprint(pages) != b
    # every page is looked at, not a sample: page 137 of 200 is on no 64-page grid
    c = rag._fingerprint(pages)
    pages[137]["embedding"] = [float(137), 1.0, 2.0]      # replaced by an equal vector: a new object
    d = rag._fingerprint(pages)
    assert d != c
    pages[138]["embedding"][-1] = 2.5                     # last value edited in place
    assert rag._fingerprint(pages) != d
    pages.append({"index": 200, "text": "x", "embedding": [0.0, 0.0, 1.0]})
    assert rag._fingerprint(pages)[0] == 201
    m = np.zeros((50, 4), dtype=np.float32)
    fm = rag._fingerprint(m)
    m[33, 2] = 3.0                                        # any element of a matrix
    assert rag._fingerprint(m) != fm
    import time
    big = [{"index": i, "text": "", "embedding": [0.5] * 8} for i in range(4032)]
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        rag._fingerprint(big)
        best = min(best, time.perf_counter() - t0)
    assert best < 0.05                                    # milliseconds at the reference's index size


def test_one_hip_runtime_whatever_the_import_order():
    """The library loaded BEFORE torch (the reference's order: load_embeddings
    runs before embed_queries imports torch) must leave the process with one
    HIP runtime, or torch's later GPU initialisation finds no device."""
    import subprocess
    import sys
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "from medical_rag_amd import _native\n"
        "_native.lib().mtr_version()\n"
        "import torch\n"
        "torch.zeros(1)\n"
        "maps = sorted({l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l})\n"
        "print(len(maps), maps)\n" % ROOT)
    run = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stderr
    assert run.stdout.startswith("1 "), run.stdout


def test_rccl_stand_in_is_built_and_exports_what_the_library_resolves():
    """tests/rccl_double: the shared-memory stand-in behind the multi-rank GPU
    tests must be in the tree (built by __graft_entry__.build / make double)."""
    import ctypes
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rccl_double", "librccl_double.so")
    assert os.path.exists(path), "make -C medical-rag_amd/csrc double"
    from medical_rag_amd import _native
    _native.lib()                    # maps the HIP runtime the stand-in links against
    lib = ctypes.CDLL(path)
    for name in ("ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclAllGather", "ncclGetErrorString"):
        assert hasattr(lib, name), name
