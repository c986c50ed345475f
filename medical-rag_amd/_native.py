"""ctypes binding of libmedtutor_retrieval.so (C ABI: include/medtutor_retrieval.h).

There is no CPU fallback: if the library is missing or a HIP call fails the
caller gets an exception.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import numpy as np

LIB_NAME = "libmedtutor_retrieval.so"
LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), LIB_NAME)

MTR_OK, MTR_ERR_INVALID, MTR_ERR_NONFINITE, MTR_ERR_HIP, MTR_ERR_COMM, MTR_ERR_OOM = 0, -1, -2, -3, -4, -5
MTR_F32, MTR_F64, MTR_BF16, MTR_F16 = 0, 1, 2, 3
MTR_ROWS_AS_STORED = 256
MTR_COSINE, MTR_IP = 0, 1
MTR_MAX_K = 1024

MTR_EXCHANGE_AUTO, MTR_EXCHANGE_COPY, MTR_EXCHANGE_RCCL = 0, 1, 2
EXCHANGES = {"auto": MTR_EXCHANGE_AUTO, "copy": MTR_EXCHANGE_COPY, "rccl": MTR_EXCHANGE_RCCL}

METRICS = {"cosine": MTR_COSINE, "ip": MTR_IP}
STORE_DTYPES = {"fp32": MTR_F32, "f32": MTR_F32, "float32": MTR_F32, "bf16": MTR_BF16, "bfloat16": MTR_BF16}

_i32, _i64, _vp = C.c_int32, C.c_int64, C.c_void_p
_pp = C.POINTER(C.c_void_p)

# name -> (restype, argtypes); every symbol the header declares.
SIGNATURES = {
    "mtr_index_alloc": (_i32, [_i64, _i32, _i32, _i32, _i32, _i64, _pp]),
    "mtr_index_load_rows": (_i32, [_vp, _i64, _vp, _i64, _i32, _i32]),
    "mtr_index_create": (_i32, [_vp, _i64, _i32, _i32, _i32, _i32, _i32, _i64, _pp]),
    "mtr_index_search": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "mtr_index_search_device": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "mtr_index_search_device_noalloc": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "mtr_index_nonfinite": (_i32, [_vp, C.POINTER(_i32)]),
    "mtr_merge_topk_device": (_i32, [_vp, _vp, _i32, _i64, _i64, _i32, _i32, _i32, _vp, _vp, _i32, _vp]),
    "mtr_comm_unique_id": (_i32, [_vp]),
    "mtr_comm_create": (_i32, [_vp, _i32, _i32, _i32, _pp]),
    "mtr_comm_info": (_i32, [_vp, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32)]),
    "mtr_comm_destroy": (None, [_vp]),
    "mtr_sharded_search_device": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "mtr_comm_exchange_merge_device": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp]),
    "mtr_multi_index_alloc": (_i32, [_i64, _i32, _i32, _i32, C.POINTER(_i32), _i32, _i32, _pp]),
    "mtr_multi_index_load_rows": (_i32, [_vp, _i64, _vp, _i64, _i32]),
    "mtr_multi_index_create": (_i32, [_vp, _i64, _i32, _i32, _i32, _i32, C.POINTER(_i32), _i32, _i32, _pp]),
    "mtr_multi_index_search": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "mtr_multi_index_search_device": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "mtr_multi_index_nonfinite": (_i32, [_vp, C.POINTER(_i32)]),
    "mtr_multi_index_info": (_i32, [_vp, C.POINTER(_i64), C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32),
                                    C.POINTER(_i32), C.POINTER(_i32)]),
    "mtr_multi_index_block": (_i32, [_vp, _i32, _pp, C.POINTER(_i32), C.POINTER(_i64), C.POINTER(_i64)]),
    "mtr_multi_index_read_rows": (_i32, [_vp, _i64, _i64, _vp]),
    "mtr_multi_index_destroy": (None, [_vp]),
    "mtr_pool_normalize": (_i32, [_vp, _i32, _vp, _i32, _i32, _i32, _vp, _i32, _vp]),
    "mtr_index_info": (_i32, [_vp, C.POINTER(_i64), C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32),
                              C.POINTER(_i32), C.POINTER(_i64)]),
    "mtr_index_read_rows": (_i32, [_vp, _i64, _i64, _vp]),
    "mtr_index_profile": (_i32, [_vp, _i32]),
    "mtr_index_scan_time": (_i32, [_vp, C.POINTER(_i32), C.POINTER(C.c_double)]),
    "mtr_index_scan_samples": (_i32, [_vp, C.POINTER(C.c_float), _i32, C.POINTER(_i32)]),
    "mtr_tuning_set": (_i32, [C.c_char_p, _i32]),
    "mtr_tuning_get": (_i32, [C.c_char_p, C.POINTER(_i32)]),
    "mtr_index_read_ceiling": (_i32, [_vp, _i32, C.POINTER(C.c_float)]),
    "mtr_index_last_kernel": (_i32, [_vp, C.c_char_p, _i32]),
    "mtr_json_index_scan": (_i32, [_vp, _i64, _pp]),
    "mtr_json_index_info": (_i32, [_vp, C.POINTER(_i64), C.POINTER(_i32), C.POINTER(_i64)]),
    "mtr_json_index_read": (_i32, [_vp, _vp, _i32, _vp, _i32]),
    "mtr_json_index_free": (None, [_vp]),
    "mtr_index_destroy": (None, [_vp]),
    "mtr_last_error": (C.c_char_p, []),
    "mtr_version": (C.c_char_p, []),
}

_lib = None
_lock = threading.Lock()


class NativeLibraryError(RuntimeError):
    pass


def _share_torch_hip_runtime():
    """One process must not hold two HIP runtimes: the second one finds no GPU.
    PyTorch-ROCm wheels bring their own libamdhip64.so (loaded by file name),
    this library asks for the system's libamdhip64.so.7 (by soname).  Loaded
    after torch, the soname resolves to torch's copy and all is well; loaded
    BEFORE torch -- the reference's order: load_embeddings at
    pipeline/async_main.py:190 runs before embed_queries imports torch at
    pipeline/utils/local_rag.py:23 -- torch would bring a second runtime and
    fail with "No HIP GPUs are available".  So when a torch wheel with a
    bundled runtime is installed, that copy is mapped first (no torch import,
    no GPU initialisation) and serves both.  MEDTUTOR_HIP_RUNTIME=system skips this."""
    if os.environ.get("MEDTUTOR_HIP_RUNTIME", "").lower() == "system":
        return None
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return None
    if spec is None or not spec.origin:
        return None
    bundled = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if not os.path.exists(bundled):
        return None
    try:
        return C.CDLL(bundled, mode=C.RTLD_GLOBAL)
    except OSError:
        return None


def lib():
    """The loaded library; raises NativeLibraryError when it has not been built."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise NativeLibraryError(
                        f"{LIB_PATH} is missing: build it with `make -C medical-rag_amd/csrc` "
                        "(or python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback.")
                _share_torch_hip_runtime()
                handle = C.CDLL(LIB_PATH)
                for name, (res, args) in SIGNATURES.items():
                    fn = getattr(handle, name)
                    fn.restype = res
                    fn.argtypes = args
                _lib = handle
    return _lib


def last_error() -> str:
    msg = lib().mtr_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc: int) -> None:
    """Maps a C return code to the exception the reference's callers expect:
    non-finite input -> ValueError (what sklearn raises through
    pipeline/utils/local_rag.py:77), everything else -> RuntimeError."""
    if rc == MTR_OK:
        return
    msg = last_error()
    if rc == MTR_ERR_NONFINITE:
        raise ValueError(msg or "Input contains NaN or infinity.")
    if rc == MTR_ERR_INVALID:
        raise ValueError(f"medtutor_retrieval: {msg}")
    if rc == MTR_ERR_OOM:
        raise MemoryError(f"medtutor_retrieval: {msg}")
    raise RuntimeError(f"medtutor_retrieval error {rc}: {msg}")


def set_option(name: str, value: int) -> int:
    """Sets a process-wide launch-shape switch (mtr_tuning_set); returns the
    previous value.  For parity tests and timing sweeps."""
    old = _i32()
    check(lib().mtr_tuning_get(name.encode(), C.byref(old)))
    check(lib().mtr_tuning_set(name.encode(), int(value)))
    return old.value


class options:
    """Context manager: launch-shape switches set for the enclosed searches."""

    def __init__(self, **values):
        self._values, self._old = values, {}

    def __enter__(self):
        for name, value in self._values.items():
            self._old[name] = set_option(name, value)
        return self

    def __exit__(self, *exc):
        for name, value in self._old.items():
            set_option(name, value)
        return False


def read_index_json(path: str, dtype=np.float64, threads: int | None = None):
    """The reference's index file (pipeline/utils/embed.py:110-113) read by the
    library's own reader: (matrix (n, d) of `dtype` float64/float32, the rest of
    the file as UTF-8 bytes with every "embedding" array replaced by null).  The
    vectors never become Python lists.  None when the reader declines the file
    (anything but a list of objects with one flat numeric "embedding" each, all
    of one length): the caller then parses it with the json module and gets that
    module's behaviour."""
    import mmap
    code = np_dtype_code(np.empty(0, dtype=dtype))
    if code not in (MTR_F32, MTR_F64):
        raise TypeError("matrix dtype must be float32 or float64")
    if threads is None:
        threads = max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count() or 1))
    L = lib()
    with open(path, "rb") as f:
        if os.fstat(f.fileno()).st_size == 0:
            return None
        with mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ) as view:
            text = np.frombuffer(view, dtype=np.uint8)
            handle = _vp()
            try:
                if L.mtr_json_index_scan(text.ctypes.data, text.size, C.byref(handle)) != MTR_OK:
                    return None
                n, d, rest_len = _i64(), _i32(), _i64()
                check(L.mtr_json_index_info(handle, C.byref(n), C.byref(d), C.byref(rest_len)))
                matrix = np.empty((n.value, d.value), dtype=dtype)
                rest = np.empty(rest_len.value, dtype=np.uint8)
                rc = L.mtr_json_index_read(handle, matrix.ctypes.data if matrix.size else None, code,
                                           rest.ctypes.data if rest.size else None, int(threads))
                if rc != MTR_OK:
                    return None
                return matrix, rest.tobytes()
            finally:
                if handle:
                    L.mtr_json_index_free(handle)
                del text    # the mapping cannot close while an array still views it


def np_dtype_code(a: np.ndarray) -> int:
    if a.dtype == np.uint16:      # raw bfloat16 bits (numpy has no bfloat16)
        return MTR_BF16
    if a.dtype == np.float32:
        return MTR_F32
    if a.dtype == np.float64:
        return MTR_F64
    raise TypeError(f"unsupported array dtype {a.dtype}")


def torch_dtype_code(t) -> int:
    import torch
    if t.dtype == torch.float32:
        return MTR_F32
    if t.dtype == torch.float64:
        return MTR_F64
    if t.dtype == torch.bfloat16:
        return MTR_BF16
    raise TypeError(f"unsupported tensor dtype {t.dtype}")


def as_host_matrix(a, name="array") -> np.ndarray:
    """C-contiguous float32/float64 2-D view of `a` (1-D becomes one row)."""
    if not isinstance(a, np.ndarray):
        a = np.asarray(a)
    if a.dtype == np.uint16 and name == "rows":   # bfloat16 bits, as a bf16 index file stores them
        if a.ndim != 2:
            raise ValueError("bfloat16 rows must be 2-D")
        return np.ascontiguousarray(a)
    if a.dtype not in (np.float32, np.float64):
        a = a.astype(np.float64 if a.dtype.kind in "iu" or a.dtype.itemsize > 4 else np.float32)
    if a.ndim == 1:
        a = a[None, :]
    if a.ndim != 2:
        raise ValueError(f"{name} must be 1-D or 2-D, got shape {a.shape}")
    return np.ascontiguousarray(a)


class DeviceShard:
    """One row shard resident in the HBM of one GPU (mtr_index_t*)."""

    def __init__(self, handle, keepalive=None):
        self._h = handle
        self._keep = keepalive
        self._device = None

    # -- construction -------------------------------------------------------
    @classmethod
    def alloc(cls, n, d, store_dtype="fp32", normalize=True, device=0, row_offset=0):
        out = C.c_void_p()
        check(lib().mtr_index_alloc(int(n), int(d), STORE_DTYPES[store_dtype], int(bool(normalize)), int(device),
                                    int(row_offset), C.byref(out)))
        return cls(out)

    @classmethod
    def from_host(cls, rows, store_dtype="fp32", normalize=True, device=0, row_offset=0, as_stored=False):
        """rows: (n, d) float32 / float64, or uint16 holding bfloat16 bits.
        as_stored: the rows are already unit length (read back from an index
        file written from a shard): they are not normalised a second time."""
        rows = as_host_matrix(rows, "rows")
        shard = cls.alloc(rows.shape[0], rows.shape[1], store_dtype=store_dtype, normalize=normalize, device=device,
                          row_offset=row_offset)
        try:
            shard.load_host(0, rows, as_stored=as_stored)
        except Exception:
            shard.close()
            raise
        return shard

    def load_host(self, row_start, rows, as_stored=False):
        rows = as_host_matrix(rows, "rows")
        if rows.shape[1] != self.d:
            raise ValueError(f"rows have {rows.shape[1]} columns, shard has {self.d}")
        code = np_dtype_code(rows) | (MTR_ROWS_AS_STORED if as_stored else 0)
        check(lib().mtr_index_load_rows(self._h, int(row_start), rows.ctypes.data_as(C.c_void_p), rows.shape[0], code, 0))

    def load_device(self, row_start, tensor):
        """tensor: contiguous (rows, d) torch tensor on this shard's GPU."""
        if tensor.dim() != 2 or tensor.shape[1] != self.d or not tensor.is_contiguous() or not tensor.is_cuda:
            raise ValueError("load_device wants a contiguous (rows, d) tensor on the GPU")
        import torch
        torch.cuda.current_stream(tensor.device).synchronize()
        check(lib().mtr_index_load_rows(self._h, int(row_start), C.c_void_p(tensor.data_ptr()), tensor.shape[0],
                                        torch_dtype_code(tensor), 1))

    # -- queries ------------------------------------------------------------
    def info(self):
        n, ro = _i64(), _i64()
        d, dt, nm, dev = _i32(), _i32(), _i32(), _i32()
        check(lib().mtr_index_info(self._h, C.byref(n), C.byref(d), C.byref(dt), C.byref(nm), C.byref(dev),
                                   C.byref(ro)))
        return {"n": n.value, "d": d.value, "store_dtype": "bf16" if dt.value == MTR_BF16 else "fp32",
                "normalized": bool(nm.value), "device": dev.value, "row_offset": ro.value}

    @property
    def device(self):
        if self._device is None:
            self._device = self.info()["device"]
        return self._device

    @property
    def n(self):
        return self.info()["n"]

    @property
    def d(self):
        return self.info()["d"]

    def search(self, queries, k=10, metric="cosine"):
        """Host in, host out: (idx int64 (B, k), score float32 (B, k))."""
        q = as_host_matrix(queries, "queries")
        if q.shape[1] != self.d:
            raise ValueError(f"queries have {q.shape[1]} columns, shard has {self.d}")
        b = q.shape[0]
        idx = np.empty((b, k), dtype=np.int64)
        score = np.empty((b, k), dtype=np.float32)
        check(lib().mtr_index_search(self._h, q.ctypes.data_as(C.c_void_p), np_dtype_code(q), 0, b, int(k),
                                     METRICS[metric], idx.ctypes.data_as(C.c_void_p),
                                     score.ctypes.data_as(C.c_void_p)))
        return idx, score

    def search_device(self, queries, k=10, metric="cosine", out=None, stream=None):
        """Device in, device out (torch tensors on the shard's GPU), enqueued on
        `stream` (a raw hipStream_t value; default: torch's current stream)
        without synchronising."""
        import torch
        if queries.dim() == 1:
            queries = queries[None, :]
        if not queries.is_cuda or not queries.is_contiguous() or queries.shape[1] != self.d:
            raise ValueError("search_device wants a contiguous (B, d) tensor on the GPU")
        if queries.device.index != self.device:
            raise ValueError(f"queries are on {queries.device}, the shard is on GPU {self.device}")
        b = queries.shape[0]
        if out is None:
            idx = torch.empty((b, k), dtype=torch.int64, device=queries.device)
            score = torch.empty((b, k), dtype=torch.float32, device=queries.device)
        else:
            idx, score = out
            for t, dt in ((idx, torch.int64), (score, torch.float32)):
                if t.device != queries.device or t.dtype != dt or t.numel() != b * k or not t.is_contiguous():
                    raise ValueError("out must be contiguous (B, k) int64 / float32 tensors on the shard's GPU")
        if stream is None:
            stream = torch.cuda.current_stream(queries.device).cuda_stream
        check(lib().mtr_index_search_device(self._h, C.c_void_p(queries.data_ptr()), torch_dtype_code(queries), b,
                                            int(k), METRICS[metric], C.c_void_p(idx.data_ptr()),
                                            C.c_void_p(score.data_ptr()), C.c_void_p(stream)))
        return idx, score

    def search_device_noalloc(self, queries, out, k=10, metric="cosine", stream=None):
        """search_device through the allocation-free entry point: the caller
        brings the (B, k) int64 / float32 output tensors, the library takes its
        scratch from a static device array and allocates nothing."""
        import torch
        idx, score = out
        b = queries.shape[0]
        if (not queries.is_cuda or not queries.is_contiguous() or queries.dim() != 2 or queries.shape[1] != self.d
                or queries.device.index != self.device):
            raise ValueError("search_device_noalloc wants a contiguous (B, d) tensor on the shard's GPU")
        for t, dt in ((idx, torch.int64), (score, torch.float32)):
            if t.device != queries.device or t.dtype != dt or t.numel() != b * k or not t.is_contiguous():
                raise ValueError("out must be contiguous (B, k) int64 / float32 tensors on the shard's GPU")
        if stream is None:
            stream = torch.cuda.current_stream(queries.device).cuda_stream
        check(lib().mtr_index_search_device_noalloc(self._h, C.c_void_p(queries.data_ptr()), torch_dtype_code(queries),
                                                    b, int(k), METRICS[metric], C.c_void_p(idx.data_ptr()),
                                                    C.c_void_p(score.data_ptr()), C.c_void_p(stream)))
        return idx, score

    def nonfinite(self) -> bool:
        """Whether the last search_device call's queries held a NaN or infinity
        (waits for that search)."""
        flag = _i32()
        check(lib().mtr_index_nonfinite(self._h, C.byref(flag)))
        return bool(flag.value)

    def read_rows_bf16(self, row_start=0, n_rows=None, chunk_rows=65536):
        """The stored rows of a bf16 shard as uint16 bfloat16 bits (exact)."""
        inf = self.info()
        if inf["store_dtype"] != "bf16":
            raise ValueError("read_rows_bf16 needs a bf16 shard")
        if n_rows is None:
            n_rows = inf["n"] - row_start
        out = np.empty((n_rows, inf["d"]), dtype=np.uint16)
        for r0 in range(0, n_rows, chunk_rows):
            part = self.read_rows(row_start + r0, min(chunk_rows, n_rows - r0))
            out[r0:r0 + len(part)] = (part.view(np.uint32) >> 16).astype(np.uint16)   # fp32 of a bf16 value: low half zero
        return out

    def read_rows(self, row_start=0, n_rows=None):
        inf = self.info()
        if n_rows is None:
            n_rows = inf["n"] - row_start
        out = np.empty((n_rows, inf["d"]), dtype=np.float32)
        check(lib().mtr_index_read_rows(self._h, int(row_start), int(n_rows), out.ctypes.data_as(C.c_void_p)))
        return out

    def profile(self, enable=True):
        check(lib().mtr_index_profile(self._h, int(bool(enable))))

    def scan_time(self):
        """(launch count, total milliseconds) of the scoring kernel since profile(True)."""
        n, ms = _i32(), C.c_double()
        check(lib().mtr_index_scan_time(self._h, C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def scan_samples(self):
        """Milliseconds of every scoring launch since profile(True), in launch order."""
        n = _i32()
        check(lib().mtr_index_scan_samples(self._h, None, 0, C.byref(n)))
        buf = (C.c_float * max(n.value, 1))()
        check(lib().mtr_index_scan_samples(self._h, buf, n.value, C.byref(n)))
        return [buf[i] for i in range(n.value)]

    def last_kernel(self) -> str:
        """Name of the scoring kernel the last search on this shard took."""
        buf = C.create_string_buffer(64)
        check(lib().mtr_index_last_kernel(self._h, buf, 64))
        return buf.value.decode()

    def read_ceiling(self, reps=20) -> float:
        """Average milliseconds of the scan's loads alone over this shard (mtr_index_read_ceiling)."""
        ms = C.c_float()
        check(lib().mtr_index_read_ceiling(self._h, int(reps), C.byref(ms)))
        return ms.value

    def close(self):
        if self._h is not None and self._h.value:
            lib().mtr_index_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiShard:
    """One index whose row blocks sit on several GPUs, driven by this one
    process (mtr_multi_index_t*): what load_embeddings builds when the
    retriever is given a list of devices.  Same search surface as DeviceShard;
    queries arrive at and results leave from the first device of the list."""

    def __init__(self, handle):
        self._h = handle
        self._info = None

    @classmethod
    def alloc(cls, n, d, devices, store_dtype="fp32", normalize=True, exchange="auto"):
        devs = (_i32 * len(devices))(*[int(x) for x in devices])
        out = C.c_void_p()
        check(lib().mtr_multi_index_alloc(int(n), int(d), STORE_DTYPES[store_dtype], int(bool(normalize)), devs,
                                          len(devices), EXCHANGES[exchange], C.byref(out)))
        return cls(out)

    @classmethod
    def from_host(cls, rows, devices, store_dtype="fp32", normalize=True, exchange="auto", as_stored=False,
                  chunk_rows=None):
        """rows: (n, d) float32 / float64 / uint16 bfloat16 bits, possibly a
        memory map: streamed to the blocks in chunks of chunk_rows rows."""
        rows = rows if isinstance(rows, np.ndarray) and rows.ndim == 2 else as_host_matrix(rows, "rows")
        shard = cls.alloc(rows.shape[0], rows.shape[1], devices, store_dtype=store_dtype, normalize=normalize,
                          exchange=exchange)
        try:
            if chunk_rows is None:
                chunk_rows = max(1, (256 << 20) // max(1, rows.shape[1] * rows.dtype.itemsize))
            for r0 in range(0, rows.shape[0], chunk_rows):
                shard.load_host(r0, rows[r0:r0 + chunk_rows], as_stored=as_stored)
        except Exception:
            shard.close()
            raise
        return shard

    def load_host(self, row_start, rows, as_stored=False):
        rows = as_host_matrix(rows, "rows")
        if rows.shape[1] != self.d:
            raise ValueError(f"rows have {rows.shape[1]} columns, the index has {self.d}")
        code = np_dtype_code(rows) | (MTR_ROWS_AS_STORED if as_stored else 0)
        check(lib().mtr_multi_index_load_rows(self._h, int(row_start), rows.ctypes.data_as(C.c_void_p), rows.shape[0], code))

    def info(self):
        if self._info is None:
            n = _i64()
            d, dt, nm, nb, ex = _i32(), _i32(), _i32(), _i32(), _i32()
            check(lib().mtr_multi_index_info(self._h, C.byref(n), C.byref(d), C.byref(dt), C.byref(nm), C.byref(nb),
                                             C.byref(ex)))
            blocks = []
            for i in range(nb.value):
                dev, lo, hi = _i32(), _i64(), _i64()
                check(lib().mtr_multi_index_block(self._h, i, None, C.byref(dev), C.byref(lo), C.byref(hi)))
                blocks.append({"device": dev.value, "rows": (lo.value, hi.value)})
            self._info = {"n": n.value, "d": d.value, "store_dtype": "bf16" if dt.value == MTR_BF16 else "fp32",
                          "normalized": bool(nm.value), "device": blocks[0]["device"], "row_offset": 0,
                          "blocks": blocks, "exchange": {v: k for k, v in EXCHANGES.items()}[ex.value]}
        return self._info

    @property
    def device(self):
        return self.info()["device"]

    @property
    def n(self):
        return self.info()["n"]

    @property
    def d(self):
        return self.info()["d"]

    def search(self, queries, k=10, metric="cosine"):
        """Host in, host out: (idx int64 (B, k), score float32 (B, k))."""
        q = as_host_matrix(queries, "queries")
        if q.shape[1] != self.d:
            raise ValueError(f"queries have {q.shape[1]} columns, the index has {self.d}")
        b = q.shape[0]
        idx = np.empty((b, k), dtype=np.int64)
        score = np.empty((b, k), dtype=np.float32)
        check(lib().mtr_multi_index_search(self._h, q.ctypes.data_as(C.c_void_p), np_dtype_code(q), b, int(k),
                                           METRICS[metric], idx.ctypes.data_as(C.c_void_p),
                                           score.ctypes.data_as(C.c_void_p)))
        return idx, score

    def search_device(self, queries, k=10, metric="cosine", out=None, stream=None):
        """Device in, device out on the first device of the list, enqueued behind
        `stream` (default: torch's current stream there) without synchronising."""
        import torch
        if queries.dim() == 1:
            queries = queries[None, :]
        if not queries.is_cuda or not queries.is_contiguous() or queries.shape[1] != self.d:
            raise ValueError("search_device wants a contiguous (B, d) tensor on the GPU")
        if queries.device.index != self.device:
            raise ValueError(f"queries are on {queries.device}, the index takes them on GPU {self.device}")
        b = queries.shape[0]
        if out is None:
            idx = torch.empty((b, k), dtype=torch.int64, device=queries.device)
            score = torch.empty((b, k), dtype=torch.float32, device=queries.device)
        else:
            idx, score = out
            for t, dt in ((idx, torch.int64), (score, torch.float32)):
                if t.device != queries.device or t.dtype != dt or t.numel() != b * k or not t.is_contiguous():
                    raise ValueError("out must be contiguous (B, k) int64 / float32 tensors on the index's first GPU")
        if stream is None:
            stream = torch.cuda.current_stream(queries.device).cuda_stream
        check(lib().mtr_multi_index_search_device(self._h, C.c_void_p(queries.data_ptr()), torch_dtype_code(queries), b,
                                                  int(k), METRICS[metric], C.c_void_p(idx.data_ptr()),
                                                  C.c_void_p(score.data_ptr()), C.c_void_p(stream)))
        return idx, score

    def nonfinite(self) -> bool:
        flag = _i32()
        check(lib().mtr_multi_index_nonfinite(self._h, C.byref(flag)))
        return bool(flag.value)

    def read_rows(self, row_start=0, n_rows=None):
        inf = self.info()
        if n_rows is None:
            n_rows = inf["n"] - row_start
        out = np.empty((n_rows, inf["d"]), dtype=np.float32)
        check(lib().mtr_multi_index_read_rows(self._h, int(row_start), int(n_rows), out.ctypes.data_as(C.c_void_p)))
        return out

    def read_rows_bf16(self, row_start=0, n_rows=None, chunk_rows=65536):
        inf = self.info()
        if inf["store_dtype"] != "bf16":
            raise ValueError("read_rows_bf16 needs a bf16 index")
        if n_rows is None:
            n_rows = inf["n"] - row_start
        out = np.empty((n_rows, inf["d"]), dtype=np.uint16)
        for r0 in range(0, n_rows, chunk_rows):
            part = self.read_rows(row_start + r0, min(chunk_rows, n_rows - r0))
            out[r0:r0 + len(part)] = (part.view(np.uint32) >> 16).astype(np.uint16)
        return out

    def close(self):
        if self._h is not None and self._h.value:
            lib().mtr_multi_index_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


MTR_COMM_ID_BYTES = 128


class Comm:
    """One rank's end of the library's RCCL communicator (mtr_comm_t*)."""

    def __init__(self, handle, rank, world, device):
        self._h, self.rank, self.world, self.device = handle, rank, world, device

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(MTR_COMM_ID_BYTES)
        check(lib().mtr_comm_unique_id(buf))
        return buf.raw

    @classmethod
    def create(cls, unique_id: bytes, rank: int, world: int, device: int):
        """Collective: every rank calls it with rank 0's unique_id()."""
        if len(unique_id) != MTR_COMM_ID_BYTES:
            raise ValueError("unique_id must be MTR_COMM_ID_BYTES bytes")
        out = C.c_void_p()
        check(lib().mtr_comm_create(C.c_char_p(unique_id), int(rank), int(world), int(device), C.byref(out)))
        return cls(out, rank, world, device)

    @classmethod
    def from_process_group(cls, device: int, group=None):
        """Builds the communicator over the ranks of a torch.distributed group,
        which only carries rank 0's id to the others."""
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [None]
        if rank == 0:
            try:
                box[0] = cls.unique_id()
            except Exception as exc:        # carried to the others, so that every rank fails instead of waiting
                box[0] = f"{type(exc).__name__}: {exc}"
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        if not isinstance(box[0], (bytes, bytearray)):
            raise RuntimeError(f"rank 0 could not make a communicator id: {box[0]}")
        return cls.create(box[0], rank, world, device)

    def sharded_search(self, shard, queries, out, k=10, metric="cosine", stream=None):
        """Local top-k on `shard` (None: this rank holds no rows), all-gather and
        merge on one stream; out = (idx (B, k) int64, score (B, k) float32)."""
        import torch
        idx, score = out
        b = queries.shape[0]
        if not queries.is_cuda or not queries.is_contiguous() or queries.device.index != self.device:
            raise ValueError("sharded_search wants a contiguous (B, d) tensor on the communicator's GPU")
        if stream is None:
            stream = torch.cuda.current_stream(queries.device).cuda_stream
        handle = shard._h if shard is not None else None
        check(lib().mtr_sharded_search_device(handle, self._h, C.c_void_p(queries.data_ptr()),
                                              torch_dtype_code(queries), b, int(k), METRICS[metric],
                                              C.c_void_p(idx.data_ptr()), C.c_void_p(score.data_ptr()),
                                              C.c_void_p(stream)))
        return idx, score

    def exchange_merge(self, local, out, k_out, stream=None):
        """local = (idx (B, k_in) int64, score (B, k_in) float32) of this rank, or
        None when it has no rows; out receives the global best k_out."""
        import torch
        idx, score = out
        if stream is None:
            stream = torch.cuda.current_stream(idx.device).cuda_stream
        b, k_in = (local[0].shape if local is not None else (idx.shape[0], k_out))
        li = C.c_void_p(local[0].data_ptr()) if local is not None else None
        ls = C.c_void_p(local[1].data_ptr()) if local is not None else None
        check(lib().mtr_comm_exchange_merge_device(self._h, li, ls, int(b), int(k_in), int(k_out),
                                                   C.c_void_p(idx.data_ptr()), C.c_void_p(score.data_ptr()),
                                                   C.c_void_p(stream)))
        return idx, score

    def close(self):
        if self._h is not None and self._h.value:
            lib().mtr_comm_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def merge_topk_device(idx, score, k_out, stream=None, out=None):
    """idx/score: (n_lists, B, k_in) int64 / float32 tensors on one GPU; their
    list dimension may be strided (views into one gathered buffer), the inner
    (B, k_in) block must be dense.  Returns (idx (B, k_out), score (B, k_out))
    ordered by (score desc, id asc)."""
    import torch
    if idx.dim() != 3 or idx.shape != score.shape or not idx.is_cuda:
        raise ValueError("merge_topk_device wants (n_lists, B, k_in) GPU tensors")
    n_lists, b, k_in = idx.shape

    def dense_inner(t):
        return t.stride(2) == 1 and t.stride(1) == k_in and (n_lists == 1 or t.stride(0) >= b * k_in)

    if not dense_inner(idx):
        idx = idx.contiguous()
    if not dense_inner(score):
        score = score.contiguous()
    if out is None:
        out_i = torch.empty((b, k_out), dtype=torch.int64, device=idx.device)
        out_s = torch.empty((b, k_out), dtype=torch.float32, device=idx.device)
    else:
        out_i, out_s = out
    if stream is None:
        stream = torch.cuda.current_stream(idx.device).cuda_stream
    check(lib().mtr_merge_topk_device(C.c_void_p(idx.data_ptr()), C.c_void_p(score.data_ptr()), n_lists,
                                      idx.stride(0) if n_lists > 1 else 0, score.stride(0) if n_lists > 1 else 0,
                                      b, k_in, int(k_out), C.c_void_p(out_i.data_ptr()), C.c_void_p(out_s.data_ptr()),
                                      idx.device.index or 0, C.c_void_p(stream)))
    return out_i, out_s


def pool_normalize(last_hidden_state, attention_mask, stream=None):
    """Masked mean pooling + L2 normalisation of encoder outputs on the GPU:
    (B, T, d) fp32/bf16/fp16 hidden states and (B, T) mask -> (B, d) float32."""
    import torch
    h = last_hidden_state
    if not h.is_cuda or h.dim() != 3:
        raise ValueError("pool_normalize wants a (B, T, d) GPU tensor")
    codes = {torch.float32: MTR_F32, torch.bfloat16: MTR_BF16, torch.float16: MTR_F16}
    if h.dtype not in codes:
        raise TypeError(f"unsupported hidden-state dtype {h.dtype}")
    h = h.contiguous()
    m = attention_mask.to(device=h.device, dtype=torch.int64).contiguous()
    b, t, d = h.shape
    out = torch.empty((b, d), dtype=torch.float32, device=h.device)
    if stream is None:
        stream = torch.cuda.current_stream(h.device).cuda_stream
    check(lib().mtr_pool_normalize(C.c_void_p(h.data_ptr()), codes[h.dtype], C.c_void_p(m.data_ptr()), b, t, d,
                                   C.c_void_p(out.data_ptr()), h.device.index or 0, C.c_void_p(stream)))
    return out
