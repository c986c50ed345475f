"""MI355X-native local dense retriever with the reference's call surface.

Mirrors pipeline/utils/local_rag.py of yale-nlp/medical-rag: the same four
functions with the same argument meaning and failure conventions, so
`from utils.local_rag import ...` in pipeline/async_main.py:19 and
pipeline/utils/retrieval.py:6 can be pointed here unchanged.  What differs is
where the work happens: the page matrix is uploaded once into HBM as
L2-normalised fp32/bf16 rows, and scoring + top-k run in hand-written HIP
kernels behind the C ABI of include/medtutor_retrieval.h.  There is no CPU
scoring path: without the native library or a GPU these functions raise.

    load_embeddings(path)                      -> PageIndex (a Sequence of page dicts)   local_rag.py:90-101
    embed_queries(queries, model, device, bs)  -> list[np.ndarray]                       local_rag.py:39-69
    retrieve_top_index(query_emb, pages)       -> int (list position)                    local_rag.py:71-78
    get_surrounding_pages(idx, pages, window)  -> list[dict]                             local_rag.py:80-88
    search(pages, Q, k, metric)                -> (idx (B,k) int64, score (B,k) f32)     batched top-k extension
"""
from __future__ import annotations

import json
import logging
import os
import threading
from collections.abc import Sequence
from pathlib import Path

import numpy as np

from . import _native, index_format

__all__ = ["PageIndex", "load_embeddings", "embed_queries", "embed_queries_device", "retrieve_top_index",
           "get_surrounding_pages", "search", "configure", "configure_from_config", "invalidate"]

def _device_list(value) -> list:
    """GPU ordinals from an int, a list of ints or a string like "0,1,2"."""
    if isinstance(value, str):
        value = [part for part in value.replace(";", ",").split(",") if part.strip()]
    if isinstance(value, (list, tuple)):
        devices = [int(v) for v in value]
    else:
        devices = [int(value)]
    if not devices or any(dev < 0 for dev in devices):
        raise ValueError("the retriever needs at least one GPU ordinal, none negative")
    return devices


_SETTINGS = {
    # one GPU, or several: the index is then cut into row blocks, one per entry
    "devices": _device_list(os.environ.get("MEDTUTOR_RETRIEVAL_DEVICE", os.environ.get("LOCAL_RANK", "0")) or 0),
    "store_dtype": os.environ.get("MEDTUTOR_RETRIEVAL_DTYPE", "fp32"),
    "top_k": 1,     # the reference keeps one page per keyword (np.argmax, local_rag.py:78)
    "window": 1,    # ... and its two neighbours (get_surrounding_pages' default)
}


def configure(device=None, store_dtype: str | None = None, top_k: int | None = None,
              window: int | None = None, devices=None) -> dict:
    """Sets the GPU -- or the list of GPUs the rows are spread over, in
    contiguous blocks, all driven by this one process -- and the storage type
    ("fp32" or "bf16") used for indexes built after this call, and the batched
    caller's hits per keyword / page window.  Defaults keep the reference's
    behaviour: fp32 on GPU 0, one hit, window 1.  Returns the settings in force
    (`devices`: the list; `device`: its first entry), a dict that configure(**it)
    takes back."""
    if devices is not None or device is not None:
        _SETTINGS["devices"] = _device_list(devices if devices is not None else device)
    if store_dtype is not None:
        if store_dtype not in _native.STORE_DTYPES:
            raise ValueError(f"store_dtype must be one of {sorted(_native.STORE_DTYPES)}")
        _SETTINGS["store_dtype"] = store_dtype
    if top_k is not None:
        if not 1 <= int(top_k) <= _native.MTR_MAX_K:
            raise ValueError(f"top_k must be in [1, {_native.MTR_MAX_K}]")
        _SETTINGS["top_k"] = int(top_k)
    if window is not None:
        if int(window) < 0:
            raise ValueError("window must not be negative")
        _SETTINGS["window"] = int(window)
    return dict(_SETTINGS, device=_SETTINGS["devices"][0])


def configure_from_config(config: dict) -> dict:
    """Reads the retriever's OPTIONAL keys from the pipeline's config dict -- the
    one pipeline/async_main.py:190-195 takes `retrievers`, `local_index_path`,
    `hf_embedding_model` and `embedding_model_device` from.  None of them exists
    in the reference, and leaving them out keeps its behaviour:
        local_index_dtype       "fp32" (default) | "bf16"
        local_retriever_devices GPU ordinal, or a list of them: the index is then
                                row-sharded over all of them and searched from
                                this one process (the vLLM workers pin their own
                                GPUs, async_main.py:39)
        local_top_k             hits per keyword for the batched caller (default 1)
        local_window            neighbouring pages on each side (default 1)
    Unknown or absent keys are ignored; bad values raise ValueError."""
    devices = config.get("local_retriever_devices")
    if isinstance(devices, (list, tuple)) and not devices:
        devices = None
    return configure(device=devices, store_dtype=config.get("local_index_dtype"), top_k=config.get("local_top_k"),
                     window=config.get("local_window"))


class PageIndex(Sequence):
    """The loaded index: behaves as the list of page dicts the reference's
    callers expect (len, truthiness, [i], slicing, items with .items()) and
    owns the device-resident embedding matrix.

    Page dicts keep every key of the index file except "embedding" -- the
    vectors live in HBM, in file order, so list position == device row.
    """

    def __init__(self, pages_meta: list, shard: "_native.DeviceShard"):
        self._pages = pages_meta
        self._shard = shard

    @classmethod
    def from_pages(cls, pages: list, device=None, store_dtype: str | None = None):
        """pages: list of dicts with an "embedding" list each (the schema
        written at pipeline/utils/embed.py:74-79)."""
        if not pages:
            raise ValueError("cannot build a PageIndex from no pages")
        mat = np.array([p["embedding"] for p in pages], dtype=np.float64)
        if mat.ndim != 2:
            raise ValueError("page embeddings must all have the same length")
        meta = [{k: v for k, v in p.items() if k != "embedding"} for p in pages]
        return cls.from_matrix(mat, meta, device=device, store_dtype=store_dtype)

    @classmethod
    def from_matrix(cls, matrix, pages_meta: list | None = None, device=None,
                    store_dtype: str | None = None, normalize: bool = True, as_stored: bool = False):
        """matrix: (n, d) floats, or uint16 bfloat16 bits (a bf16 index file).
        as_stored: rows read back from a shard, not to be normalised again.
        device: one GPU ordinal, or a list of them (default: what configure set):
        with several the rows are cut into contiguous blocks, one per GPU."""
        devices = _SETTINGS["devices"] if device is None else _device_list(device)
        if len(devices) > 1:
            if not (isinstance(matrix, np.ndarray) and matrix.ndim == 2):
                matrix = _native.as_host_matrix(matrix, "rows")
            if pages_meta is None:
                pages_meta = [{"index": i, "text": ""} for i in range(matrix.shape[0])]
            if len(pages_meta) != matrix.shape[0]:
                raise ValueError("one page dict per matrix row is required")
            shard = _native.MultiShard.from_host(matrix, devices, store_dtype=store_dtype or _SETTINGS["store_dtype"],
                                                 normalize=normalize, as_stored=as_stored)
            return cls(pages_meta, shard)
        matrix = _native.as_host_matrix(matrix, "rows")
        if pages_meta is None:
            pages_meta = [{"index": i, "text": ""} for i in range(matrix.shape[0])]
        if len(pages_meta) != matrix.shape[0]:
            raise ValueError("one page dict per matrix row is required")
        shard = _native.DeviceShard.from_host(
            matrix, store_dtype=store_dtype or _SETTINGS["store_dtype"], normalize=normalize,
            device=devices[0], as_stored=as_stored)
        return cls(pages_meta, shard)

    def save_sidecar(self, json_path: str, bind_to_source: bool = True):
        """Writes the binary sidecar of json_path from this index's shard: a bf16
        shard gives a bf16 file holding exactly its stored rows (half the bytes
        of the fp32 form on disk and over PCIe; loading it reproduces the shard
        bit for bit), an fp32 shard the fp32 form.  Only a normalised shard's rows
        are marked "as stored" (loaded without a second normalisation): the rows of
        a shard built with normalize=False are written as they are and normalised
        when load_embeddings reads them, as every index is."""
        info = self._shard.info()
        rows = self._shard.read_rows_bf16() if info["store_dtype"] == "bf16" else self._shard.read_rows()
        return index_format.write_sidecar(json_path, rows, self._pages, bind_to_source=bind_to_source,
                                          as_stored=bool(info["normalized"]))

    # -- Sequence protocol ----------------------------------------------------
    def __len__(self):
        return len(self._pages)

    def __getitem__(self, i):
        return self._pages[i]

    def __bool__(self):
        return len(self._pages) > 0

    # -- search -----------------------------------------------------------------
    @property
    def shard(self):
        return self._shard

    def search(self, queries, k: int = 10, metric: str = "cosine"):
        """Exact top-k: (idx (B, k) int64 list positions, score (B, k) float32),
        rows ordered by (score desc, position asc); idx[:, 0] is what
        retrieve_top_index returns for each query.  numpy / list queries give
        numpy results; a torch tensor already on the shard's GPU (what
        embed_queries_device returns) is searched in place and gives device
        tensors, enqueued on torch's current stream without a host round trip."""
        if type(queries).__module__.startswith("torch") and getattr(queries, "is_cuda", False):
            return self._shard.search_device(queries.contiguous(), k=k, metric=metric)
        return self._shard.search(queries, k=k, metric=metric)

    def top1(self, query_embedding) -> int:
        idx, _ = self._shard.search(query_embedding, k=1, metric="cosine")
        return int(idx[0, 0])

    def close(self):
        self._shard.close()


# Plain page lists handed to retrieve_top_index are uploaded once and reused
# while they are the same list with the same vectors in it (_fingerprint looks
# at every page on every call); at most _UPLOADED_MAX such uploads are kept.
_UPLOADED: dict = {}
_UPLOADED_MAX = 2
_UPLOADED_LOCK = threading.Lock()


_FULL_VALUE_MARKS = 16384   # page lists up to this long have two values of EVERY vector in their mark


def _fingerprint(pages) -> tuple:
    """Identity of a page container's contents, cheap beside a search.  The
    reference rebuilds its matrix on every call
    (pipeline/utils/local_rag.py:76), so any change to the list shows in its next
    answer; here a list is uploaded once and reused while its mark is unchanged.
    The mark covers EVERY page: the identity and length of each vector object,
    and its first and last value (for lists beyond _FULL_VALUE_MARKS pages the
    values of 4096 evenly spaced ones).  Replacing any page's vector, changing
    the number of pages or a vector's length, or rescaling vectors in place
    changes it; an edit of single inner elements of a vector object does not and
    needs invalidate().  A numpy matrix is marked by a checksum of all its bytes."""
    n = len(pages)
    if isinstance(pages, np.ndarray):
        flat = np.ascontiguousarray(pages).view(np.uint8).reshape(-1)
        try:
            import xxhash
            digest = xxhash.xxh3_64_intdigest(flat)
        except ImportError:
            import zlib
            digest = zlib.adler32(flat)
        return (n, pages.shape, str(pages.dtype), digest)
    vectors = [p["embedding"] if isinstance(p, dict) else p for p in pages]
    shape_mark = hash(tuple([(id(v), len(v)) for v in vectors]))
    step = 1 if n <= _FULL_VALUE_MARKS else max(1, n // 4096)
    value_mark = hash(tuple([(float(v[0]), float(v[-1])) for v in vectors[::step] if len(v)]))
    return (n, shape_mark, value_mark)


def invalidate(page_embeddings=None) -> None:
    """Forgets the uploaded copy of a plain page list (all of them when called
    without one), for callers that edit vectors in place."""
    with _UPLOADED_LOCK:
        if page_embeddings is None:
            _UPLOADED.clear()
        else:
            _UPLOADED.pop(id(page_embeddings), None)


def _index_for(page_embeddings) -> PageIndex:
    if isinstance(page_embeddings, PageIndex):
        return page_embeddings
    key = id(page_embeddings)
    mark = _fingerprint(page_embeddings)
    with _UPLOADED_LOCK:
        hit = _UPLOADED.get(key)
        # The entry keeps the container alive, so its id cannot be reused by another object.
        if hit is not None and hit[0] is page_embeddings and hit[1] == mark:
            return hit[2]
    if isinstance(page_embeddings, np.ndarray):
        index = PageIndex.from_matrix(page_embeddings)
    else:
        index = PageIndex.from_pages(list(page_embeddings))
    with _UPLOADED_LOCK:
        # Evicted entries are dropped, not closed: another thread may still be
        # searching one, and its shard is freed when the last reference goes.
        _UPLOADED.pop(key, None)
        while len(_UPLOADED) >= _UPLOADED_MAX:
            _UPLOADED.pop(next(iter(_UPLOADED)))
        _UPLOADED[key] = (page_embeddings, mark, index)
    return index


def load_embeddings(path: str):
    """Loads the page index written by pipeline/utils/embed.py:110-113 and puts
    its vectors into HBM.  Missing file or undecodable JSON -> [] plus an error
    log line, as pipeline/utils/local_rag.py:93-95,99-101 do; a missing native
    library or GPU raises instead of degrading."""
    path_obj = Path(path)
    if not path_obj.exists():
        logging.error(f"Embedding file not found: {path_obj}")
        return []
    cached = index_format.open_sidecar(str(path_obj), with_header=True)
    if cached is not None:
        matrix, meta, hdr = cached
        # a bf16 file is a bf16 shard's stored rows: into a bf16 shard as they are; an fp32
        # shard (local_index_dtype "fp32") takes the same values and normalises them in fp32
        bf16_file = hdr["dtype"] == index_format.DTYPE_BF16
        to_bf16 = _SETTINGS["store_dtype"] in ("bf16", "bfloat16")
        index = PageIndex.from_matrix(matrix, meta, as_stored=hdr["as_stored"] and (to_bf16 or not bf16_file))
        logging.info(f"Loaded {len(index)} page embeddings (d={index.shard.d}) from the binary sidecar of {path_obj}")
        return index
    try:
        fast = index_format.read_json_index(str(path_obj))
        if fast is None:        # not the plain list of pages: let json.load say what it is
            with open(path_obj, "r", encoding="utf-8") as f:
                pages = json.load(f)
    except json.JSONDecodeError:
        logging.error(f"Error decoding JSON from embedding file: {path_obj}")
        return []
    if fast is not None:
        matrix, meta = fast
        if not meta:
            return []
        index = PageIndex.from_matrix(matrix, meta)
    else:
        if not pages:
            return pages
        index = PageIndex.from_pages(pages)
    where = index.shard.info()
    gpus = sorted({blk["device"] for blk in where["blocks"]}) if "blocks" in where else [where["device"]]
    logging.info(f"Loaded {len(index)} page embeddings (d={index.shard.d}) into HBM on GPU {', '.join(map(str, gpus))}")
    return index


def retrieve_top_index(query_embedding: np.ndarray, page_embeddings) -> int:
    """List position of the page most cosine-similar to the query
    (pipeline/utils/local_rag.py:71-78).  Empty query or no pages -> 0;
    NaN/inf anywhere -> ValueError (what sklearn raises there)."""
    query_embedding = np.asarray(query_embedding)
    if query_embedding.size == 0 or len(page_embeddings) == 0:
        return 0
    return _index_for(page_embeddings).top1(query_embedding.reshape(1, -1))


def get_surrounding_pages(target_index: int, all_pages, window_size: int = 1) -> list:
    """The hit page and its neighbours, without their vectors
    (pipeline/utils/local_rag.py:80-88)."""
    n = len(all_pages)
    if n == 0 or not (0 <= target_index < n):
        return []
    first = max(0, target_index - window_size)
    last = min(n, target_index + window_size + 1)
    return [{key: val for key, val in all_pages[i].items() if key != "embedding"} for i in range(first, last)]


def search(page_embeddings, queries, k: int = 10, metric: str = "cosine"):
    """Batched exact top-k over a loaded index (or a plain page list / matrix,
    uploaded on first use)."""
    return _index_for(page_embeddings).search(queries, k=k, metric=metric)


# ---------------------------------------------------------------------------
# Query encoder: stays a PyTorch(-ROCm) forward, as in the reference.
# ---------------------------------------------------------------------------
_ENCODER = {"key": None, "model": None, "tokenizer": None}


def _encoder_for(model_name: str, device: str):
    key = (model_name, device)
    if _ENCODER["key"] == key:
        return _ENCODER["model"], _ENCODER["tokenizer"]
    # transformers/torch are imported only here so that the parent process does
    # not initialise a GPU before its workers spawn (pipeline/utils/local_rag.py:15).
    from transformers import AutoModel, AutoTokenizer
    logging.info(f"Loading query embedding model '{model_name}' onto device '{device}'.")
    try:
        tok = AutoTokenizer.from_pretrained(model_name)
        model = AutoModel.from_pretrained(model_name).to(device).eval()
        if tok.pad_token is None:
            tok.pad_token = tok.eos_token
    except Exception as exc:  # same soft-fail as local_rag.py:34-37
        logging.error(f"Failed to load query embedding model '{model_name}': {exc}", exc_info=True)
        _ENCODER.update(key=None, model=None, tokenizer=None)
        return None, None
    _ENCODER.update(key=key, model=model, tokenizer=tok)
    return model, tok


def pool_and_normalize(last_hidden_state, attention_mask):
    """Masked mean over tokens then L2 normalisation, the arithmetic of
    pipeline/utils/local_rag.py:58-65 (torch tensors in, (B, d) tensor out).
    It runs where the encoder ran: as one HIP kernel for GPU tensors, as the
    reference's torch expression when the caller keeps the encoder on the CPU
    (the reference's default embedding_model_device).  Scoring and top-k have
    no such choice: they only exist on the GPU."""
    import torch
    if last_hidden_state.is_cuda:
        # one fused HIP kernel instead of five elementwise/reduction launches
        return _native.pool_normalize(last_hidden_state, attention_mask)
    mask = attention_mask.to(last_hidden_state.dtype).unsqueeze(-1)
    summed = (last_hidden_state * mask).sum(dim=1)
    counts = mask.sum(dim=1).clamp(min=1e-9)
    return torch.nn.functional.normalize(summed / counts, p=2, dim=1)


def _embed_batches(queries: list, model, tok, device: str, batch_size: int):
    """Yields the pooled, normalised (b, d) float32 tensor of each batch, on `device`."""
    import torch
    with torch.no_grad():
        for start in range(0, len(queries), batch_size):
            enc = tok(queries[start:start + batch_size], padding=True, truncation=True, max_length=512,
                      return_tensors="pt").to(device)
            hidden = model(**enc).last_hidden_state
            yield pool_and_normalize(hidden, enc["attention_mask"]).float()


def embed_queries(queries: list, model_name: str, device: str, batch_size: int = 64) -> list:
    """One float32 (d,) vector per query text (pipeline/utils/local_rag.py:39-69):
    batches of `batch_size`, truncation at 512 tokens, masked mean pooling, L2
    normalisation.  [] for no queries; empty arrays when the model cannot load."""
    if not queries:
        return []
    model, tok = _encoder_for(model_name, device)
    if model is None or tok is None:
        return [np.array([]) for _ in queries]
    vectors = []
    for pooled in _embed_batches(queries, model, tok, device, batch_size):
        vectors.extend(v.cpu().numpy() for v in pooled)
    return vectors


def embed_queries_device(queries: list, model_name: str, device: str = "cuda", batch_size: int = 64):
    """The same vectors as embed_queries, left on the encoder's device as one
    (B, d) float32 tensor so that PageIndex.search can score them without a
    device-to-host-to-device round trip.  None for no queries or when the model
    cannot load (the caller then has nothing to search with)."""
    if not queries:
        return None
    model, tok = _encoder_for(model_name, device)
    if model is None or tok is None:
        return None
    import torch
    return torch.cat(list(_embed_batches(queries, model, tok, device, batch_size)), dim=0).contiguous()
