"""Offline builder of the page-embedding index on ROCm.

Counterpart of pipeline/utils/embed.py of the reference: read the OCR splits
(embed.py:19-32), embed every page's markdown with a HuggingFace encoder
(batches of 32, truncation at 8192 tokens, masked mean pooling, L2
normalisation, embed.py:53-79) and write the index.  Differences, all on
purpose:
  * the pooling + normalisation run in one HIP kernel when the encoder is on the
    GPU (csrc/pool_normalize.hip) instead of five torch launches;
  * besides the reference's JSON (same schema, so the reference can still read
    it) the binary sidecar of index_format.py is written, which is what
    load_embeddings uses;
  * a batch that fails is not dropped silently (embed.py:80-81): its page
    numbers are returned in a manifest, so list position <-> page stays
    auditable.
"""
from __future__ import annotations

import glob
import json
import logging
import os

import numpy as np

from . import index_format
from .local_rag import pool_and_normalize


def load_all_pages(split_path_pattern: str) -> list:
    """Concatenated "pages" lists of every file matching the pattern, files in
    sorted order (embed.py:19-32)."""
    files = sorted(glob.glob(split_path_pattern))
    if not files:
        logging.error(f"No files found for pattern: {split_path_pattern}")
        return []
    pages = []
    for name in files:
        with open(name, "r", encoding="utf-8") as f:
            pages.extend(json.load(f).get("pages", []))
    return pages


def generate_embeddings_hf(pages: list, model_name: str, device: str, batch_size: int = 32, max_length: int = 8192,
                           failed: list | None = None) -> list:
    """[{"index", "text", "embedding"}] for every page that could be embedded,
    in page order (embed.py:34-83).  `failed`, when given, receives the list
    positions of the pages of every batch that raised."""
    import torch
    from transformers import AutoModel, AutoTokenizer
    try:
        tok = AutoTokenizer.from_pretrained(model_name)
        model = AutoModel.from_pretrained(model_name).to(device).eval()
    except Exception as exc:
        logging.error(f"Failed to load embedding model '{model_name}': {exc}")
        return []
    if tok.pad_token is None:
        tok.pad_token = tok.eos_token
    out = []
    for start in range(0, len(pages), batch_size):
        batch = pages[start:start + batch_size]
        texts = [p.get("markdown", "") for p in batch]
        try:
            enc = tok(texts, padding=True, truncation=True, max_length=max_length, return_tensors="pt").to(device)
            with torch.no_grad():
                vectors = pool_and_normalize(model(**enc).last_hidden_state, enc["attention_mask"]).float().cpu()
        except Exception as exc:
            logging.error(f"Failed to embed pages {start}..{start + len(batch) - 1}: {exc}")
            if failed is not None:
                failed.extend(range(start, start + len(batch)))
            continue
        for page, text, vec in zip(batch, texts, vectors):
            out.append({"index": page.get("index"), "text": text, "embedding": vec.tolist()})
    return out


def index_file_name(data_dir: str, model_name: str) -> str:
    """data/embedded_pages_hf_<model with '/' -> '_'>.json (embed.py:100,110-111)."""
    return os.path.join(data_dir, "embedded_pages_hf_" + model_name.replace("/", "_") + ".json")


def write_index(embedded_pages: list, path: str, write_json: bool = True):
    """Writes the reference's JSON (indent 2, embed.py:112-113) and the binary sidecar."""
    if write_json:
        with open(path, "w", encoding="utf-8") as f:
            json.dump(embedded_pages, f, indent=2)
    matrix = np.array([p["embedding"] for p in embedded_pages], dtype=np.float32)
    meta = [{k: v for k, v in p.items() if k != "embedding"} for p in embedded_pages]
    return index_format.write_sidecar(path, matrix, meta, bind_to_source=write_json)


def build(split_path_pattern: str, model_name: str, device: str, data_dir: str, batch_size: int = 32) -> dict:
    pages = load_all_pages(split_path_pattern)
    if not pages:
        return {"pages": 0, "embedded": 0, "failed": [], "path": None}
    failed: list = []
    embedded = generate_embeddings_hf(pages, model_name, device, batch_size=batch_size, failed=failed)
    path = None
    if embedded:
        path = index_file_name(data_dir, model_name)
        write_index(embedded, path)
    return {"pages": len(pages), "embedded": len(embedded), "failed": failed, "path": path}


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description="Embed the OCR page splits and write the local retrieval index.")
    ap.add_argument("--splits", default="data/ocr_batches/split_*.json")
    ap.add_argument("--model", default="Alibaba-NLP/gte-Qwen2-7B-instruct")  # the reference's fallback, embed.py:91
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--data-dir", default="data")
    ap.add_argument("--batch-size", type=int, default=32)
    args = ap.parse_args(argv)
    logging.basicConfig(level=logging.INFO)
    report = build(args.splits, args.model, args.device, args.data_dir, args.batch_size)
    print(json.dumps(report))
    return 0 if report["embedded"] else 1


if __name__ == "__main__":
    raise SystemExit(main())
