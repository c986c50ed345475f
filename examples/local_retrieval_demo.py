#!/usr/bin/env python3
"""The reference's local-retrieval stage, end to end, on the MI355X retriever.

What pipeline/async_main.py:187-195 and pipeline/utils/retrieval.py:70-76 do for
every case -- collect the unique keywords, embed them once, find the best page
per keyword and return it with its neighbours -- but with one batched device
search for all keywords (medical_rag_amd.caller) instead of one CPU scan per
(case, keyword).

Offline demo: builds a small index with the tiny local encoder shipped for the
tests, then queries it.  With real data pass your own files:

    python examples/local_retrieval_demo.py --index data/embedded_pages_hf_<model>.json \\
        --keywords pipeline/data/keywords_sample/keyword_generator_output_sample.jsonl \\
        --model Qwen/Qwen3-Embedding-8B --device cuda
"""
import argparse
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
TINY = os.path.join(ROOT, "tests", "golden", "tiny_encoder")

DEMO_PAGES = [
    "Pulmonary edema appears as bilateral perihilar opacities with Kerley B lines.",
    "Pneumothorax: absent lung markings beyond a visible visceral pleural line.",
    "Pleural effusion blunts the costophrenic angle and layers on decubitus views.",
    "Cardiomegaly is a cardiothoracic ratio above one half on a PA film.",
    "Lobar pneumonia gives a dense consolidation with air bronchograms.",
    "Atelectasis shows volume loss with shift of fissures toward the opacity.",
]
DEMO_CASES = [
    {"case_id": "demo-1", "keywords": ["Pulmonary edema", "Cardiomegaly"]},
    {"case_id": "demo-2", "keywords": ["Pneumothorax", " pleural effusion "]},  # the padded keyword is skipped, as in the reference
]


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--index", help="embedded_pages_hf_*.json written by the index builder (default: a demo index)")
    ap.add_argument("--keywords", help="JSONL with one case per line and a 'keywords' list (default: two demo cases)")
    ap.add_argument("--model", default=TINY)
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--window", type=int, default=1)
    args = ap.parse_args()

    from medical_rag_amd import caller, index_builder
    from medical_rag_amd import local_rag as rag

    if args.index:
        index_path = args.index
    else:
        tmp = tempfile.mkdtemp(prefix="medtutor_demo_")
        pages = [{"index": i + 1, "text": t} for i, t in enumerate(DEMO_PAGES)]
        embedded = index_builder.generate_embeddings_hf(pages, args.model, args.device, batch_size=4)
        index_path = index_builder.index_file_name(tmp, "demo/tiny")
        index_builder.write_index(embedded, index_path)
    pages = rag.load_embeddings(index_path)
    if not pages:
        raise SystemExit(f"no pages loaded from {index_path}")

    if args.keywords:
        with open(args.keywords, encoding="utf-8") as f:
            cases = [json.loads(line) for line in f if line.strip()]
    else:
        cases = DEMO_CASES

    unique = caller.unique_keywords(cases)
    vectors = rag.embed_queries(unique, args.model, args.device)
    kw_map = dict(zip(unique, vectors))
    per_case = caller.local_pages_for_cases(cases, kw_map, pages, window_size=args.window)
    for case, hits in zip(cases, per_case):
        print(json.dumps({"case_id": case.get("case_id"),
                          "local_pages_per_keyword": {kw: [p["index"] for p in found] for kw, found in hits.items()}}))


if __name__ == "__main__":
    main()
