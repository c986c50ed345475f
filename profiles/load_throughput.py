#!/usr/bin/env python3
"""Host -> HBM index load rate (mtr_index_load_rows: pinned double-buffered PCIe
copy + finiteness check + normalise + cast) for pageable host matrices, fp32
sources and bf16 (as-stored index file) sources.
usage: load_throughput.py [quick]      prints one JSON object"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def tiled(np, rows, dim, dtype):
    rng = np.random.default_rng(0)
    block = rng.standard_normal((65536, dim)).astype(np.float32)
    if dtype == "bf16":
        block /= np.linalg.norm(block, axis=1, keepdims=True)
        block = (block.view(np.uint32) >> 16).astype(np.uint16)
    host = np.empty((rows, dim), dtype=block.dtype)
    for r0 in range(0, rows, 65536):
        host[r0:r0 + 65536] = block[: min(65536, rows - r0)]
    return host


def main():
    import numpy as np
    from medical_rag_amd import _native as native
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    cases = [(4032, 4096, "fp32", "fp32"), (4032, 4096, "fp32", "bf16"), (1_000_000, 1024, "fp32", "bf16"),
             (1_000_000, 1024, "bf16", "bf16")]
    if not quick:
        cases += [(10_000_000, 1024, "fp32", "bf16"), (10_000_000, 1024, "bf16", "bf16")]
    out = []
    for rows, dim, src, store in cases:
        host = tiled(np, rows, dim, src)
        rec = {"rows": rows, "dim": dim, "source": src + (" (as stored)" if src == "bf16" else ""), "store": store,
               "host_GB": host.nbytes / 1e9, "loads": []}
        for _ in range(3):
            t0 = time.perf_counter()
            shard = native.DeviceShard.from_host(host, store_dtype=store, as_stored=(src == "bf16"))
            dt = time.perf_counter() - t0
            rec["loads"].append({"s": dt, "host_GBps": host.nbytes / dt / 1e9})
            shard.close()
        rec["best_host_GBps"] = max(l["host_GBps"] for l in rec["loads"])
        out.append(rec)
        print(json.dumps(rec), file=sys.stderr, flush=True)
        del host
    print(json.dumps({"what": "DeviceShard.from_host wall time, pageable numpy source", "cases": out}))


if __name__ == "__main__":
    main()
