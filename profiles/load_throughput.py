#!/usr/bin/env python3
"""Host -> HBM index load rate (mtr_index_load_rows: PCIe copy + normalise + cast)
for an fp32 host matrix.  usage: load_throughput.py [rows dim store]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    from medical_rag_amd import _native as native
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    dim = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    store = sys.argv[3] if len(sys.argv) > 3 else "bf16"
    rng = np.random.default_rng(0)
    block = rng.standard_normal((65536, dim)).astype(np.float32)
    host = np.empty((rows, dim), dtype=np.float32)
    for r0 in range(0, rows, 65536):
        host[r0:r0 + 65536] = block[: min(65536, rows - r0)]
    out = {"rows": rows, "dim": dim, "store": store, "host_GB": host.nbytes / 1e9}
    for attempt in range(2):
        t0 = time.perf_counter()
        shard = native.DeviceShard.from_host(host, store_dtype=store)
        dt = time.perf_counter() - t0
        out[f"load_s_{attempt}"] = dt
        out[f"GBps_{attempt}"] = host.nbytes / dt / 1e9
        shard.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
