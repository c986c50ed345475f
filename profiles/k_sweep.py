#!/usr/bin/env python3
"""Time of one single-query search against the length of the result list, with
the scan's short candidate lists (default) and with full lists
(scan_short_lists=0), side by side on one box.
usage: k_sweep.py [rows dim bf16|fp32] -> JSON on stdout"""
import json
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from medical_rag_amd import _native as native  # noqa: E402


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    dim = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    store = sys.argv[3] if len(sys.argv) > 3 else "bf16"
    gen = torch.Generator(device="cuda").manual_seed(7)
    shard = native.DeviceShard.alloc(rows, dim, store_dtype=store)
    for r0 in range(0, rows, 262144):
        n = min(262144, rows - r0)
        shard.load_device(r0, torch.randn((n, dim), generator=gen, device="cuda"))
    q = torch.randn((1, dim), generator=gen, device="cuda")
    out = {"rows": rows, "dim": dim, "store": store, "points": []}
    for k in (1, 10, 32, 33, 64, 100, 256, 257, 512, 1024):
        point = {"k": k}
        ids = {}
        for name, opts in (("short_lists", {}), ("full_lists", {"scan_short_lists": 0})):
            reps = 10 if (k <= 256 or not opts) else 3
            with native.options(**opts):
                for _ in range(2):
                    res = shard.search_device(q, k=k)
                torch.cuda.synchronize()
                ids[name] = res[0].clone()
                t0 = time.perf_counter()
                for _ in range(reps):
                    shard.search_device(q, k=k)
                torch.cuda.synchronize()
                point[name + "_ms"] = round((time.perf_counter() - t0) / reps * 1e3, 4)
        point["same_ids"] = bool(torch.equal(ids["short_lists"], ids["full_lists"]))
        out["points"].append(point)
        print(point, file=sys.stderr, flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
