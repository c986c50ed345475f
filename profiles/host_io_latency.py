#!/usr/bin/env python3
"""Per-call time of the host-buffer entry point (numpy query in, numpy result
out: mtr_index_search = H2D copy + kernels + D2H copy + stream sync) beside the
device-resident call, on one synthetic shard.  usage: host_io_latency.py [rows dim store batch k]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    import bench
    from medical_rag_amd import _native as native
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    dim = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    store = sys.argv[3] if len(sys.argv) > 3 else "bf16"
    batch = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    k = int(sys.argv[5]) if len(sys.argv) > 5 else 10
    torch.cuda.set_device(0)
    gq = torch.Generator(device="cuda:0").manual_seed(12345)
    queries = torch.randn((batch, dim), generator=gq, device="cuda:0", dtype=torch.float32)
    shard = bench.build_shard(native, torch, 0, rows, dim, store, 0, [], queries, k)
    host_q = queries.cpu().numpy()
    out = {"rows": rows, "dim": dim, "store": store, "batch": batch, "k": k}
    for name, call in (("host_buffers", lambda: shard.search(host_q, k=k)),
                       ("device_resident", lambda: shard.search_device(queries, k=k))):
        for _ in range(5):
            call()
        torch.cuda.synchronize()
        n = 100
        t0 = time.perf_counter()
        for _ in range(n):
            call()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        out[name] = {"ms_per_call": dt * 1e3, "queries_per_s": batch / dt}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
