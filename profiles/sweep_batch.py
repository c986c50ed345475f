#!/usr/bin/env python3
"""Batch-size sweep on one GPU: queries/s and time per search for B = 1 .. 4096
over one synthetic corpus, to locate the hand-over from the HBM-bound scan to the
matrix-core kernel.  Each point carries the roof that binds it (the longer of one
pass over the rows at the HBM peak and the multiply-adds at the dense matrix peak
of the store type) and the fraction of it reached.
usage: sweep_batch.py [rows dim store k] -> JSON on stdout"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    from medical_rag_amd import _native as native
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    dim = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    store = sys.argv[3] if len(sys.argv) > 3 else "bf16"
    k = int(sys.argv[4]) if len(sys.argv) > 4 else 10
    torch.cuda.set_device(0)
    gq = torch.Generator(device="cuda:0").manual_seed(12345)
    queries = torch.randn((4096, dim), generator=gq, device="cuda:0", dtype=torch.float32)
    shard = bench.build_shard(native, torch, 0, rows, dim, store, 0, [], queries, k)
    esize = 2 if store == "bf16" else 4
    out = {"rows": rows, "dim": dim, "store": store, "k": k, "points": []}
    batches = [int(x) for x in os.environ.get("SWEEP_BATCHES", "1,2,3,4,8,9,16,32,33,64,128,256,512,1024,2048,4096").split(",")]
    for b in batches:
        q = queries[:b].contiguous()
        for _ in range(3):
            shard.search_device(q, k=k)
        torch.cuda.synchronize()
        steps = 20 if b <= 64 else 5
        t0 = time.perf_counter()
        for _ in range(steps):
            shard.search_device(q, k=k)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        t_hbm = rows * dim * esize / (bench.HBM_PEAK_GBS * 1e9)
        t_mfma = 2.0 * b * rows * dim / (bench.MFMA_PEAK_TFLOPS[store] * 1e12)
        out["points"].append({"batch": b, "ms_per_search": dt * 1e3, "queries_per_s": b / dt,
                              "corpus_passes_GBps": rows * dim * esize / dt / 1e9,
                              "tflops": 2.0 * b * rows * dim / dt / 1e12,
                              "binding_roof": "hbm" if t_hbm >= t_mfma else "mfma",
                              "roof_ms": max(t_hbm, t_mfma) * 1e3,
                              "frac_of_binding_roof": max(t_hbm, t_mfma) / dt})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
