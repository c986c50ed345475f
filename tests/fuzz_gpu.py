#!/usr/bin/env python3
"""Long randomized parity run on the GPU, beyond what the pytest suite repeats
every time: random shapes, storage types, row offsets, forced kernel forms
(streaming loads, ring, 256-row tiles, rows through registers, sampling launch, 256 x 256-tile kernel,
scan hand-over) and all three search entry points, on small-integer data under
the inner-product metric, where ids AND scores must equal the numpy oracle bit
for bit.     usage: python tests/fuzz_gpu.py SEED CASES"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from medical_rag_amd import _native as N
from oracle import local_rag_oracle as orc
seed = int(sys.argv[1]); cases = int(sys.argv[2])
rng = np.random.default_rng(seed)
t0 = time.time(); fails = 0
for case in range(cases):
    store = str(rng.choice(["bf16", "fp32"]))
    n = int(rng.choice([1, 3, 127, 128, 129, 255, 256, 257, 511, 1000, 4097, 20000, 60001, 131072]))
    d = int(rng.choice([1, 8, 32, 33, 64, 100, 128, 192, 200, 256, 300, 320, 512, 768]))
    b = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 16, 31, 32, 33, 48, 64, 65, 80, 90, 112, 128, 129, 200, 257]))
    k = int(rng.choice([1, 2, 5, 10, 32, 33, 100, 128, 129, 256, 300]))
    opts = {}
    if rng.random() < 0.6:
        opts = dict(mfma_stream_rows=int(rng.integers(0, 2)), mfma_stages=int(rng.choice([2, 3])),
                    mfma_sample_stride=int(rng.choice([0, 2, 5])), scan_nontemporal=int(rng.integers(0, 2)))
        if rng.random() < 0.3: opts["mfma_tile_rows"] = int(rng.choice([128, 256]))
        if store == "bf16" and rng.random() < 0.3: opts["wide_min_batch"] = 1
        if rng.random() < 0.2: opts["mfma_min_batch"] = int(rng.choice([2, 100]))
        if rng.random() < 0.5: opts["mfma_via_regs"] = int(rng.integers(0, 3))
    corpus = rng.integers(-3, 4, size=(n, d)).astype(np.float32)
    queries = rng.integers(-3, 4, size=(b, d)).astype(np.float32)
    shard = N.DeviceShard.from_host(corpus, store_dtype=store, normalize=False, row_offset=int(rng.choice([0, 12345])))
    off = shard.info()["row_offset"]
    want_i, want_s = orc.search(queries, corpus, k, metric="ip")
    want_i = np.where(want_i >= 0, want_i + off, -1)
    with N.options(**opts):
        mode = int(rng.integers(0, 3))
        if mode == 0:
            gi, gs = shard.search(queries, k=k, metric="ip")
        else:
            q = torch.from_numpy(queries).cuda()
            if mode == 1:
                ti, ts = shard.search_device(q, k=k, metric="ip")
            else:
                ti = torch.empty((b, k), dtype=torch.int64, device="cuda"); ts = torch.empty((b, k), dtype=torch.float32, device="cuda")
                try:
                    shard.search_device_noalloc(q, (ti, ts), k=k, metric="ip")
                except ValueError as e:
                    if "bytes of scratch" in str(e):
                        ti, ts = shard.search_device(q, k=k, metric="ip")
                    else:
                        raise
            torch.cuda.synchronize()
            gi, gs = ti.cpu().numpy(), ts.cpu().numpy()
    ok = (gi == want_i).all() and (gs == want_s.astype(np.float32)).all()
    if not ok:
        fails += 1
        print("FAIL", dict(case=case, store=store, n=n, d=d, b=b, k=k, opts=opts, mode=mode), flush=True)
    shard.close()
    if case % 100 == 99: print("progress", case + 1, round(time.time() - t0, 1), "s", flush=True)
print("done", cases, "cases", fails, "failures", round(time.time() - t0, 1), "s")
