#!/usr/bin/env python3
"""Single-query scan on one shard under several settings of the launch-shape
switches, interleaved over several rounds on ONE box (devices and runs differ
by a per cent, so settings are only comparable side by side): per-launch HIP
event time of the scan kernel.
usage: scan_split_sweep.py rows dim [bf16|fp32] [reps [batch [k]]] -- name=value,name=value ...   (one group per setting;
"default" = no switch set).  SWEEP_LIBRARY=path loads another build of the library (e.g. one with
-DMTR_TIMING_SWITCHES, whose timing_* switches take parts of a kernel out)."""
import json
import os
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from medical_rag_amd import _native as native  # noqa: E402


def main():
    if os.environ.get("SWEEP_LIBRARY"):
        native.LIB_PATH = os.path.abspath(os.environ["SWEEP_LIBRARY"])
    args = sys.argv[1:]
    cut = args.index("--")
    rows, dim = int(args[0]), int(args[1])
    store = args[2] if cut > 2 else "bf16"
    reps = int(args[3]) if cut > 3 else 20
    batch = int(args[4]) if cut > 4 else 1
    k = int(args[5]) if cut > 5 else 10
    settings = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in grp.split(",") if kv and kv != "default")
                for grp in args[cut + 1:]]
    gen = torch.Generator(device="cuda").manual_seed(7)
    shard = native.DeviceShard.alloc(rows, dim, store_dtype=store)
    for r0 in range(0, rows, 262144):
        n = min(262144, rows - r0)
        shard.load_device(r0, torch.randn((n, dim), generator=gen, device="cuda"))
    q = torch.randn((batch, dim), generator=gen, device="cuda")
    times = [[] for _ in settings]
    whole = [[] for _ in settings]
    ids = []
    kernels = []
    for rnd in range(4):
        for si, st in enumerate(settings):
            with native.options(**st):
                for _ in range(3):
                    out = shard.search_device(q, k=k)
                torch.cuda.synchronize()
                if rnd == 0:
                    ids.append(out[0].clone())
                    kernels.append(shard.last_kernel())
                shard.profile(True)
                for _ in range(reps):
                    shard.search_device(q, k=k)
                torch.cuda.synchronize()
                n, ms = shard.scan_time()
                shard.profile(False)
                times[si].append(ms / n)
                # ... and the whole search (every launch of it), without the per-launch events
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    shard.search_device(q, k=k)
                e1.record()
                torch.cuda.synchronize()
                whole[si].append(e0.elapsed_time(e1) / reps)
    nbytes = rows * dim * (2 if store == "bf16" else 4)   # the widths used here need no padding
    for st, t, w in zip(settings, times, whole):
        best = min(t)
        print(json.dumps({"rows": rows, "dim": dim, "store": store, "batch": batch, "k": k, "kernel": kernels[settings.index(st)], "setting": st, "kernel_ms": [round(x, 4) for x in t],
                          "min_ms": round(best, 4), "GBps_at_min": round(nbytes / best / 1e6, 1), "search_ms_min": round(min(w), 4),
                          "same_ids_as_first": bool(torch.equal(ids[0], ids[settings.index(st)]))}))


if __name__ == "__main__":
    main()
