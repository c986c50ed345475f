#!/usr/bin/env python3
"""HBM-side traffic of the scoring kernel per search, from two rocprofv3 --pmc
passes (FETCH_SIZE, WRITE_SIZE; each its own run) of bench.py per workload.
Run on the GPU box from the repo root:  python3 profiles/collect_pmc_traffic.py OUT.json
A search on the matrix-core path may be two launches (sample + rest): traffic
is summed over a search's launches.  bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024:
on gfx950 FETCH_SIZE (KiB) counts half the bytes of 16-byte-per-lane streaming loads."""
import json
import subprocess
import sys

STEPS, WARMUP = 3, 1
KERNEL = {"c4": "scan_", "c2": "scan_", "c3": "score_"}  # scan_stream_kernel on these shards (scan_topk_kernel under MTR_SCAN_FORM=1);  # c3: sampling launch (score_topk_mfma_kernel) + score_filter_wide_kernel
out = {}
for wl, kern in KERNEL.items():
    vals = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        tag = f"{wl}_{counter.lower()}"
        subprocess.run(["bash", "profiles/pmc_pass.sh", tag, counter, kern, "--workload", wl, "--steps", str(STEPS),
                        "--warmup", str(WARMUP), "--no-cpu-baseline", "--no-ceiling"], check=True, stdout=subprocess.DEVNULL)
        vals[counter] = json.load(open(f"gpurun_out/pmc/{tag}.json"))[counter]
    searches = STEPS + WARMUP
    fetch = vals["FETCH_SIZE"]["sum"] / searches
    write = vals["WRITE_SIZE"]["sum"] / searches
    out[wl] = {"kernel": kern, "searches": searches, "dispatches": vals["FETCH_SIZE"]["dispatches"],
               "FETCH_SIZE_KiB_per_search": fetch, "WRITE_SIZE_KiB_per_search": write,
               "traffic_bytes_per_launch": (2 * fetch + write) * 1024,
               "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of bench.py (1 GPU), summed "
                         "over the launches of one search; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: on gfx950 "
                         "FETCH_SIZE counts half the bytes of 16-byte-per-lane streaming loads"}
    print(wl, out[wl]["traffic_bytes_per_launch"] / 1e9, "GB", flush=True)
json.dump(out, open(sys.argv[1], "w"), indent=1)
