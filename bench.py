#!/usr/bin/env python3
"""Benchmark of the local dense-retrieval hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload NAME]

One "step" is one search call: B queries scored against the whole corpus with
exact top-k, corpus and queries already resident in HBM.  With N > 1 (launched
by torch.distributed.run, or by this script itself from a bare shell, one rank
per GPU) the corpus is row-sharded over the
ranks (total size fixed: strong scaling) and each step ends with the RCCL
all-gather of the per-rank candidates and the k-way merge.

Workloads:
    c4  10M x 1024 bf16, B=1, top-10   (default: the north-star headline shape)
    c2   1M x  768 fp32, B=1, top-10   (HBM-bound GEMV regime)
    c3   1M x 1024 bf16, B=4096, top-100 (matrix-core regime)
    c1   4032 x 4096 fp32, B=34, top-10 (the reference's own index size)
    c5shard  12.5M x 768 bf16, B=64, top-10 (one GPU's share of the 100M-row corpus)
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "queries/sec for exact top-10 over N×d corpus; achieved HBM GB/s vs peak"
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak

WORKLOADS = {
    "c4": dict(rows=10_000_000, dim=1024, store="bf16", batch=1, k=10,
               name="synthetic 10Mx1024 bf16 corpus, 1 query/batch, exact top-10"),
    "c2": dict(rows=1_000_000, dim=768, store="fp32", batch=1, k=10,
               name="synthetic 1Mx768 fp32 corpus, 1 query/batch, exact top-10"),
    "c3": dict(rows=1_000_000, dim=1024, store="bf16", batch=4096, k=100,
               name="synthetic 1Mx1024 bf16 corpus, 4096-query batch, exact top-100"),
    "c1": dict(rows=4032, dim=4096, store="fp32", batch=34, k=10,
               name="MedTutor-sized index: 4032 pages x 4096 (Qwen3-Embedding-8B width) fp32, 34 keyword queries, top-10"),
    "c5shard": dict(rows=12_500_000, dim=768, store="bf16", batch=64, k=10,
                    name="one of the 8 shards of the 100Mx768 bf16 corpus (12.5M rows), 64-query batch, top-10"),
}
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}  # MI355X dense matrix peaks (fp32 = v_mfma_f32_32x32x2_f32)


def needle_vector(torch, queries, qi, j, k, row, device):
    """Planted row j of query qi: the query plus noise sized so that its cosine
    to the query is about 0.95 - 0.45*j/k (well above the random background)."""
    dim = queries.shape[1]
    cos = 0.95 - 0.45 * j / k
    amp = 2.0 * (1.0 / (cos * cos) - 1.0) ** 0.5
    gen = torch.Generator(device=device).manual_seed(77 + row)
    noise = torch.randn(dim, generator=gen, device=device)
    return queries[qi] * 2.0 + amp * noise * (queries[qi].norm() / dim ** 0.5)


def build_shard(native, torch, lo, hi, dim, store, device, needles, queries, k):
    """Rows [lo, hi) of the synthetic corpus: i.i.d. N(0,1) rows (normalised by
    the loader), generated on the GPU in chunks seeded by the chunk's global
    position, plus planted needle rows  query + noise  at known global ids."""
    shard = native.DeviceShard.alloc(hi - lo, dim, store_dtype=store, normalize=True, device=device, row_offset=lo)
    chunk = 262_144
    gen = torch.Generator(device=f"cuda:{device}")
    first = (lo // chunk) * chunk
    for c0 in range(first, hi, chunk):
        gen.manual_seed(0xC0FFEE + c0 // chunk)
        block = torch.randn((chunk, dim), generator=gen, device=f"cuda:{device}", dtype=torch.float32)
        a, b = max(c0, lo), min(c0 + chunk, hi)
        part = block[a - c0:b - c0]
        for (qi, j, row) in needles:
            if a <= row < b:
                part[row - a] = needle_vector(torch, queries, qi, j, k, row, f"cuda:{device}")
        shard.load_device(a - lo, part.contiguous())
        del block, part
    return shard


def pmc_traffic(workload, default_shape):
    """HBM-side bytes per scoring launch from the committed rocprofv3 PMC passes
    (profiles/*pmc_traffic.json, newest round first); None when the run is not
    the profiled shape."""
    import glob
    if not default_shape:
        return None, None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json")), reverse=True):
        try:
            with open(path) as f:
                rec = json.load(f).get(workload)
        except (OSError, ValueError):
            continue
        if rec:
            return rec["traffic_bytes_per_launch"], os.path.relpath(path, ROOT)
    return None, None


def cpu_baseline(dim, k, full_rows, budget_s=12.0):
    """The numpy float64 restatement of the reference's per-query path
    (row re-normalisation + dgemv + top-k on every call, as
    pipeline/utils/local_rag.py:76-78 does) on a bounded row sample."""
    import numpy as np
    from oracle import local_rag_oracle as orc
    sample_rows = 200_000 if dim <= 1024 else 50_000
    rng = np.random.default_rng(0)
    corpus = rng.standard_normal((sample_rows, dim))
    q = rng.standard_normal((1, dim))
    orc.search(q, corpus, k)  # warm-up
    n_calls, t0 = 0, time.perf_counter()
    while True:
        orc.search(q, corpus, k)
        n_calls += 1
        el = time.perf_counter() - t0
        if el > budget_s or n_calls >= 50:
            break
    per_call = el / n_calls
    threads = int(os.environ.get("OMP_NUM_THREADS", 0) or 0) or len(os.sched_getaffinity(0))
    # The strongest CPU form of the same arithmetic, for scale: rows normalised
    # once and kept in fp32, so a call is one sgemv plus a partial sort.
    unit_rows = orc.normalize_rows(corpus).astype(np.float32)
    q32 = orc.normalize_rows(q).astype(np.float32)
    m_calls, t1 = 0, time.perf_counter()
    while True:
        s = unit_rows @ q32[0]
        top = np.argpartition(-s, k)[:k]
        top = top[np.lexsort((top, -s[top]))]
        m_calls += 1
        el2 = time.perf_counter() - t1
        if el2 > 3.0 or m_calls >= 50:
            break
    per_call32 = el2 / m_calls
    # ... and the float64 restatement on ONE core (BLAS threads limited to 1)
    one_core = None
    try:
        from threadpoolctl import threadpool_limits
        small = corpus[: sample_rows // 4]
        with threadpool_limits(limits=1):
            orc.search(q, small, k)
            c_calls, t2 = 0, time.perf_counter()
            while True:
                orc.search(q, small, k)
                c_calls += 1
                el3 = time.perf_counter() - t2
                if el3 > 4.0 or c_calls >= 20:
                    break
        one_core = {"value": 1.0 / (el3 / c_calls * full_rows / len(small)), "unit": "queries/s", "cores": 1,
                    "sample": f"{c_calls} calls over {len(small)}x{dim} rows, scaled linearly"}
    except ImportError:
        pass
    return {
        "value": 1.0 / (per_call * full_rows / sample_rows),
        "unit": "queries/s",
        "cores": threads,
        "kind": "port",
        "sample": f"numpy float64 oracle, {n_calls} single-query calls over a {sample_rows}x{dim} row sample "
                  f"({per_call * 1e3:.1f} ms/call), scaled linearly to {full_rows} rows",
        "one_core": one_core,
        "prenormalised_fp32_sgemv": {"value": 1.0 / (per_call32 * full_rows / sample_rows), "unit": "queries/s",
                                     "ms_per_call_on_sample": per_call32 * 1e3},
    }


def launch_ranks(n_gpus):
    """`python bench.py --gpus N` from a bare shell: this process has touched no
    GPU yet; it starts one child per GPU with the torch.distributed environment
    set (what torch.distributed.run would do), lets rank 0 print the JSON line
    and returns the worst exit code."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    children = []
    for rank in range(n_gpus):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n_gpus),
                   LOCAL_WORLD_SIZE=str(n_gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs between processes on this driver
        children.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    codes = [child.wait() for child in children]
    return max((abs(c) for c in codes), default=0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c4", choices=sorted(WORKLOADS))
    ap.add_argument("--rows", type=int, default=None)
    ap.add_argument("--dim", type=int, default=None)
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--k", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ceiling", action="store_true",
                    help="skip the loads-only twin of the scan after the timed region (for counter passes: it is an "
                         "instantiation of the same kernel)")
    ap.add_argument("--force-exchange", action="store_true",
                    help="initialise the process group and run the all-gather + merge even with one rank")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="all ranks share GPU 0 and exchange over gloo (RCCL refuses two ranks on one device): a "
                         "rehearsal of the multi-rank code path on a one-GPU box, not a measurement")
    ap.add_argument("--torch-exchange", action="store_true",
                    help="all-gather through torch.distributed instead of the library's RCCL communicator")
    ap.add_argument("--pipeline-depth", type=int, default=-1,
                    help="multi-GPU runs: searches whose exchange may still be in flight when the next scan starts "
                         "(0: every search waits for its own exchange on the main stream; default: 2 from four "
                         "ranks, where the all-gather takes long enough to be worth hiding, else 0)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))   # nothing in this process has touched a GPU

    # Only the JSON line may reach stdout: RCCL prints a version banner there when a
    # communicator is created.  Everything else this process (and the libraries it loads)
    # writes to file descriptor 1 goes to stderr; the line itself goes to the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from medical_rag_amd import _native as native
    from medical_rag_amd.sharded import ShardedIndex, shard_bounds

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"rank {rank} wants GPU {local_rank}, but only {torch.cuda.device_count()} are visible")
    torch.cuda.set_device(local_rank)
    comm = None
    comm_note = None
    if world > 1 or args.force_exchange:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if args.rehearse_on_one_gpu:
            # gloo carries the id, barriers and timing; the library's communicator is rehearsed too when
            # MTR_RCCL_LIBRARY names a collective library that accepts several ranks on one device
            # (the tests' shared-memory stand-in, tests/rccl_double)
            dist.init_process_group("gloo", rank=rank, world_size=world)
            want_comm = bool(os.environ.get("MTR_RCCL_LIBRARY")) and not args.torch_exchange
            flag_device = "cpu"
        else:
            # torch.distributed carries the RCCL unique id, the barriers and the max over ranks of the
            # elapsed time; the searches' all-gather runs on the library's own communicator.
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local_rank}"))
            want_comm = not args.torch_exchange
            flag_device = f"cuda:{local_rank}"
        if want_comm:
            # Either every rank exchanges through the library's communicator or none does.
            try:
                comm = native.Comm.from_process_group(local_rank)
            except Exception as exc:
                comm_note = f"{type(exc).__name__}: {exc}"
                print(f"[bench] rank {rank}: no library communicator ({comm_note})", file=sys.stderr, flush=True)
            agreed = torch.tensor([1 if comm is not None else 0], device=flag_device, dtype=torch.int32)
            dist.all_reduce(agreed, op=dist.ReduceOp.MIN)
            if int(agreed.item()) == 0:
                if comm is not None:
                    comm.close()
                    comm = None
                comm_note = comm_note or "another rank could not create it"

    w = dict(WORKLOADS[args.workload])
    if args.rows:
        w["rows"] = args.rows
    if args.dim:
        w["dim"] = args.dim
    if args.batch:
        w["batch"] = args.batch
    if args.k:
        w["k"] = args.k
    rows, dim, store, batch, k = w["rows"], w["dim"], w["store"], w["batch"], w["k"]
    esize = 2 if store == "bf16" else 4
    if args.rows or args.dim or args.batch or args.k:
        # an overridden shape is named by what actually runs, not by the workload it started from
        w["name"] = (f"synthetic {rows}x{dim} {store} corpus, {batch} quer{'y' if batch == 1 else 'ies'}/batch, "
                     f"exact top-{k} (workload {args.workload} with overrides)")

    gq = torch.Generator(device=f"cuda:{local_rank}").manual_seed(12345)
    queries = torch.randn((batch, dim), generator=gq, device=f"cuda:{local_rank}", dtype=torch.float32)
    # k needle rows per query (first 4 queries), spread over the whole corpus
    needles = []
    for qi in range(min(batch, 4)):
        for j in range(k):
            needles.append((qi, j, (qi * 7919 + (j * 2 + 1) * (rows // (2 * k)) + 13) % rows))
    lo, hi = shard_bounds(rows, world, rank)
    if hi <= lo:
        raise SystemExit(f"rank {rank} would hold no rows ({rows} rows over {world} ranks)")
    t_build = time.perf_counter()
    shard = build_shard(native, torch, lo, hi, dim, store, local_rank, needles, queries, k)
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t_build
    index = ShardedIndex(shard, always_exchange=args.force_exchange, comm=comm)

    # Steps are independent searches.  With several ranks the all-gather and the
    # final merge of a search run on a side stream while the next search's scan
    # starts (a serving loop would do the same); every result is waited for
    # before the clock stops.
    exchanging = world > 1 or args.force_exchange
    depth = args.pipeline_depth if args.pipeline_depth >= 0 else (2 if world >= 4 else 0)
    if not exchanging:
        depth = 0
    pending = []

    def step():
        if depth <= 0:
            return index.search_device(queries, k=k, metric="cosine")
        pending.append(index.search_device_async(queries, k=k, metric="cosine", depth=depth))
        if len(pending) > depth - 1:
            return pending.pop(0).result()
        return None

    def drain(last):
        while pending:
            last = pending.pop(0).result()
        return last

    out = None
    for _ in range(args.warmup):
        out = step() or out
    out = drain(out)
    torch.cuda.synchronize()
    shard.profile(True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step() or out
    out = drain(out)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    last_kernel = shard.last_kernel()
    ceiling_ms = shard.read_ceiling(20) if (rank == 0 and batch == 1 and not args.no_ceiling) else None
    n_launch, scan_ms = shard.scan_time()
    samples = sorted(shard.scan_samples())
    shard.profile(False)
    if world > 1:
        t = torch.tensor([elapsed], device="cpu" if args.rehearse_on_one_gpu else f"cuda:{local_rank}", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # parity at full size: exactly the planted needles must come back, each with
    # its float64 cosine (within the storage tolerance), best first
    ids = out[0].cpu().numpy()
    got_scores = out[1].cpu().numpy()
    tol = 1e-3 if store == "bf16" else 1e-6
    needle_ok = True
    for qi in range(min(batch, 4)):
        mine = [(j, row) for (q_, j, row) in needles if q_ == qi]
        want = {}
        for j, row in mine:
            v = needle_vector(torch, queries, qi, j, k, row, f"cuda:{local_rank}").double()
            q64 = queries[qi].double()
            want[row] = float((v @ q64) / (v.norm() * q64.norm()))
        needle_ok &= sorted(ids[qi].tolist()) == sorted(want)
        if needle_ok:
            needle_ok &= all(abs(want[r] - s) <= tol for r, s in zip(ids[qi].tolist(), got_scores[qi].tolist()))
            needle_ok &= all(a >= b for a, b in zip(got_scores[qi][:-1], got_scores[qi][1:]))

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        scan_avg_ms = scan_ms / max(n_launch, 1)
        local_bytes = (hi - lo) * dim * esize  # algorithmic bytes of one scan launch on this rank
        achieved = local_bytes / (scan_avg_ms * 1e-3) / 1e9
        passes = n_launch / args.steps
        default_shape = world == 1 and not (args.rows or args.batch or args.k or args.dim)
        traffic, traffic_file = pmc_traffic(args.workload, default_shape)
        # The library hands batches of >= 2 (bf16) / 5 (fp32) queries to the matrix-core
        # kernel when rows are a multiple of 128 bytes and k <= 256.
        min_b = int(os.environ.get("MTR_MFMA_MIN_BATCH", "0") or 0) or (2 if store == "bf16" else 5)
        # (rows are stored with a stride of whole 128-byte steps when that pads at most an eighth)
        per_step = 128 // esize
        ld_step = -(-dim // per_step) * per_step
        stride_bytes = ld_step * esize if (dim >= per_step and (ld_step - dim) * 8 <= dim) else -(-dim * esize // 16) * 16
        on_mfma = batch >= min_b and k <= 256 and stride_bytes % 128 == 0 and stride_bytes >= 128
        kernel = "score_topk_mfma_kernel" if on_mfma else "scan_topk_kernel"
        # bf16 shards of >= 1 GB: batches above 128 take the 256 x 256-tile kernel behind the sampling launch
        wide_min = int(os.environ.get("MTR_WIDE_MIN_BATCH", "129"))
        if on_mfma and store == "bf16" and batch >= wide_min and local_bytes >= 1.0e9:
            kernel = "score_filter_wide_kernel"
        kernel = last_kernel or kernel           # what the library says it launched
        on_mfma = kernel in ("score_topk_mfma_kernel", "score_filter_wide_kernel")
        flops = 2.0 * batch * (hi - lo) * dim
        peak_tf = MFMA_PEAK_TFLOPS[store]
        # the roofline that binds is the one whose minimum time is longer
        if on_mfma and flops / (peak_tf * 1e12) > local_bytes / (HBM_PEAK_GBS * 1e9):
            tf = flops / (scan_avg_ms * 1e-3) / 1e12
            roof = {"bound": "mfma", "achieved": tf, "peak": peak_tf, "unit": "TFLOP/s",
                    "frac": tf / peak_tf, "traffic": traffic, "kernel": kernel,
                    "kernel_avg_ms": scan_avg_ms, "launches_per_step": passes,
                    "algorithmic_flops_per_launch": flops}
        else:
            roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                    "kernel": kernel, "kernel_avg_ms": scan_avg_ms,
                    "launches_per_step": passes, "algorithmic_bytes_per_launch": local_bytes}
        if roof["bound"] == "hbm" and ceiling_ms:
            # the scan's loads alone (no arithmetic), measured in this run on this shard
            roof["ceiling_measured"] = {"GBps": local_bytes / (ceiling_ms * 1e-3) / 1e9, "kernel_ms": ceiling_ms,
                                        "frac_of_ceiling": ceiling_ms / scan_avg_ms,
                                        "what": "the single-query scan's loads without its arithmetic (same kernel form, "
                                                "launch shape, cache policy and row hand-out), 20 launches"}
        if traffic is not None:
            roof["traffic_source"] = (f"{traffic_file}: separate rocprofv3 --pmc passes over this command, committed; "
                                      "not collected during this run")
        for r in (roof,):  # spread of the per-launch HIP-event times behind kernel_avg_ms
            if samples:
                r["kernel_ms_p50"] = samples[len(samples) // 2]
                r["kernel_ms_p95"] = samples[min(len(samples) - 1, int(0.95 * len(samples)))]
        res = {
            "metric": METRIC,
            "value": batch * args.steps / elapsed,
            "unit": "queries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "bf16" if store == "bf16" else "f32",
            "data": "synthetic",
            "config": {"workload": w["name"], "rows": rows, "dim": dim, "batch": batch, "k": k,
                       "rows_per_gpu": hi - lo,
                       "parallelism": f"row-shard x{world}" + (" + RCCL all-gather + k-way merge" if world > 1 else ""),
                       "exchange": ("REHEARSAL: all ranks on one GPU, the library's communicator over " + os.environ.get("MTR_RCCL_LIBRARY", "?")
                                    if comm is not None and args.rehearse_on_one_gpu
                                    else "library RCCL communicator (mtr_sharded_search_device)" if comm is not None
                                    else "REHEARSAL: all ranks on one GPU, gloo" if args.rehearse_on_one_gpu and exchanging
                                    else "torch.distributed" + (f" (library communicator unavailable: {comm_note})" if comm_note else "")
                                    if exchanging else None),
                       "exchange_pipeline_depth": depth},
            "roofline": roof,
            "needles_found": bool(needle_ok),
            "build_s": t_build,
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(dim, k, rows)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(res) + "\n").encode())
    shard.close()
    if comm is not None:
        comm.close()
    if dist.is_initialized():
        dist.destroy_process_group()
    if not needle_ok:
        raise SystemExit("planted needle rows were not returned")


if __name__ == "__main__":
    main()
