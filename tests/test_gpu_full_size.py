"""Parity of the kernels AT THE LAUNCH FORMS the benchmark numbers come from:
the streaming (non-temporal) single-query scan on shards beyond the caches, and
the 256 x 256-tile kernel on the full 1M x 1024, 4096-query, top-100 batch.

A full float64 re-scoring of such shards on the CPU is out of reach per run, so
each check re-scores, with the numpy oracle, the ORIGINAL fp32 rows of
(every returned row) united with (a random row sample) and holds the result to
it through parity.check_topk: scores within the storage tolerance, every index
difference a near-tie below it.  Planted rows pin the exact ids."""
import numpy as np
import pytest

from oracle import local_rag_oracle as orc
from parity import check_topk

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def native():
    import torch
    assert torch.cuda.is_available()
    from medical_rag_amd import _native
    return _native


def _corpus_on_gpu(torch, n, d, seed):
    gen = torch.Generator(device="cuda").manual_seed(seed)
    out = torch.empty((n, d), device="cuda", dtype=torch.float32)
    step = 131_072
    for lo in range(0, n, step):
        hi = min(n, lo + step)
        out[lo:hi] = torch.randn((hi - lo, d), generator=gen, device="cuda")
    return out, gen


def _check_against_subsample(torch, corpus, queries_np, got_i, got_s, k, tol, n_sample, seed):
    """Oracle scores over (returned rows) U (n_sample random rows); ids mapped to
    their positions in that sorted column set (monotone, so tie order is kept)."""
    n = corpus.shape[0]
    rng = np.random.default_rng(seed)
    cols = np.union1d(rng.choice(n, size=min(n, n_sample), replace=False), got_i[got_i >= 0].ravel())
    sub = corpus[torch.from_numpy(cols).cuda()].cpu().numpy()
    s64 = orc.scores(queries_np, sub)
    pos = np.searchsorted(cols, got_i)
    assert (cols[pos] == got_i).all()
    return check_topk(pos, got_s, s64, k, tol)


@pytest.mark.parametrize("store,n,d", [("bf16", 400_000, 1024), ("fp32", 200_000, 768), ("bf16", 1_000_000, 256)])
def test_streaming_single_query_scan_at_launch_size(native, store, n, d):
    """B = 1 on a shard above the 192 MiB cache threshold: scan_topk_kernel<T, 1, NT=true>,
    one workgroup per CU -- the kernel form behind the headline number."""
    import torch
    esize = 2 if store == "bf16" else 4
    assert n * d * esize > 192 * 1024 * 1024
    corpus, gen = _corpus_on_gpu(torch, n, d, 31 + d)
    nq, k = 6, 10
    queries = torch.randn((nq, d), generator=gen, device="cuda")
    needles = [[(q * 7919 + j * (n // 3) + 5) % n for j in range(3)] for q in range(nq)]
    needles[0][0], needles[1][1] = 0, n - 1          # first and last row of the shard
    for q in range(nq):
        for j, row in enumerate(needles[q]):
            corpus[row] = queries[q] * 2.0 + 0.25 * (j + 1) * torch.randn(d, generator=gen, device="cuda")
    shard = native.DeviceShard.alloc(n, d, store_dtype=store)
    shard.load_device(0, corpus)
    assert native.set_option("scan_nontemporal", -1) == -1      # the default rule picks the streaming form here
    got_i = np.empty((nq, k), dtype=np.int64)
    got_s = np.empty((nq, k), dtype=np.float32)
    for q in range(nq):
        i1, s1 = shard.search_device(queries[q:q + 1].contiguous(), k=k)
        hi, hs = shard.search(queries[q].cpu().numpy(), k=k)          # host entry, same kernel
        torch.cuda.synchronize()
        assert (i1.cpu().numpy() == hi).all() and (s1.cpu().numpy() == hs).all()
        got_i[q], got_s[q] = hi[0], hs[0]
        assert hi[0, :3].tolist() == needles[q]
    tol = 1e-3 if store == "bf16" else 1e-6
    nt = _check_against_subsample(torch, corpus, queries.cpu().numpy(), got_i, got_s, k, tol, 100_000, 1)
    # the cached form of the same kernel on the same shard: identical
    with native.options(scan_nontemporal=0):
        ci, cs = shard.search(queries[0].cpu().numpy(), k=k)
    assert (ci[0] == got_i[0]).all() and (cs[0] == got_s[0]).all()
    print(f"streaming scan {store} {n}x{d}: {nt} near-tie differences")
    shard.close()


def test_c3_batch_at_full_size(native):
    """1M x 1024 bf16, 4096 queries, top-100: the sampling form + 256 x 256-tile
    filter kernel + key selection at their real launch shapes (16 query tiles,
    real sample stride and list capacities).  256 queries (16 per query tile) are
    re-scored by the oracle over their returned rows and 100k random rows; the
    general kernel's answer for the whole batch must agree up to swaps among
    scores equal to 1e-6."""
    import torch
    n, d, b, k = 1_000_000, 1024, 4096, 100
    corpus, gen = _corpus_on_gpu(torch, n, d, 4096)
    queries = torch.randn((b, d), generator=gen, device="cuda")
    planted = {}
    for q in (0, 255, 256, 2049, 4095):      # first/last query of tiles, k planted rows each
        rows_q = [(q * 104_729 + (2 * j + 1) * (n // (2 * k)) + 13) % n for j in range(k)]
        for j, row in enumerate(rows_q):
            cos = 0.95 - 0.45 * j / k
            amp = (1.0 / (cos * cos) - 1.0) ** 0.5
            corpus[row] = queries[q] / queries[q].norm() + amp * torch.randn(d, generator=gen, device="cuda") / d ** 0.5
        planted[q] = rows_q
    shard = native.DeviceShard.alloc(n, d, store_dtype="bf16")
    shard.load_device(0, corpus)
    wi, ws = shard.search_device(queries, k=k)
    torch.cuda.synchronize()
    wi2, ws2 = (t.clone() for t in shard.search_device(queries, k=k))
    assert torch.equal(wi, wi2) and torch.equal(ws, ws2)
    for q, rows_q in planted.items():
        assert sorted(wi[q].tolist()) == sorted(rows_q), q
    pick = np.arange(0, b, 16)
    got_i, got_s = wi[pick].cpu().numpy(), ws[pick].cpu().numpy()
    nt = _check_against_subsample(torch, corpus, queries[pick].cpu().numpy(), got_i, got_s, k, 1e-3, 100_000, 2)
    # the general (128-row tile) kernel on the same batch
    with native.options(wide_min_batch=1 << 30):
        gi, gs = shard.search_device(queries, k=k)
        torch.cuda.synchronize()
    assert (gs - ws).abs().max().item() <= 1e-6
    differ = gi != wi
    if differ.any():
        assert (gs[differ] - ws[differ]).abs().max().item() <= 1e-6
        assert differ.float().mean().item() < 0.01
    print(f"c3 full size: {nt} near-tie differences vs the oracle on 256 queries, "
          f"{int(differ.sum())} swaps between the two kernels over {b * k} results")
    shard.close()


def test_searches_on_alternating_streams_share_one_handle(native):
    """A handle's scratch buffers serve every stream it is called with: searches
    enqueued back to back on two torch streams and through the host entry (the
    handle's own stream) must each return their own answer."""
    import torch
    gen = torch.Generator(device="cuda").manual_seed(77)
    n, d, k = 300_000, 256, 10
    rows = torch.randn((n, d), generator=gen, device="cuda")
    shard = native.DeviceShard.alloc(n, d, store_dtype="bf16")
    shard.load_device(0, rows)
    batches = [torch.randn((b, d), generator=gen, device="cuda") for b in (48, 1, 33, 5, 64, 1, 2, 40)]
    torch.cuda.synchronize()
    want = []
    for q in batches:
        i, s = shard.search_device(q, k=k)
        torch.cuda.synchronize()
        want.append((i.clone(), s.clone()))
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for _ in range(3):
        got = []
        for j, q in enumerate(batches):
            if j % 3 == 2:
                hi, hs = shard.search(q.cpu().numpy(), k=k)
                got.append((torch.from_numpy(hi).cuda(), torch.from_numpy(hs).cuda()))
            else:
                st = streams[j % 2]
                with torch.cuda.stream(st):
                    got.append(shard.search_device(q, k=k))   # no synchronisation between the calls
        torch.cuda.synchronize()
        for (wi, ws), (gi, gs) in zip(want, got):
            assert torch.equal(wi, gi) and torch.equal(ws, gs)
    shard.close()


def test_device_path_reports_nonfinite_queries(native):
    import torch
    gen = torch.Generator(device="cuda").manual_seed(3)
    rows = torch.randn((5000, 128), generator=gen, device="cuda")
    shard = native.DeviceShard.alloc(5000, 128, store_dtype="fp32")
    shard.load_device(0, rows)
    for b in (1, 7):
        q = torch.randn((b, 128), generator=gen, device="cuda")
        shard.search_device(q, k=5)
        assert shard.nonfinite() is False
        q[b - 1, 17] = float("nan")
        shard.search_device(q, k=5)
        assert shard.nonfinite() is True
        q[b - 1, 17] = float("inf")
        shard.search_device(q, k=5)
        assert shard.nonfinite() is True
    q = torch.randn((2, 128), generator=gen, device="cuda")
    shard.search_device(q, k=5)
    assert shard.nonfinite() is False
    other = torch.empty((2, 5), dtype=torch.int32, device="cuda")
    with pytest.raises(ValueError):
        shard.search_device(q, k=5, out=(other, torch.empty((2, 5), device="cuda")))
    shard.close()


@pytest.mark.parametrize("store,n,d,batches", [("fp32", 1_000_000, 768, (1,)), ("bf16", 10_000_000, 1024, (1, 64, 128, 192)),
                                               ("bf16", 12_500_000, 768, (1, 64, 100))])
def test_benchmark_shapes(native, store, n, d, batches):
    """The shapes the benchmark numbers are quoted on, with the library's own
    launch rules (no switches): 1M x 768 fp32 and 10M x 1024 bf16 with a single
    query, the latter and one GPU's 12.5M x 768 bf16 share of the 100M-row corpus
    also with a 64-query batch, with 100 or 128 queries (the general kernel's form
    that keeps rows in flight in registers, behind the filter kernel's sampling
    form) and with 192 (the filter kernel with a 192-wide query tile and streaming
    rows).  Rows are generated on the GPU; k rows per query
    are planted for the first four queries and must come back, in order; the
    oracle re-scores the original fp32 rows of (returned rows U 100k random rows);
    a repeated search is identical."""
    import torch
    k = 10
    corpus = torch.empty((n, d), device="cuda", dtype=torch.float32)
    gen = torch.Generator(device="cuda").manual_seed(1000 + d)
    step = 262_144
    for lo in range(0, n, step):
        hi = min(n, lo + step)
        corpus[lo:hi] = torch.randn((hi - lo, d), generator=gen, device="cuda")
    queries = torch.randn((max(max(batches), 4), d), generator=gen, device="cuda")
    planted = {}
    for q in range(4):
        rows_q = [(q * 7919 + (2 * j + 1) * (n // (2 * k)) + 13) % n for j in range(k)]
        for j, row in enumerate(rows_q):
            cos = 0.95 - 0.45 * j / k
            amp = (1.0 / (cos * cos) - 1.0) ** 0.5
            corpus[row] = queries[q] / queries[q].norm() + amp * torch.randn(d, generator=gen, device="cuda") / d ** 0.5
        planted[q] = rows_q
    corpus[0] = queries[0] * 3.0                        # the shard's first row: cosine 1, ahead of the ten planted
    planted[0] = [0] + planted[0]
    for q, rows_q in planted.items():                   # expected order: by the rows' float64 cosine (the noise moves it a little)
        cand = corpus[torch.tensor(rows_q, device="cuda")].double()
        cos64 = (cand @ queries[q].double()) / (cand.norm(dim=1) * queries[q].double().norm())
        order = torch.argsort(cos64, descending=True, stable=True).tolist()
        assert float(cos64[order[k - 1]] - cos64[order[k]] if len(rows_q) > k else 1.0) > 2 * (1e-3 if store == "bf16" else 1e-6)
        planted[q] = [rows_q[i] for i in order[:k]]
    shard = native.DeviceShard.alloc(n, d, store_dtype=store)
    for lo in range(0, n, 1 << 20):
        shard.load_device(lo, corpus[lo:lo + (1 << 20)])
    tol = 1e-3 if store == "bf16" else 1e-6
    for b in batches:
        q = queries[:b].contiguous()
        gi, gs = (t.clone() for t in shard.search_device(q, k=k))
        torch.cuda.synchronize()
        assert not shard.nonfinite()
        gi2, gs2 = shard.search_device(q, k=k)
        torch.cuda.synchronize()
        assert torch.equal(gi, gi2) and torch.equal(gs, gs2)
        for qi in range(min(b, 4)):
            assert gi[qi].tolist() == planted[qi], (b, qi)   # planted cosines are some 0.045 apart, far above the tolerance
        pick = np.arange(0, b, max(1, b // 16))
        nt = _check_against_subsample(torch, corpus, q[pick].cpu().numpy(), gi[pick].cpu().numpy(), gs[pick].cpu().numpy(),
                                      k, tol, 100_000, 3)
        hi_i, hi_s = shard.search(q[:1].cpu().numpy(), k=k)                    # the host entry on the same shard
        if b == 1:
            assert (hi_i == gi.cpu().numpy()).all() and (hi_s == gs.cpu().numpy()).all()
        print(f"bench shape {store} {n}x{d} B={b}: {nt} near-tie differences on {len(pick)} queries")
    shard.close()
    del corpus


def _bench_line(extra_args, env_extra):
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **env_extra)
    for name in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(name, None)          # bench.py starts its own ranks from a bare environment
    run = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + extra_args, env=env, capture_output=True,
                         text=True, timeout=600)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-4000:]
    lines = [ln for ln in run.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, run.stdout                   # exactly ONE JSON line on stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("depth", [0, 2])
def test_bench_multi_rank_command_rehearsed_on_one_gpu(depth):
    """The command the driver issues for N > 1 (`bench.py --gpus N`, here N = 2 and
    a small corpus) end to end: rank start-up, the communicator id broadcast,
    agreement of the ranks on the exchange, the row shards, the fused and the
    pipelined step, the max over ranks, ONE JSON line.  Both ranks share this
    box's GPU, so the collective underneath the library's communicator is the
    tests' stand-in (RCCL refuses two ranks on one device): a rehearsal of
    bench.py's and the library's code, not a measurement."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    double = os.path.join(root, "tests", "rccl_double", "librccl_double.so")
    assert os.path.exists(double)
    res = _bench_line(["--gpus", "2", "--rehearse-on-one-gpu", "--rows", "400000", "--steps", "3", "--warmup", "2",
                       "--pipeline-depth", str(depth)], {"MTR_RCCL_LIBRARY": double})
    assert res["n_gpus"] == 2 and res["steps"] == 3 and res["needles_found"] is True
    assert res["config"]["rows"] == 400000 and res["config"]["rows_per_gpu"] == 200000
    assert "library's communicator" in res["config"]["exchange"] and res["config"]["exchange_pipeline_depth"] == depth
    assert "400000x1024" in res["config"]["workload"] and res["value"] > 0
    assert "cpu_baseline" not in res                      # rank 0 times the CPU only in single-GPU runs


def test_bench_names_the_shape_it_ran():
    """config.workload follows --rows / --batch / --k overrides."""
    res = _bench_line(["--workload", "c2", "--rows", "300000", "--batch", "3", "--k", "5", "--steps", "3", "--warmup", "1",
                       "--no-cpu-baseline"], {})
    assert res["config"]["workload"].startswith("synthetic 300000x768 fp32 corpus, 3 queries/batch, exact top-5")
    assert res["roofline"]["bound"] == "hbm" and res["needles_found"] is True
