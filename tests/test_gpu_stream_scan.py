"""The streaming form of the single-query scan (four scanning waves and a
claiming wave per workgroup, two register sets, the last rows handed out through
a device-wide counter), forced onto shards small enough for the oracle
(`scan_form=2`; by itself the library takes it on shards beyond the caches).

Integer vectors under the inner product: every product and partial sum is exact,
so ids AND scores must equal the oracle's bit for bit, ties included; and the
classic form's answer on the same shard must be the same bits on real-valued
data.  The counter words have to be back at zero after every launch, so each
shape is searched several times in a row."""
import numpy as np
import pytest

from oracle import local_rag_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def native():
    import torch
    assert torch.cuda.is_available()
    from medical_rag_amd import _native
    return _native


# (store, n, d): rows of 2, 3, 4, 6, 8 and 12 sweeps (units of 2, 3 and 4 sweeps), widths
# whose last sweep is partly empty, row counts that are no multiple of a group's
# sixteen, and a shard with barely two groups per wave
SHAPES = [("bf16", 40_007, 256), ("bf16", 70_001, 384), ("bf16", 33_000, 512), ("bf16", 50_000, 768),
          ("bf16", 36_111, 1000), ("fp32", 40_000, 128), ("fp32", 35_003, 192), ("fp32", 33_333, 768),
          ("fp32", 33_001, 250), ("bf16", 200_003, 256)]


@pytest.mark.parametrize("store,n,d", SHAPES)
def test_streaming_form_is_exact_on_integer_data(native, store, n, d):
    rng = np.random.default_rng(n + d)
    corpus = rng.integers(-3, 4, size=(n, d)).astype(np.float32)
    queries = rng.integers(-3, 4, size=(3, d)).astype(np.float32)
    corpus[n - 1] = queries[0] * 1.0            # the shard's last row is the best row of query 0
    corpus[0] = queries[1] * 1.0
    shard = native.DeviceShard.from_host(corpus, store_dtype=store, normalize=False)
    with native.options(scan_form=2, mfma_min_batch=1 << 30, scan_max_qb=1):
        for k in (1, 10, 100):
            for q in range(3):
                want_i, want_s = orc.search(queries[q:q + 1], corpus, k, metric="ip")
                for _ in range(2):              # twice: the second launch needs the first one's counters back at zero
                    idx, score = shard.search(queries[q:q + 1], k=k, metric="ip")
                    assert (idx == want_i).all(), (store, n, d, k, q)
                    assert (score == want_s.astype(np.float32)).all(), (store, n, d, k, q)
        # a batch goes through the same form, one query per pass
        idx, score = shard.search(queries, k=5, metric="ip")
        want_i, want_s = orc.search(queries, corpus, 5, metric="ip")
        assert (idx == want_i).all() and (score == want_s.astype(np.float32)).all()
    shard.close()


@pytest.mark.parametrize("store,n,d", [("bf16", 120_000, 1024), ("fp32", 60_000, 768), ("bf16", 150_001, 768)])
def test_streaming_form_equals_classic_form(native, store, n, d):
    """Real-valued rows, cosine: the two forms add a row's products in the same
    order, so scores and ids are the same bits; both streaming cache policies."""
    import torch
    gen = torch.Generator(device="cuda").manual_seed(n + d)
    rows = torch.randn((n, d), generator=gen, device="cuda")
    queries = torch.randn((4, d), generator=gen, device="cuda")
    rows[n - 1] = queries[2] * 0.5
    shard = native.DeviceShard.alloc(n, d, store_dtype=store)
    shard.load_device(0, rows)
    for q in range(4):
        one = queries[q:q + 1].contiguous()
        with native.options(scan_form=1):
            ci, cs = (t.clone() for t in shard.search_device(one, k=10))
        for nt in (0, 1):
            with native.options(scan_form=2, scan_nontemporal=nt):
                si, ss = (t.clone() for t in shard.search_device(one, k=10))
                hi, hs = shard.search(one.cpu().numpy(), k=10)
            torch.cuda.synchronize()
            assert torch.equal(ci, si) and torch.equal(cs, ss), (store, n, d, q, nt)
            assert (hi == si.cpu().numpy()).all() and (hs == ss.cpu().numpy()).all()
    assert int(shard.search_device(queries[2:3].contiguous(), k=1)[0][0, 0]) == n - 1
    # NaN in the query is reported by this form too
    bad = queries[:1].clone()
    bad[0, 3] = float("nan")
    with native.options(scan_form=2):
        shard.search_device(bad, k=3)
        assert shard.nonfinite() is True
        shard.search_device(queries[:1].contiguous(), k=3)
        assert shard.nonfinite() is False
    shard.close()


def test_streaming_form_read_ceiling_runs(native):
    """The loads-only twin of the form (the live ceiling beside the bench number)
    uses the same counters: it must leave them ready for the next search."""
    import torch
    gen = torch.Generator(device="cuda").manual_seed(5)
    n, d = 100_000, 512
    rows = torch.randn((n, d), generator=gen, device="cuda")
    q = torch.randn((1, d), generator=gen, device="cuda")
    shard = native.DeviceShard.alloc(n, d, store_dtype="bf16")
    shard.load_device(0, rows)
    assert shard.last_kernel() == ""
    with native.options(scan_form=2):
        want = tuple(t.clone() for t in shard.search_device(q, k=10))
        assert shard.last_kernel() == "scan_stream_kernel"
        ms = shard.read_ceiling(reps=3)
        assert ms > 0
        got = shard.search_device(q, k=10)
        torch.cuda.synchronize()
        assert torch.equal(want[0], got[0]) and torch.equal(want[1], got[1])
    with native.options(scan_form=1):
        shard.search_device(q, k=10)
        assert shard.last_kernel() == "scan_topk_kernel"
    shard.search_device(torch.randn((40, d), generator=gen, device="cuda"), k=10)
    assert shard.last_kernel() == "score_topk_mfma_kernel"
    shard.close()


@pytest.mark.parametrize("n,d,b", [(300_011, 256, 64), (400_000, 320, 33), (290_000, 512, 2), (1_200_000, 256, 20)])
def test_matrix_core_ring_with_claimed_tail(native, n, d, b):
    """The tall ring form of the matrix-core kernel with its ranges interleaved and
    the last tiles claimed from a counter (a ninth wave claims), forced onto shards
    of a few tiles per workgroup: integer rows, so ids and scores equal the oracle
    bit for bit; the answer of the form with an even split is the same bits on
    real-valued rows; searched three times in a row (the counters go back to zero)."""
    import torch
    rng = np.random.default_rng(n + d + b)
    forms = dict(mfma_stream_rows=1, mfma_stages=3, mfma_tile_rows=256, mfma_sample_stride=0)
    if n <= 400_000:
        corpus = rng.integers(-3, 4, size=(n, d)).astype(np.float32)
        queries = rng.integers(-3, 4, size=(b, d)).astype(np.float32)
        corpus[n - 1] = queries[0]
        shard = native.DeviceShard.from_host(corpus, store_dtype="bf16", normalize=False)
        want_i, want_s = orc.search(queries, corpus, 10, metric="ip")
        for permille in (880, 500, 990):
            with native.options(mfma_dynamic_tail=1, mfma_static_permille=permille, **forms):
                for _ in range(3):
                    idx, score = shard.search(queries, k=10, metric="ip")
                    assert (idx == want_i).all(), (n, d, b, permille)
                    assert (score == want_s.astype(np.float32)).all(), (n, d, b, permille)
        shard.close()
    gen = torch.Generator(device="cuda").manual_seed(n)
    rows = torch.randn((n, d), generator=gen, device="cuda")
    q = torch.randn((b, d), generator=gen, device="cuda")
    shard = native.DeviceShard.alloc(n, d, store_dtype="bf16")
    shard.load_device(0, rows)
    with native.options(mfma_dynamic_tail=0, **forms):
        ei, es = (t.clone() for t in shard.search_device(q, k=10))
    with native.options(mfma_dynamic_tail=1, **forms):
        for _ in range(3):
            di, ds = (t.clone() for t in shard.search_device(q, k=10))
            torch.cuda.synchronize()
            assert torch.equal(ei, di) and torch.equal(es, ds)
    shard.close()


@pytest.mark.parametrize("n,d,b,k", [(300_011, 256, 128, 10), (150_000, 512, 100, 24), (290_000, 256, 64, 10),
                                     (70_001, 768, 65, 100), (1_000, 256, 128, 10), (200_000, 1024, 40, 5)])
def test_matrix_core_ring_with_rows_through_registers(native, n, d, b, k):
    """The tall ring form whose rows stay in flight in registers (four K steps per
    wave, written into two LDS stages a step ahead; the query tile in a ring of
    five), forced onto small shards, on both matrix instructions (switch value 1:
    32 x 32 x 16 like the other forms; 2: 16 x 16 x 32 like the 256-row filter
    kernel, whose sampling form then provides the floors).  Integer rows: ids and
    scores equal the oracle bit for bit.  Real-valued rows: form 1 returns the bits
    of the form that loads rows straight into LDS, form 2 the bits of the filter
    kernel (same instruction, same order of a sum's terms), and the two agree up to
    swaps among scores equal to 1e-6.  With and without a sampling launch; searched
    several times in a row; a shard of fewer tiles than workgroups and a ragged
    last tile included."""
    import torch
    rng = np.random.default_rng(n + d + b)
    forms = dict(mfma_stream_rows=1, mfma_stages=3, mfma_tile_rows=256, mfma_dynamic_tail=0)
    corpus = rng.integers(-3, 4, size=(n, d)).astype(np.float32)
    queries = rng.integers(-3, 4, size=(b, d)).astype(np.float32)
    corpus[n - 1] = queries[0]
    shard = native.DeviceShard.from_host(corpus, store_dtype="bf16", normalize=False)
    want_i, want_s = orc.search(queries, corpus, k, metric="ip")
    for form in (1, 2):
        for stride in (0, 3):
            with native.options(mfma_via_regs=form, mfma_sample_stride=stride, **forms):
                for _ in range(2):
                    idx, score = shard.search(queries, k=k, metric="ip")
                    assert shard.last_kernel() == "score_topk_mfma_kernel"
                    assert (idx == want_i).all(), (n, d, b, form, stride)
                    assert (score == want_s.astype(np.float32)).all(), (n, d, b, form, stride)
    shard.close()
    gen = torch.Generator(device="cuda").manual_seed(n)
    rows = torch.randn((n, d), generator=gen, device="cuda")
    q = torch.randn((b, d), generator=gen, device="cuda")
    shard = native.DeviceShard.alloc(n, d, store_dtype="bf16")
    shard.load_device(0, rows)
    with native.options(mfma_via_regs=0, mfma_sample_stride=0, **forms):
        ei, es = (t.clone() for t in shard.search_device(q, k=k))
    with native.options(mfma_via_regs=1, mfma_sample_stride=0, **forms):
        for _ in range(3):
            di, ds = (t.clone() for t in shard.search_device(q, k=k))
            torch.cuda.synchronize()
            assert torch.equal(ei, di) and torch.equal(es, ds)
    if n >= 256 * 2 * k * 3:   # the filter kernel always samples: at least 2k sampled tiles
        with native.options(wide_min_batch=1, mfma_sample_stride=3):
            wi, ws = (t.clone() for t in shard.search_device(q, k=k))
            assert shard.last_kernel() == "score_filter_wide_kernel"
    else:
        wi = ws = None
    for stride in (0, 3):
        with native.options(mfma_via_regs=2, mfma_sample_stride=stride, **forms):
            for _ in range(2):
                mi, ms = (t.clone() for t in shard.search_device(q, k=k))
                torch.cuda.synchronize()
                if wi is not None:
                    assert torch.equal(wi, mi) and torch.equal(ws, ms), (n, d, b, stride)
                assert float((ms - es).abs().max()) <= 1e-6
                differ = mi != ei
                if bool(differ.any()):   # a swap inside a run of scores equal to the last bit or two
                    assert float((ms[differ] - es[differ]).abs().max()) <= 1e-6 and float(differ.float().mean()) < 0.01
    shard.close()
