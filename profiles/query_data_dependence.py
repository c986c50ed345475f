import ctypes as C, json, sys
sys.path.insert(0, "profiles")
from ab_bench import bind
import torch
path = sys.argv[1]; rows, dim, batch = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]); K = int(sys.argv[5]) if len(sys.argv) > 5 else 10
lib = bind(path)
h = C.c_void_p()
assert lib.mtr_index_alloc(rows, dim, 2, 0, 0, 0, C.byref(h)) == 0   # not normalised: inner product
g = torch.Generator(device="cuda").manual_seed(7)
for r0 in range(0, rows, 262144):
    n = min(262144, rows - r0)
    block = torch.randn((n, dim), generator=g, device="cuda"); torch.cuda.synchronize()
    assert lib.mtr_index_load_rows(h, r0, C.c_void_p(block.data_ptr()), n, 0, 1) == 0
stream = torch.cuda.current_stream().cuda_stream
gen = torch.Generator(device="cuda").manual_seed(1)
for name, q in (("random", torch.randn((batch, dim), generator=gen, device="cuda")),
                ("zeros", torch.zeros((batch, dim), device="cuda")),
                ("random", torch.randn((batch, dim), generator=gen, device="cuda")),
                ("one-hot", torch.eye(batch, dim, device="cuda"))):
    oi = torch.empty((batch, K), dtype=torch.int64, device="cuda"); os_ = torch.empty((batch, K), dtype=torch.float32, device="cuda")
    res = []
    for rnd in range(3):
        for _ in range(3):
            lib.mtr_index_search_device(h, C.c_void_p(q.data_ptr()), 0, batch, K, 1, C.c_void_p(oi.data_ptr()), C.c_void_p(os_.data_ptr()), C.c_void_p(stream))
        torch.cuda.synchronize()
        lib.mtr_index_profile(h, 1)
        for _ in range(20):
            lib.mtr_index_search_device(h, C.c_void_p(q.data_ptr()), 0, batch, K, 1, C.c_void_p(oi.data_ptr()), C.c_void_p(os_.data_ptr()), C.c_void_p(stream))
        torch.cuda.synchronize()
        n, ms = C.c_int32(), C.c_double()
        lib.mtr_index_scan_time(h, C.byref(n), C.byref(ms)); lib.mtr_index_profile(h, 0)
        res.append(round(ms.value / 20, 4))
    print(json.dumps({"batch": batch, "queries": name, "kernel_ms": res}), flush=True)
