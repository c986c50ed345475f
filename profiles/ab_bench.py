#!/usr/bin/env python3
"""A/B of two builds of the library on ONE box (devices differ by a few percent
under load, so builds are only comparable side by side): same synthetic shard,
same queries, per-launch HIP-event time of the scoring kernel(s) through the
entry points both builds export.
usage: ab_bench.py LIB_A LIB_B rows dim batch k [reps [bf16|fp32]]"""
import ctypes as C
import json
import sys


def bind(path):
    lib = C.CDLL(path)
    lib.mtr_last_error.restype = C.c_char_p
    lib.mtr_index_alloc.argtypes = [C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.POINTER(C.c_void_p)]
    lib.mtr_index_load_rows.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32, C.c_int32]
    lib.mtr_index_search_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                            C.c_void_p, C.c_void_p]
    lib.mtr_index_profile.argtypes = [C.c_void_p, C.c_int32]
    lib.mtr_index_scan_time.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_double)]
    lib.mtr_index_destroy.argtypes = [C.c_void_p]
    lib.mtr_index_destroy.restype = None
    return lib


def main():
    import torch
    paths = sys.argv[1:3]
    rows, dim, batch, k = (int(x) for x in sys.argv[3:7])
    reps = int(sys.argv[7]) if len(sys.argv) > 7 else 30
    store = 0 if (len(sys.argv) > 8 and sys.argv[8] == "fp32") else 2
    gen = torch.Generator(device="cuda").manual_seed(1)
    queries = torch.randn((batch, dim), generator=gen, device="cuda")
    out_i = torch.empty((batch, k), dtype=torch.int64, device="cuda")
    out_s = torch.empty((batch, k), dtype=torch.float32, device="cuda")
    libs = [bind(p) for p in paths]
    handles = []
    for lib in libs:
        h = C.c_void_p()
        assert lib.mtr_index_alloc(rows, dim, store, 1, 0, 0, C.byref(h)) == 0, lib.mtr_last_error()
        g = torch.Generator(device="cuda").manual_seed(7)
        for r0 in range(0, rows, 262144):
            n = min(262144, rows - r0)
            block = torch.randn((n, dim), generator=g, device="cuda")
            torch.cuda.synchronize()
            assert lib.mtr_index_load_rows(h, r0, C.c_void_p(block.data_ptr()), n, 0, 1) == 0, lib.mtr_last_error()
        handles.append(h)
    stream = torch.cuda.current_stream().cuda_stream
    results = {p: [] for p in paths}
    first = {}
    for rnd in range(4):          # interleaved rounds: drift hits both builds alike
        for p, lib, h in zip(paths, libs, handles):
            for _ in range(3):
                lib.mtr_index_search_device(h, C.c_void_p(queries.data_ptr()), 0, batch, k, 0, C.c_void_p(out_i.data_ptr()),
                                            C.c_void_p(out_s.data_ptr()), C.c_void_p(stream))
            torch.cuda.synchronize()
            first.setdefault(p, out_i.clone())
            lib.mtr_index_profile(h, 1)
            t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(reps):
                lib.mtr_index_search_device(h, C.c_void_p(queries.data_ptr()), 0, batch, k, 0, C.c_void_p(out_i.data_ptr()),
                                            C.c_void_p(out_s.data_ptr()), C.c_void_p(stream))
            t1.record()
            torch.cuda.synchronize()
            n, ms = C.c_int32(), C.c_double()
            lib.mtr_index_scan_time(h, C.byref(n), C.byref(ms))
            lib.mtr_index_profile(h, 0)
            results[p].append({"kernel_ms_per_search": ms.value / reps, "step_ms": t0.elapsed_time(t1) / reps})
    same = bool(torch.equal(first[paths[0]], first[paths[1]]))
    print(json.dumps({"rows": rows, "dim": dim, "batch": batch, "k": k, "same_ids": same,
                      "builds": {p: {"kernel_ms": [round(r["kernel_ms_per_search"], 4) for r in v],
                                     "step_ms": [round(r["step_ms"], 4) for r in v]} for p, v in results.items()}}))
    for lib, h in zip(libs, handles):
        lib.mtr_index_destroy(h)


if __name__ == "__main__":
    main()
