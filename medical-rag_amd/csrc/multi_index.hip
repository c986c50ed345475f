// One index over several GPUs, driven by ONE process: the form the reference's
// own caller needs.  pipeline/async_main.py:190-195 loads the index once into
// one object of one asyncio process, and pipeline/utils/retrieval.py:70-76
// searches it inline; a corpus that is row-sharded over the GPUs of a node has
// to sit behind that one object.  (comm_rccl.hip is the other form: one
// process per GPU.)
//
// Layout.  The rows are cut into contiguous blocks, one per entry of the
// device list (block i = rows [lo_i, lo_i+1), n / g rows each, the first n % g
// blocks one more; contiguity keeps the lowest-row-wins tie rule across
// blocks).  Every block is an ordinary shard (mtr_index) with row_offset lo_i
// on its GPU, so its candidates carry global row ids.
//
// A search.  The host thread enqueues everything and waits once, at the end:
//   queries -> every GPU (host copy, or a peer copy from the first GPU)
//   local top-k on every GPU, on a stream per block, straight into the
//       block's packed (ids, scores) send buffer
//   ONE exchange: a grouped RCCL all-gather over the communicators of
//       ncclCommInitAll when the devices are distinct (xGMI), or -- when the
//       list names a device more than once, which RCCL refuses, or no RCCL
//       can be loaded -- one peer copy per block into the first GPU's gather
//       buffer behind an event
//   k-way merge (merge_topk.hip) on the first GPU.
// No host thread per device and no host synchronisation between the steps.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/medtutor_retrieval.h"
#include "mtr_device.h"
#include "rccl_loader.h"

namespace mtr {
int set_error(int code, const char* fmt, ...);  // mtr_api.hip: thread-local message behind mtr_last_error()
}

namespace {

#define HIP_TRY(expr)                                                                                        \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess)                                                                                \
            return mtr::set_error(e_ == hipErrorOutOfMemory ? MTR_ERR_OOM : MTR_ERR_HIP, "%s: %s", #expr,    \
                                  hipGetErrorString(e_));                                                    \
    } while (0)

#define RCCL_TRY(r, expr)                                                                              \
    do {                                                                                               \
        ncclResult_t n_ = (expr);                                                                      \
        if (n_ != ncclSuccess) return mtr::set_error(MTR_ERR_COMM, "%s: %s", #expr, (r)->GetErrorString(n_)); \
    } while (0)

struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

int dtype_bytes(int dt) { return dt == MTR_F32 ? 4 : dt == MTR_F64 ? 8 : dt == MTR_BF16 ? 2 : 0; }

// What one block contributes to a search: b*k int64 ids, then b*k fp32 scores,
// padded to whole 8-byte words (the layout comm_rccl.hip exchanges too).
size_t block_words(int b, int k) { return (size_t)b * k + ((size_t)b * k + 1) / 2; }

}  // namespace

struct mtr_multi_index {
    int64_t n = 0;
    int d = 0, dtype = MTR_F32, normalized = 0;
    int g = 0;                        // blocks = entries of the device list
    int exchange = MTR_EXCHANGE_COPY; // the form in use (never AUTO)
    std::mutex mu;
    struct Block {
        int device = 0;
        int64_t lo = 0, hi = 0;
        mtr_index_t* shard = nullptr;  // NULL: no rows (fewer rows than blocks)
        hipStream_t stream = nullptr;
        hipEvent_t done = nullptr;     // behind the block's part of the current search
        void* q = nullptr;             // the queries on this GPU
        size_t q_cap = 0;
        long long* send = nullptr;     // this block's candidates (not used when they go straight into the gather buffer)
        long long* recv = nullptr;     // RCCL form: every block's candidates, as the all-gather leaves them
        size_t words_cap = 0;
        ncclComm_t comm = nullptr;
    };
    std::vector<Block> blocks;
    // on the first GPU (blocks[0].device), where queries arrive and results leave
    long long* gather = nullptr;       // copy form: g blocks of candidates
    size_t gather_words = 0;           // per block
    long long* out_i = nullptr;        // host-entry results before they are copied out
    float* out_s = nullptr;
    size_t out_cells = 0;
    hipEvent_t q_ready = nullptr;      // the caller's stream has the queries in place
    hipEvent_t merged = nullptr;       // the last search's merge has read the gather buffer
    bool merged_set = false;
};

namespace {

int home(const mtr_multi_index* mi) { return mi->blocks[0].device; }

int grow_block(mtr_multi_index* mi, mtr_multi_index::Block& blk, size_t words, size_t q_bytes) {
    DeviceGuard guard(blk.device);
    if (!guard.ok) return mtr::set_error(MTR_ERR_HIP, "hipSetDevice(%d) failed", blk.device);
    if (q_bytes > blk.q_cap) {
        HIP_TRY(hipStreamSynchronize(blk.stream));
        if (blk.q) (void)hipFree(blk.q);
        blk.q = nullptr;
        blk.q_cap = 0;
        HIP_TRY(hipMalloc(&blk.q, q_bytes));
        blk.q_cap = q_bytes;
    }
    if (words > blk.words_cap) {
        HIP_TRY(hipStreamSynchronize(blk.stream));
        if (blk.send) (void)hipFree(blk.send);
        if (blk.recv) (void)hipFree(blk.recv);
        blk.send = blk.recv = nullptr;
        blk.words_cap = 0;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&blk.send), words * 8));
        if (mi->exchange == MTR_EXCHANGE_RCCL)
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&blk.recv), words * 8 * (size_t)mi->g));
        blk.words_cap = words;
    }
    return MTR_OK;
}

// Buffers for a search of b queries and k results, grown on demand and kept.
int grow(mtr_multi_index* mi, int b, int k, size_t q_bytes, bool host_entry) {
    const size_t words = block_words(b, k);
    int rc;
    for (auto& blk : mi->blocks)
        if ((rc = grow_block(mi, blk, words, q_bytes)) != MTR_OK) return rc;
    DeviceGuard guard(home(mi));
    if (mi->exchange == MTR_EXCHANGE_COPY && words > mi->gather_words) {
        // the merge of an earlier search may still read the old buffer
        HIP_TRY(hipStreamSynchronize(mi->blocks[0].stream));
        if (mi->gather) (void)hipFree(mi->gather);
        mi->gather = nullptr;
        mi->gather_words = 0;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&mi->gather), words * 8 * (size_t)mi->g));
        mi->gather_words = words;
    }
    const size_t cells = (size_t)b * k;
    if (host_entry && cells > mi->out_cells) {
        HIP_TRY(hipStreamSynchronize(mi->blocks[0].stream));
        if (mi->out_i) (void)hipFree(mi->out_i);
        if (mi->out_s) (void)hipFree(mi->out_s);
        mi->out_i = nullptr;
        mi->out_s = nullptr;
        mi->out_cells = 0;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&mi->out_i), cells * 8));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&mi->out_s), cells * 4));
        mi->out_cells = cells;
    }
    return MTR_OK;
}

// ids -1, scores -inf: what a block without rows contributes.
int fill_padding(long long* block, int b, int k, hipStream_t stream) {
    const size_t cells = (size_t)b * k;
    HIP_TRY(mtr::launch_fill_u32(block, 0xFFFFFFFFu, cells * 2, stream));
    HIP_TRY(mtr::launch_fill_u32(block + cells, 0xFF800000u, cells, stream));
    return MTR_OK;
}

// Everything between "the queries are on every GPU" and "the merged result is in
// dev_out_* on the first GPU", enqueued on the blocks' streams.  q_of(i): where
// block i finds the queries on its GPU.
template <typename QueryOf>
int enqueue_search(mtr_multi_index* mi, QueryOf q_of, int q_dtype, int b, int k, int metric, long long* dev_out_idx,
                   float* dev_out_score) {
    const size_t words = block_words(b, k);
    const size_t cells = (size_t)b * k;
    int rc;
    // 1. local top-k of every block into its send block
    for (int i = 0; i < mi->g; ++i) {
        auto& blk = mi->blocks[i];
        DeviceGuard guard(blk.device);
        // Copy form: a block on the first GPU writes straight into its slot of the gather buffer.
        long long* dst = blk.send;
        if (mi->exchange == MTR_EXCHANGE_COPY && blk.device == home(mi)) dst = mi->gather + (size_t)i * mi->gather_words;
        if (blk.shard) {
            rc = mtr_index_search_device(blk.shard, q_of(i), q_dtype, b, k, metric, reinterpret_cast<int64_t*>(dst),
                                         reinterpret_cast<float*>(dst + cells), blk.stream);
            if (rc != MTR_OK) return rc;
        } else if ((rc = fill_padding(dst, b, k, blk.stream)) != MTR_OK) {
            return rc;
        }
    }
    // 2. the exchange
    const long long* lists = nullptr;
    long long list_stride = 0;
    if (mi->exchange == MTR_EXCHANGE_RCCL) {
        const mtr::Rccl* r = mtr::rccl();
        RCCL_TRY(r, r->GroupStart());
        ncclResult_t worst = ncclSuccess;
        for (auto& blk : mi->blocks) {
            const ncclResult_t e = r->AllGather(blk.send, blk.recv, words, ncclInt64, blk.comm, blk.stream);
            if (e != ncclSuccess) worst = e;
        }
        const ncclResult_t ge = r->GroupEnd();   // always closed, also after a failed call inside the group
        if (worst != ncclSuccess) return mtr::set_error(MTR_ERR_COMM, "ncclAllGather: %s", r->GetErrorString(worst));
        if (ge != ncclSuccess) return mtr::set_error(MTR_ERR_COMM, "ncclGroupEnd: %s", r->GetErrorString(ge));
        lists = mi->blocks[0].recv;
        list_stride = (long long)words;
    } else {
        for (int i = 1; i < mi->g; ++i) {
            auto& blk = mi->blocks[i];
            DeviceGuard guard(blk.device);
            if (blk.device != home(mi))
                HIP_TRY(hipMemcpyPeerAsync(mi->gather + (size_t)i * mi->gather_words, home(mi), blk.send, blk.device,
                                           words * 8, blk.stream));
            HIP_TRY(hipEventRecord(blk.done, blk.stream));
        }
        DeviceGuard guard(home(mi));
        for (int i = 1; i < mi->g; ++i) HIP_TRY(hipStreamWaitEvent(mi->blocks[0].stream, mi->blocks[i].done, 0));
        lists = mi->gather;
        list_stride = (long long)mi->gather_words;
    }
    // 3. merge on the first GPU
    DeviceGuard guard(home(mi));
    const float* scores = reinterpret_cast<const float*>(lists) + 2 * cells;
    HIP_TRY(mtr::launch_merge_topk(scores, nullptr, lists, mi->g, list_stride, 2 * list_stride, (long long)k, k, b, k, 0,
                                   dev_out_idx, dev_out_score, mi->blocks[0].stream));
    HIP_TRY(hipEventRecord(mi->merged, mi->blocks[0].stream));
    mi->merged_set = true;
    return MTR_OK;
}

// A new search may not overwrite the gather buffer (or a send block) while the
// previous search's merge or exchange reads it: every block's stream first waits
// for that merge.
int wait_for_previous(mtr_multi_index* mi) {
    if (!mi->merged_set) return MTR_OK;
    for (int i = 1; i < mi->g; ++i) {
        DeviceGuard guard(mi->blocks[i].device);
        HIP_TRY(hipStreamWaitEvent(mi->blocks[i].stream, mi->merged, 0));
    }
    return MTR_OK;
}

int check_args(const mtr_multi_index* mi, const void* q, const void* oi, const void* os, int q_dtype, int b, int k,
               int metric) {
    if (!mi) return mtr::set_error(MTR_ERR_INVALID, "index handle is NULL");
    if (!q || !oi || !os) return mtr::set_error(MTR_ERR_INVALID, "NULL argument");
    if (b < 1) return mtr::set_error(MTR_ERR_INVALID, "b must be >= 1");
    if (k < 1 || k > MTR_MAX_K) return mtr::set_error(MTR_ERR_INVALID, "k must be in [1, %d]", MTR_MAX_K);
    if (metric != MTR_COSINE && metric != MTR_IP) return mtr::set_error(MTR_ERR_INVALID, "bad metric %d", metric);
    if (!dtype_bytes(q_dtype)) return mtr::set_error(MTR_ERR_INVALID, "bad q_dtype %d", q_dtype);
    return MTR_OK;
}

// After a failed enqueue: nothing may stay in flight on buffers the caller is
// about to reuse or free.
void drain(mtr_multi_index* mi) {
    for (auto& blk : mi->blocks) {
        DeviceGuard guard(blk.device);
        if (blk.stream) (void)hipStreamSynchronize(blk.stream);
    }
}

}  // namespace

extern "C" {

int mtr_multi_index_alloc(int64_t n, int32_t d, int32_t store_dtype, int32_t normalize, const int32_t* device_ids,
                          int32_t n_devices, int32_t exchange, mtr_multi_index_t** out) {
    if (!out) return mtr::set_error(MTR_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!device_ids || n_devices < 1 || n_devices > 64)
        return mtr::set_error(MTR_ERR_INVALID, "the device list must hold 1 to 64 entries");
    if (n < 1 || d < 1) return mtr::set_error(MTR_ERR_INVALID, "n and d must be positive (n=%lld, d=%d)", (long long)n, d);
    if (exchange != MTR_EXCHANGE_AUTO && exchange != MTR_EXCHANGE_COPY && exchange != MTR_EXCHANGE_RCCL)
        return mtr::set_error(MTR_ERR_INVALID, "bad exchange form %d", exchange);
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    bool distinct = true;
    for (int i = 0; i < n_devices; ++i) {
        if (device_ids[i] < 0 || device_ids[i] >= ndev)
            return mtr::set_error(MTR_ERR_INVALID, "device %d of %d", device_ids[i], ndev);
        for (int j = 0; j < i; ++j) distinct = distinct && device_ids[j] != device_ids[i];
    }
    mtr_multi_index* mi = new (std::nothrow) mtr_multi_index();
    if (!mi) return mtr::set_error(MTR_ERR_OOM, "host allocation failed");
    mi->n = n;
    mi->d = d;
    mi->dtype = store_dtype;
    mi->normalized = normalize ? 1 : 0;
    mi->g = n_devices;
    mi->blocks.resize(n_devices);
    auto fail_with = [&](int rc) {
        const std::string keep = mtr_last_error();
        mtr_multi_index_destroy(mi);
        return mtr::set_error(rc, "%s", keep.c_str());
    };
    const int64_t per = n / n_devices, extra = n % n_devices;
    for (int i = 0; i < n_devices; ++i) {
        auto& blk = mi->blocks[i];
        blk.device = device_ids[i];
        blk.lo = i * per + std::min<int64_t>(i, extra);
        blk.hi = blk.lo + per + (i < extra ? 1 : 0);
        DeviceGuard guard(blk.device);
        if (!guard.ok) return fail_with(mtr::set_error(MTR_ERR_HIP, "hipSetDevice(%d) failed", blk.device));
        if (blk.hi > blk.lo) {
            const int rc = mtr_index_alloc(blk.hi - blk.lo, d, store_dtype, normalize, blk.device, blk.lo, &blk.shard);
            if (rc != MTR_OK) return fail_with(rc);
        }
        hipError_t e = hipStreamCreateWithFlags(&blk.stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&blk.done, hipEventDisableTiming);
        if (e != hipSuccess) return fail_with(mtr::set_error(MTR_ERR_HIP, "block set-up: %s", hipGetErrorString(e)));
    }
    {
        DeviceGuard guard(home(mi));
        hipError_t e = hipEventCreateWithFlags(&mi->q_ready, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&mi->merged, hipEventDisableTiming);
        if (e != hipSuccess) return fail_with(mtr::set_error(MTR_ERR_HIP, "index set-up: %s", hipGetErrorString(e)));
        // direct copies over xGMI between the first GPU and the others (it works without, through staging)
        for (int i = 1; i < n_devices; ++i)
            if (device_ids[i] != home(mi)) {
                (void)hipDeviceEnablePeerAccess(device_ids[i], 0);
                (void)hipGetLastError();
            }
    }
    for (int i = 1; i < n_devices; ++i)
        if (device_ids[i] != home(mi)) {
            DeviceGuard guard(device_ids[i]);
            (void)hipDeviceEnablePeerAccess(home(mi), 0);
            (void)hipGetLastError();
        }
    // The exchange form.  RCCL puts one rank on one device, so a list that names a
    // device twice (a test of the block logic on one GPU) exchanges by copies.
    const mtr::Rccl* r = mtr::rccl();
    const bool rccl_ok = r->handle && r->CommInitAll && r->GroupStart && r->GroupEnd;
    mi->exchange = exchange;
    if (exchange == MTR_EXCHANGE_AUTO)
        mi->exchange = (n_devices > 1 && distinct && rccl_ok) ? MTR_EXCHANGE_RCCL : MTR_EXCHANGE_COPY;
    if (mi->exchange == MTR_EXCHANGE_RCCL) {
        if (!rccl_ok)
            return fail_with(mtr::set_error(MTR_ERR_COMM, "%s", r->handle ? "the loaded RCCL lacks ncclCommInitAll / ncclGroupStart / ncclGroupEnd"
                                                                           : r->why));
        std::vector<ncclComm_t> comms(n_devices, nullptr);
        const ncclResult_t e = r->CommInitAll(comms.data(), n_devices, device_ids);
        if (e != ncclSuccess) {
            if (exchange == MTR_EXCHANGE_AUTO) {
                mi->exchange = MTR_EXCHANGE_COPY;   // peer copies need no communicator
            } else {
                return fail_with(mtr::set_error(MTR_ERR_COMM, "ncclCommInitAll(%d devices): %s", n_devices, r->GetErrorString(e)));
            }
        } else {
            for (int i = 0; i < n_devices; ++i) mi->blocks[i].comm = comms[i];
        }
    }
    *out = mi;
    return MTR_OK;
}

int mtr_multi_index_load_rows(mtr_multi_index_t* mi, int64_t row_start, const void* host_rows, int64_t n_rows,
                              int32_t src_dtype) {
    if (!mi) return mtr::set_error(MTR_ERR_INVALID, "index handle is NULL");
    if (!host_rows && n_rows > 0) return mtr::set_error(MTR_ERR_INVALID, "rows is NULL");
    const int ssize = dtype_bytes(src_dtype & ~MTR_ROWS_AS_STORED);
    if (!ssize) return mtr::set_error(MTR_ERR_INVALID, "bad src_dtype %d", src_dtype);
    if (row_start < 0 || n_rows < 0 || row_start + n_rows > mi->n)
        return mtr::set_error(MTR_ERR_INVALID, "rows [%lld, %lld) outside the index of %lld rows", (long long)row_start,
                              (long long)(row_start + n_rows), (long long)mi->n);
    std::lock_guard<std::mutex> lock(mi->mu);
    // Every block's part goes to its GPU from a host thread of its own: the
    // staged uploads of different GPUs then cross PCIe side by side.
    struct Part {
        mtr_index_t* shard;
        int64_t local, count;
        const char* src;
        int rc = MTR_OK;
        std::string msg;
    };
    std::vector<Part> parts;
    const size_t row_bytes = (size_t)mi->d * ssize;
    for (auto& blk : mi->blocks) {
        const int64_t a = std::max(row_start, blk.lo), z = std::min(row_start + n_rows, blk.hi);
        if (z > a && blk.shard)
            parts.push_back(Part{blk.shard, a - blk.lo, z - a, static_cast<const char*>(host_rows) + (size_t)(a - row_start) * row_bytes});
    }
    auto run = [src_dtype](Part* p) {
        p->rc = mtr_index_load_rows(p->shard, p->local, p->src, p->count, src_dtype, 0);
        if (p->rc != MTR_OK) p->msg = mtr_last_error();
    };
    if (parts.size() == 1) {
        run(&parts[0]);
    } else {
        std::vector<std::thread> threads;
        for (auto& p : parts) threads.emplace_back(run, &p);
        for (auto& t : threads) t.join();
    }
    for (auto& p : parts)
        if (p.rc != MTR_OK) return mtr::set_error(p.rc, "%s", p.msg.c_str());
    return MTR_OK;
}

int mtr_multi_index_create(const void* host_rows, int64_t n, int32_t d, int32_t src_dtype, int32_t store_dtype,
                           int32_t normalize, const int32_t* device_ids, int32_t n_devices, int32_t exchange,
                           mtr_multi_index_t** out) {
    int rc = mtr_multi_index_alloc(n, d, store_dtype, normalize, device_ids, n_devices, exchange, out);
    if (rc != MTR_OK) return rc;
    rc = mtr_multi_index_load_rows(*out, 0, host_rows, n, src_dtype);
    if (rc != MTR_OK) {
        const std::string keep = mtr_last_error();
        mtr_multi_index_destroy(*out);
        *out = nullptr;
        return mtr::set_error(rc, "%s", keep.c_str());
    }
    return MTR_OK;
}

int mtr_multi_index_search(mtr_multi_index_t* mi, const void* queries, int32_t q_dtype, int32_t b, int32_t k,
                           int32_t metric, int64_t* out_idx, float* out_score) {
    int rc = check_args(mi, queries, out_idx, out_score, q_dtype, b, k, metric);
    if (rc != MTR_OK) return rc;
    std::lock_guard<std::mutex> lock(mi->mu);
    const size_t q_bytes = (size_t)b * mi->d * dtype_bytes(q_dtype);
    if ((rc = grow(mi, b, k, q_bytes, true)) != MTR_OK) return rc;
    if ((rc = wait_for_previous(mi)) != MTR_OK) return rc;
    for (auto& blk : mi->blocks) {
        if (!blk.shard) continue;
        DeviceGuard guard(blk.device);
        HIP_TRY(hipMemcpyAsync(blk.q, queries, q_bytes, hipMemcpyHostToDevice, blk.stream));
    }
    rc = enqueue_search(mi, [&](int i) { return (const void*)mi->blocks[i].q; }, q_dtype, b, k, metric, mi->out_i, mi->out_s);
    if (rc != MTR_OK) {
        const std::string keep = mtr_last_error();
        drain(mi);
        return mtr::set_error(rc, "%s", keep.c_str());
    }
    DeviceGuard guard(home(mi));
    hipStream_t s0 = mi->blocks[0].stream;
    HIP_TRY(hipMemcpyAsync(out_idx, mi->out_i, (size_t)b * k * 8, hipMemcpyDeviceToHost, s0));
    HIP_TRY(hipMemcpyAsync(out_score, mi->out_s, (size_t)b * k * 4, hipMemcpyDeviceToHost, s0));
    HIP_TRY(hipStreamSynchronize(s0));
    // every block saw the same queries: the first block's report is everybody's
    int32_t bad = 0;
    if ((rc = mtr_index_nonfinite(mi->blocks[0].shard, &bad)) != MTR_OK) return rc;
    if (bad) return mtr::set_error(MTR_ERR_NONFINITE, "Input contains NaN or infinity.");
    return MTR_OK;
}

int mtr_multi_index_search_device(mtr_multi_index_t* mi, const void* dev_queries, int32_t q_dtype, int32_t b, int32_t k,
                                  int32_t metric, int64_t* dev_out_idx, float* dev_out_score, void* stream) {
    int rc = check_args(mi, dev_queries, dev_out_idx, dev_out_score, q_dtype, b, k, metric);
    if (rc != MTR_OK) return rc;
    std::lock_guard<std::mutex> lock(mi->mu);
    const size_t q_bytes = (size_t)b * mi->d * dtype_bytes(q_dtype);
    if ((rc = grow(mi, b, k, q_bytes, false)) != MTR_OK) return rc;
    if ((rc = wait_for_previous(mi)) != MTR_OK) return rc;
    hipStream_t cs = static_cast<hipStream_t>(stream);
    {
        DeviceGuard guard(home(mi));
        HIP_TRY(hipEventRecord(mi->q_ready, cs));
    }
    for (auto& blk : mi->blocks) {
        DeviceGuard guard(blk.device);
        HIP_TRY(hipStreamWaitEvent(blk.stream, mi->q_ready, 0));
        if (blk.shard && blk.device != home(mi))
            HIP_TRY(hipMemcpyPeerAsync(blk.q, blk.device, dev_queries, home(mi), q_bytes, blk.stream));
    }
    rc = enqueue_search(mi, [&](int i) { return mi->blocks[i].device == home(mi) ? dev_queries : (const void*)mi->blocks[i].q; },
                        q_dtype, b, k, metric, reinterpret_cast<long long*>(dev_out_idx), dev_out_score);
    if (rc != MTR_OK) {
        const std::string keep = mtr_last_error();
        drain(mi);
        return mtr::set_error(rc, "%s", keep.c_str());
    }
    DeviceGuard guard(home(mi));
    HIP_TRY(hipStreamWaitEvent(cs, mi->merged, 0));   // the caller's stream continues behind the merge
    return MTR_OK;
}

int mtr_multi_index_nonfinite(mtr_multi_index_t* mi, int32_t* nonfinite) {
    if (!mi) return mtr::set_error(MTR_ERR_INVALID, "index handle is NULL");
    std::lock_guard<std::mutex> lock(mi->mu);
    return mtr_index_nonfinite(mi->blocks[0].shard, nonfinite);
}

int mtr_multi_index_info(const mtr_multi_index_t* mi, int64_t* n, int32_t* d, int32_t* store_dtype, int32_t* normalized,
                         int32_t* n_blocks, int32_t* exchange) {
    if (!mi) return mtr::set_error(MTR_ERR_INVALID, "index handle is NULL");
    if (n) *n = mi->n;
    if (d) *d = mi->d;
    if (store_dtype) *store_dtype = mi->dtype;
    if (normalized) *normalized = mi->normalized;
    if (n_blocks) *n_blocks = mi->g;
    if (exchange) *exchange = mi->exchange;
    return MTR_OK;
}

int mtr_multi_index_block(mtr_multi_index_t* mi, int32_t i, mtr_index_t** shard, int32_t* device_id, int64_t* row_lo,
                          int64_t* row_hi) {
    if (!mi) return mtr::set_error(MTR_ERR_INVALID, "index handle is NULL");
    if (i < 0 || i >= mi->g) return mtr::set_error(MTR_ERR_INVALID, "block %d of %d", i, mi->g);
    const auto& blk = mi->blocks[i];
    if (shard) *shard = blk.shard;
    if (device_id) *device_id = blk.device;
    if (row_lo) *row_lo = blk.lo;
    if (row_hi) *row_hi = blk.hi;
    return MTR_OK;
}

int mtr_multi_index_read_rows(mtr_multi_index_t* mi, int64_t row_start, int64_t n_rows, float* host_out) {
    if (!mi) return mtr::set_error(MTR_ERR_INVALID, "index handle is NULL");
    if (!host_out || row_start < 0 || n_rows < 0 || row_start + n_rows > mi->n)
        return mtr::set_error(MTR_ERR_INVALID, "bad row range");
    std::lock_guard<std::mutex> lock(mi->mu);
    for (auto& blk : mi->blocks) {
        const int64_t a = std::max(row_start, blk.lo), z = std::min(row_start + n_rows, blk.hi);
        if (z <= a || !blk.shard) continue;
        const int rc = mtr_index_read_rows(blk.shard, a - blk.lo, z - a, host_out + (size_t)(a - row_start) * mi->d);
        if (rc != MTR_OK) return rc;
    }
    return MTR_OK;
}

void mtr_multi_index_destroy(mtr_multi_index_t* mi) {
    if (!mi) return;
    for (auto& blk : mi->blocks) {
        DeviceGuard guard(blk.device);
        if (blk.stream) (void)hipStreamSynchronize(blk.stream);
    }
    for (auto& blk : mi->blocks) {
        DeviceGuard guard(blk.device);
        if (blk.comm) (void)mtr::rccl()->CommDestroy(blk.comm);
        if (blk.shard) mtr_index_destroy(blk.shard);
        if (blk.q) (void)hipFree(blk.q);
        if (blk.send) (void)hipFree(blk.send);
        if (blk.recv) (void)hipFree(blk.recv);
        if (blk.done) (void)hipEventDestroy(blk.done);
        if (blk.stream) (void)hipStreamDestroy(blk.stream);
    }
    if (!mi->blocks.empty()) {
        DeviceGuard guard(home(mi));
        if (mi->gather) (void)hipFree(mi->gather);
        if (mi->out_i) (void)hipFree(mi->out_i);
        if (mi->out_s) (void)hipFree(mi->out_s);
        if (mi->q_ready) (void)hipEventDestroy(mi->q_ready);
        if (mi->merged) (void)hipEventDestroy(mi->merged);
    }
    delete mi;
}

}  // extern "C"
