// The exchange step of the row-sharded search behind the C ABI: one process per
// GPU, each with its shard; a search is  local top-k  ->  ONE RCCL all-gather of
// the packed (id, score) candidates over xGMI  ->  k-way merge on every rank,
// all enqueued on one stream.
//
// The reference is single-process and single-shard: its whole "merge" is the
// argmax at pipeline/utils/local_rag.py:78.  This file is what lets a host in
// any language search a corpus that spans the GPUs of a node.
//
// RCCL is loaded on first use (dlopen), so a single-GPU host needs no RCCL
// installation; when PyTorch already has its RCCL mapped, that copy is used.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>

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

}  // namespace

namespace mtr {

// The library, resolved once.  An RCCL that the process already has mapped
// (PyTorch ships one) is preferred over loading a second copy.
const Rccl* rccl() {
    static Rccl r = [] {
        Rccl v;
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        // MTR_RCCL_LIBRARY: the one to use, by path (a particular RCCL build; the tests' stand-in
        // for several ranks on one GPU)
        const char* chosen = getenv("MTR_RCCL_LIBRARY");
        if (chosen && *chosen) {
            v.handle = dlopen(chosen, RTLD_NOW | RTLD_LOCAL);
            if (!v.handle) {
                snprintf(v.why, sizeof(v.why), "cannot load MTR_RCCL_LIBRARY=%s: %s", chosen, dlerror());
                return v;
            }
        }
        for (const char* n : names)
            if (!v.handle) v.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        for (const char* n : names)
            if (!v.handle) v.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (!v.handle) {
            snprintf(v.why, sizeof(v.why), "cannot load librccl.so.1: %s", dlerror());
            return v;
        }
        v.GetUniqueId = reinterpret_cast<decltype(v.GetUniqueId)>(dlsym(v.handle, "ncclGetUniqueId"));
        v.CommInitRank = reinterpret_cast<decltype(v.CommInitRank)>(dlsym(v.handle, "ncclCommInitRank"));
        v.CommDestroy = reinterpret_cast<decltype(v.CommDestroy)>(dlsym(v.handle, "ncclCommDestroy"));
        v.AllGather = reinterpret_cast<decltype(v.AllGather)>(dlsym(v.handle, "ncclAllGather"));
        v.GetErrorString = reinterpret_cast<decltype(v.GetErrorString)>(dlsym(v.handle, "ncclGetErrorString"));
        if (!v.GetUniqueId || !v.CommInitRank || !v.CommDestroy || !v.AllGather || !v.GetErrorString) {
            snprintf(v.why, sizeof(v.why), "the loaded RCCL lacks a needed entry point");
            v.handle = nullptr;
            return v;
        }
        // one process driving several GPUs (multi_index.hip); absent from a library: that form is unavailable
        v.CommInitAll = reinterpret_cast<decltype(v.CommInitAll)>(dlsym(v.handle, "ncclCommInitAll"));
        v.GroupStart = reinterpret_cast<decltype(v.GroupStart)>(dlsym(v.handle, "ncclGroupStart"));
        v.GroupEnd = reinterpret_cast<decltype(v.GroupEnd)>(dlsym(v.handle, "ncclGroupEnd"));
        return v;
    }();
    return &r;
}

}  // namespace mtr

namespace {

using mtr::Rccl;
using mtr::rccl;

#define RCCL_TRY(r, expr)                                                                              \
    do {                                                                                               \
        ncclResult_t n_ = (expr);                                                                      \
        if (n_ != ncclSuccess) return mtr::set_error(MTR_ERR_COMM, "%s: %s", #expr, (r)->GetErrorString(n_)); \
    } while (0)

constexpr int kSlots = 4;  // exchanges that may be in flight at once on different streams

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

}  // namespace

// One rank's end of the communicator plus its exchange buffers.  A block (what
// one rank contributes for one search) is b*k int64 ids followed by b*k fp32
// scores, padded to whole 8-byte words; the all-gather leaves the blocks of all
// ranks one after the other, in rank order, which is row order.
struct mtr_comm {
    int device = 0, rank = 0, world = 1;
    ncclComm_t comm = nullptr;
    std::mutex mu;
    struct Slot {
        long long* send = nullptr;
        long long* recv = nullptr;
        size_t words = 0;   // capacity of send; recv holds world times as much
        hipEvent_t done = nullptr;
        hipStream_t last = nullptr;
        bool used = false;
    } slots[kSlots];
    int next = 0;
    // Set when this rank could not take part in a collective its peers have
    // entered (no memory for the exchange buffers): the ranks' sequences of
    // all-gathers no longer match, so every later call fails at once.
    bool poisoned = false;
};

namespace {

size_t block_words(int b, int k) { return (size_t)b * k + ((size_t)b * k + 1) / 2; }

// The next buffer set, large enough for (b, k), safe to overwrite from `stream`.
int take_slot_unguarded(mtr_comm* c, int b, int k, hipStream_t stream, mtr_comm::Slot** out) {
    mtr_comm::Slot& s = c->slots[c->next];
    c->next = (c->next + 1) % kSlots;
    const size_t words = block_words(b, k);
    if (s.used && s.last != stream) HIP_TRY(hipStreamWaitEvent(stream, s.done, 0));
    if (words > s.words) {
        if (s.used) HIP_TRY(hipEventSynchronize(s.done));
        if (s.send) (void)hipFree(s.send);
        if (s.recv) (void)hipFree(s.recv);
        s.send = s.recv = nullptr;
        s.words = 0;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s.send), words * 8));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s.recv), words * 8 * (size_t)c->world));
        s.words = words;
    }
    if (!s.done) HIP_TRY(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
    *out = &s;
    return MTR_OK;
}

// Without buffers this rank cannot enter the all-gather the other ranks are
// about to enter: the communicator is of no use from here on.
int take_slot(mtr_comm* c, int b, int k, hipStream_t stream, mtr_comm::Slot** out) {
    if (c->poisoned)
        return mtr::set_error(MTR_ERR_COMM, "the communicator missed a collective after an earlier failure on this rank: "
                                            "destroy it on every rank");
    const int rc = take_slot_unguarded(c, b, k, stream, out);
    if (rc != MTR_OK) c->poisoned = true;
    return rc;
}

// All-gather of the slot's send block and the merge of every rank's block.
int exchange_and_merge(mtr_comm* c, mtr_comm::Slot* s, int b, int k_in, int k_out, long long* dev_out_idx,
                       float* dev_out_score, hipStream_t stream) {
    const Rccl* r = rccl();
    const size_t words = block_words(b, k_in);
    RCCL_TRY(r, r->AllGather(s->send, s->recv, words, ncclInt64, c->comm, stream));
    const float* scores = reinterpret_cast<const float*>(s->recv) + 2 * (size_t)b * k_in;
    HIP_TRY(mtr::launch_merge_topk(scores, nullptr, s->recv, c->world, (long long)words, 2 * (long long)words,
                                   (long long)k_in, k_in, b, k_out, 0, dev_out_idx, dev_out_score, stream));
    HIP_TRY(hipEventRecord(s->done, stream));
    s->last = stream;
    s->used = true;
    return MTR_OK;
}

// A block of padding: ids -1, scores -inf (what a rank without rows contributes).
int fill_padding(mtr_comm::Slot* s, int b, int k, hipStream_t stream) {
    const size_t cells = (size_t)b * k;
    HIP_TRY(mtr::launch_fill_u32(s->send, 0xFFFFFFFFu, cells * 2, stream));
    HIP_TRY(mtr::launch_fill_u32(s->send + cells, 0xFF800000u, cells, stream));
    return MTR_OK;
}

int check_shape(const mtr_comm* c, int b, int k_in, int k_out, const void* out_idx, const void* out_score) {
    if (!c) return mtr::set_error(MTR_ERR_INVALID, "communicator handle is NULL");
    if (!out_idx || !out_score) return mtr::set_error(MTR_ERR_INVALID, "NULL output");
    if (b < 1 || k_in < 1 || k_out < 1 || k_out > MTR_MAX_K || k_in > MTR_MAX_K)
        return mtr::set_error(MTR_ERR_INVALID, "bad exchange shape (b=%d, k_in=%d, k_out=%d)", b, k_in, k_out);
    return MTR_OK;
}

}  // namespace

extern "C" {

int mtr_comm_unique_id(void* out_id) {
    if (!out_id) return mtr::set_error(MTR_ERR_INVALID, "out_id is NULL");
    const Rccl* r = rccl();
    if (!r->handle) return mtr::set_error(MTR_ERR_COMM, "%s", r->why);
    static_assert(sizeof(ncclUniqueId) == MTR_COMM_ID_BYTES, "MTR_COMM_ID_BYTES must be RCCL's unique-id size");
    ncclUniqueId id;
    RCCL_TRY(r, r->GetUniqueId(&id));
    memcpy(out_id, &id, sizeof(id));
    return MTR_OK;
}

int mtr_comm_create(const void* unique_id, int32_t rank, int32_t world, int32_t device_id, mtr_comm_t** out) {
    if (!out) return mtr::set_error(MTR_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!unique_id || world < 1 || rank < 0 || rank >= world)
        return mtr::set_error(MTR_ERR_INVALID, "bad communicator arguments (rank %d of %d)", rank, world);
    const Rccl* r = rccl();
    if (!r->handle) return mtr::set_error(MTR_ERR_COMM, "%s", r->why);
    DeviceGuard guard(device_id);
    if (!guard.ok) return mtr::set_error(MTR_ERR_HIP, "hipSetDevice(%d) failed", device_id);
    mtr_comm* c = new (std::nothrow) mtr_comm();
    if (!c) return mtr::set_error(MTR_ERR_OOM, "host allocation failed");
    c->device = device_id;
    c->rank = rank;
    c->world = world;
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclResult_t n = r->CommInitRank(&c->comm, world, id, rank);
    if (n != ncclSuccess) {
        delete c;
        return mtr::set_error(MTR_ERR_COMM, "ncclCommInitRank(rank %d of %d): %s", rank, world, r->GetErrorString(n));
    }
    *out = c;
    return MTR_OK;
}

void mtr_comm_destroy(mtr_comm_t* c) {
    if (!c) return;
    {
        DeviceGuard guard(c->device);
        for (auto& s : c->slots) {
            if (s.used) (void)hipEventSynchronize(s.done);
            if (s.done) (void)hipEventDestroy(s.done);
            if (s.send) (void)hipFree(s.send);
            if (s.recv) (void)hipFree(s.recv);
        }
        if (c->comm) (void)rccl()->CommDestroy(c->comm);
    }
    delete c;
}

int mtr_comm_info(const mtr_comm_t* c, int32_t* rank, int32_t* world, int32_t* device_id) {
    if (!c) return mtr::set_error(MTR_ERR_INVALID, "communicator handle is NULL");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    if (device_id) *device_id = c->device;
    return MTR_OK;
}

int mtr_comm_exchange_merge_device(mtr_comm_t* c, const int64_t* dev_local_idx, const float* dev_local_score, int32_t b,
                                   int32_t k_in, int32_t k_out, int64_t* dev_out_idx, float* dev_out_score,
                                   void* stream) {
    int rc = check_shape(c, b, k_in, k_out, dev_out_idx, dev_out_score);
    if (rc != MTR_OK) return rc;
    if ((dev_local_idx == nullptr) != (dev_local_score == nullptr))
        return mtr::set_error(MTR_ERR_INVALID, "local ids and scores must both be given or both be NULL");
    std::lock_guard<std::mutex> lock(c->mu);
    DeviceGuard guard(c->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    mtr_comm::Slot* slot = nullptr;
    if ((rc = take_slot(c, b, k_in, s, &slot)) != MTR_OK) return rc;
    const size_t cells = (size_t)b * k_in;
    auto stage = [&]() -> int {
        if (!dev_local_idx) return fill_padding(slot, b, k_in, s);
        HIP_TRY(hipMemcpyAsync(slot->send, dev_local_idx, cells * 8, hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipMemcpyAsync(slot->send + cells, dev_local_score, cells * 4, hipMemcpyDeviceToDevice, s));
        return MTR_OK;
    };
    if ((rc = stage()) != MTR_OK) {
        c->poisoned = true;   // this rank stays out of an all-gather its peers enter
        return rc;
    }
    return exchange_and_merge(c, slot, b, k_in, k_out, reinterpret_cast<long long*>(dev_out_idx), dev_out_score, s);
}

int mtr_sharded_search_device(mtr_index_t* index, mtr_comm_t* c, const void* dev_queries, int32_t q_dtype, int32_t b,
                              int32_t k, int32_t metric, int64_t* dev_out_idx, float* dev_out_score, void* stream) {
    int rc = check_shape(c, b, k, k, dev_out_idx, dev_out_score);
    if (rc != MTR_OK) return rc;
    std::lock_guard<std::mutex> lock(c->mu);
    DeviceGuard guard(c->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    mtr_comm::Slot* slot = nullptr;
    if ((rc = take_slot(c, b, k, s, &slot)) != MTR_OK) return rc;
    const size_t cells = (size_t)b * k;
    // A failure of this rank's own search must not keep it out of the all-gather
    // that the other ranks have entered or will enter: it then contributes a block
    // of padding, runs the exchange and the merge like everybody else, and
    // returns its error afterwards.
    int local_rc = MTR_OK;
    std::string local_msg;
    if (index) {
        int32_t dev = -1;
        local_rc = mtr_index_info(index, nullptr, nullptr, nullptr, nullptr, &dev, nullptr);
        if (local_rc == MTR_OK && dev != c->device)
            local_rc = mtr::set_error(MTR_ERR_INVALID, "the shard is on GPU %d, the communicator on GPU %d", dev, c->device);
        // the local top-k goes straight into the send block
        if (local_rc == MTR_OK)
            local_rc = mtr_index_search_device(index, dev_queries, q_dtype, b, k, metric,
                                               reinterpret_cast<int64_t*>(slot->send),
                                               reinterpret_cast<float*>(slot->send + cells), stream);
        if (local_rc != MTR_OK) local_msg = mtr_last_error();
    }
    if (!index || local_rc != MTR_OK) {   // a rank without rows (or without a result) still takes part
        if ((rc = fill_padding(slot, b, k, s)) != MTR_OK) {
            c->poisoned = true;
            return rc;
        }
    }
    rc = exchange_and_merge(c, slot, b, k, k, reinterpret_cast<long long*>(dev_out_idx), dev_out_score, s);
    if (local_rc != MTR_OK) return mtr::set_error(local_rc, "%s", local_msg.c_str());
    return rc;
}

}  // extern "C"
