// C ABI of the retrieval library (declared in include/medtutor_retrieval.h):
// shard handles, the row loader, search and candidate merge.  Host-side only;
// the kernels are in scan_gemv.hip (single queries), score_mfma.hip (batches),
// merge_topk.hip, load_rows.hip and pool_normalize.hip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/medtutor_retrieval.h"
#include "mtr_device.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

}  // namespace

namespace mtr {
// The same message slot for the library's other source files.
int set_error(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
}  // namespace mtr

namespace {

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return fail(e_ == hipErrorOutOfMemory ? MTR_ERR_OOM : MTR_ERR_HIP, "%s: %s", #expr, \
                        hipGetErrorString(e_));                                            \
    } while (0)

int dtype_size(int dt) { return dt == MTR_F32 ? 4 : dt == MTR_F64 ? 8 : dt == MTR_BF16 ? 2 : 0; }

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return MTR_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = std::max(bytes, (size_t)4096);
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            p = nullptr;
            return fail(e == hipErrorOutOfMemory ? MTR_ERR_OOM : MTR_ERR_HIP, "hipMalloc(%zu): %s", want,
                        hipGetErrorString(e));
        }
        cap = want;
        return MTR_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

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

// Scratch memory of the allocation-free entry point
// (mtr_index_search_device_noalloc): a file-scope device array instead of
// buffers from hipMalloc.  Every GPU the library runs on holds its own copy, at
// its own address, until the process ends.  The name is a host-side stand-in
// for the symbol, not a device address: host code only ever uses the per-device
// address that hipGetSymbolAddress returned (static_scratch_for).  That host
// reference is also what makes the compiler give the device-side symbol
// external visibility, which the runtime's lookup needs (marking the array
// `used` instead keeps it local, and the lookup then fails).
constexpr size_t kStaticScratchBytes = (size_t)MTR_NOALLOC_SCRATCH_BYTES;
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mu;
unsigned char* g_scratch_ptr[kMaxDevices] = {};

// The calling thread's current device must be `device`.
int static_scratch_for(int device, unsigned char** out) {
    if (device < 0 || device >= kMaxDevices) return fail(MTR_ERR_INVALID, "device %d out of range", device);
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[device]) {
        void* ptr = nullptr;
        HIP_TRY(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
        if (!ptr) return fail(MTR_ERR_HIP, "hipGetSymbolAddress returned NULL for the scratch array on device %d", device);
        g_scratch_ptr[device] = static_cast<unsigned char*>(ptr);
    }
    *out = g_scratch_ptr[device];
    return MTR_OK;
}
}  // namespace

// The device buffers a search needs beyond its inputs and outputs.
enum ScratchSlot {
    kQF32,        // padded fp32 queries of one scan pass
    kCandS,       // per-workgroup / per-range candidate scores ...
    kCandI,       // ... and rows
    kQTiles,      // query tiles (storage type) for the matrix-core kernels
    kMfmaCand,    // the matrix-core kernel's per-workgroup candidate buffers
    kMfmaFloor,   // per-query floor shared by the row ranges
    kTileMax,     // the sampling launch's best score per (query, sampled tile)
    kWideKeys,    // the 256 x 256-tile kernel's key lists ...
    kWideCounts,  // ... their counts, and one more word: the overflow flag
    kScratchSlots
};

struct mtr_index {
    int device = 0;
    int64_t n = 0;
    int d = 0;
    int ld = 0;  // stored row stride in elements
    int dtype = MTR_F32;
    int normalized = 0;
    int64_t row_offset = 0;
    int num_cus = 256;
    void* rows = nullptr;
    int* flag = nullptr;  // device word that kernels raise to the call's epoch on NaN/inf (mtr::NonfiniteFlag)
    int epoch = 0;        // last value handed out; the word never needs clearing
    int device_epoch = 0; // ... and the one of the last search enqueued without waiting
    // The allocation-free entry point may run under stream capture, where a value
    // passed to a kernel is frozen into the graph: it uses the block's second
    // word instead, cleared by a (captured) fill kernel at the start of every call
    // and raised to 1, so that every replay reports its own queries.
    bool last_device_noalloc = false;
    hipStream_t stream = nullptr;
    std::mutex mu;
    DevBuf scratch[kScratchSlots];  // grown on demand, kept between searches
    int64_t q_init_rows = 0;   // rows [0, q_init_rows) of scratch[kQTiles] have been written or zeroed at least once ...
    int64_t q_data_rows = 0;   // ... and only rows [0, q_data_rows) of them can hold non-zero data
    DevBuf q_in;       // staging for host queries
    DevBuf out_i, out_s;
    DevBuf stage[2];   // device staging for host rows (two: chunk i+1 is copied while chunk i is converted)
    void* pinned[2] = {nullptr, nullptr};  // page-locked host staging for the same chunks
    size_t pinned_cap = 0;
    hipEvent_t stage_done[2] = {nullptr, nullptr};
    // The scratch buffers are shared by every search on the handle, whatever
    // stream it is enqueued on: the event marks the end of the last enqueued
    // work, and work on another stream waits for it first.
    hipEvent_t last_use = nullptr;
    hipStream_t last_stream = nullptr;
    bool used = false;
    unsigned char* static_scratch = nullptr;  // this device's copy of the file-scope scratch array
    const char* last_kernel = "";             // scoring kernel of the last enqueued search
    // profiling
    bool profiling = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    size_t events_used = 0;
};

namespace {

constexpr size_t kStageBytes = 64u << 20;

int check_index(const mtr_index* ix) {
    if (!ix) return fail(MTR_ERR_INVALID, "index handle is NULL");
    return MTR_OK;
}

// Where a search takes its scratch buffers from: the handle's own buffers,
// grown with hipMalloc on demand and kept for the next search.
struct Scratch {
    mtr_index* ix = nullptr;
    void* p[kScratchSlots] = {};
    int need(ScratchSlot slot, size_t bytes) {
        int rc = ix->scratch[slot].reserve(bytes);
        p[slot] = ix->scratch[slot].p;
        return rc;
    }
    int ready() const { return MTR_OK; }
    // The query-tile buffer keeps its contents between searches, so only rows
    // never initialised, or holding an earlier, larger batch, need clearing.
    bool keeps_query_tiles() const { return true; }
    bool may_profile() const { return true; }
    bool allows_wide() const { return true; }
};

// ... or, for the allocation-free entry point, from the device's static array:
// a bump allocation per call, nothing kept from one call to the next.
struct StaticScratch {
    mtr_index* ix = nullptr;
    void* p[kScratchSlots] = {};
    size_t used = 0;
    int need(ScratchSlot slot, size_t bytes) {
        used = (used + 255) & ~(size_t)255;
        if (used + bytes <= kStaticScratchBytes) p[slot] = ix->static_scratch + used;
        used += bytes;
        return MTR_OK;
    }
    int ready() const {
        if (used > kStaticScratchBytes)
            return fail(MTR_ERR_INVALID,
                        "this search needs %zu bytes of scratch, the allocation-free entry point has %zu: use a smaller "
                        "batch or k, or mtr_index_search_device",
                        used, kStaticScratchBytes);
        return MTR_OK;
    }
    bool keeps_query_tiles() const { return false; }
    bool may_profile() const { return false; }   // creating events allocates
    bool allows_wide() const { return false; }   // its key lists run to gigabytes
};

// The words of the handle's flag block through which the waves of the streaming
// single-query scan claim rows (words 32..63 of 64; zero between launches:
// the last wave of a launch clears them).
unsigned int* scan_sync(mtr_index* ix) { return reinterpret_cast<unsigned int*>(ix->flag) + 32; }

// Makes `stream` wait for whatever was last enqueued on the handle from another
// stream (the scratch buffers, the non-finite flag and the staging buffers are
// shared), and marks the end of the newly enqueued work.
int begin_use(mtr_index* ix, hipStream_t stream) {
    if (ix->used && stream != ix->last_stream) HIP_TRY(hipStreamWaitEvent(stream, ix->last_use, 0));
    return MTR_OK;
}
int end_use(mtr_index* ix, hipStream_t stream) {
    HIP_TRY(hipEventRecord(ix->last_use, stream));
    ix->last_stream = stream;
    ix->used = true;
    return MTR_OK;
}

// Smallest batch that goes to the matrix-core kernel; below it the HBM-bound
// scan handles the batch in one pass of up to four queries.
// Measured hand-over (1M rows, ms per search): bf16 x 1024 -- scan 0.315 for one
// query, 0.349 / 0.360 / 0.367 for two / three / four, matrix cores (32-query
// tiles) 0.349 for two to sixteen; fp32 x 768 -- scan 0.46 / 0.49 / 0.51 / 0.51 for
// one to four and 0.55-0.58 for five to eight (one pass of eight), matrix cores
// (16-query tiles of v_mfma_f32_16x16x4_f32) 0.51 for two to sixteen.
int mfma_min_batch(int dtype) {
    const int forced = mtr::tuning().mfma_min_batch;
    if (forced > 0) return forced;
    return dtype == MTR_BF16 ? 2 : 5;
}

// Every how many row tiles one goes into the sampling launch (< 2: no sample).
// The sample costs 1/stride of a full launch; what the full launch appends
// shrinks as k * n / sample.  A sample of about 5 * sqrt(k * n) rows balanced
// the two for the wide kernel (whose sample is matrix-core work like its full
// launch) on 1M and 12.5M rows at k 10 and 100.  The general kernel's appends
// cost its workgroups time in proportion to the queries of a tile, while its
// sample only reads rows: with `b` > 0 (the queries of one tile) the sample grows
// with sqrt(b), up to 2.5 times (12.5M x 768 bf16, k 10, whole search at strides
// 223 / 128 / 90 / 45: 64 queries 2.828 / 2.822 / 2.817 / 2.840 ms, 32 queries
// 2.747 / 2.743 / 2.736 / 2.759, 8 queries 2.724 / 2.735 / 2.729 / 2.758).
// The sample needs at least 2k tiles (its floor is the k-th largest per-tile best
// score) and at most kMaxSampleTiles.
constexpr double kMinTwoPassBytes = 1.0e9;
int sample_stride(long long n, long long n_tiles, int k, int b = 0) {
    const int forced = mtr::tuning().mfma_sample_stride;
    const double div = b > 0 ? std::min(12.5, std::max(5.0, 2.2 * std::sqrt((double)b))) : 5.0;
    long long s = forced >= 0 ? forced : (long long)(std::sqrt((double)n / (double)k) / div);
    if (forced < 0) s = std::min<long long>(1024, std::max<long long>(4, s));
    if (s < 2) return 0;
    s = std::max(s, (n_tiles + mtr::kMaxSampleTiles - 1) / mtr::kMaxSampleTiles);
    s = std::min(s, n_tiles / (2LL * k));
    return s >= 2 ? (int)s : 0;
}

// One event pair of the per-launch profile (created on first use, then reused).
int next_event_pair(mtr_index* ix, hipEvent_t* e0, hipEvent_t* e1) {
    if (ix->events_used == ix->events.size()) {
        hipEvent_t a, c;
        HIP_TRY(hipEventCreate(&a));
        HIP_TRY(hipEventCreate(&c));
        ix->events.emplace_back(a, c);
    }
    *e0 = ix->events[ix->events_used].first;
    *e1 = ix->events[ix->events_used].second;
    ix->events_used++;
    return MTR_OK;
}

// Matrix-core path: queries converted to the shard's type, fused score + top-k
// launch, then the merge of the per-range lists.
template <typename S>
int enqueue_mfma(mtr_index* ix, S& sc, const mtr::MfmaPlan& plan, const void* dev_queries, int q_dtype, int b, int k,
                 int metric, mtr::NonfiniteFlag nf, long long* dev_out_idx, float* dev_out_score, hipStream_t stream) {
    int rc;
    // Large shards get a sampling launch first: every stride-th row tile is
    // scored and only each query's best score per tile is kept.  The k-th
    // largest of those is a lower bound of the query's final k-th best score
    // (k different rows reach it), so the full launch starts from that floor
    // instead of collecting and re-selecting the first tiles of every row range.
    // The extra launches cost about 0.05 ms: not worth it on shards that one
    // launch scores in a few tenths of a ms.
    const bool big = mtr::tuning().mfma_sample_stride >= 0 ||
                     (double)ix->n * ix->ld * dtype_size(ix->dtype) >= kMinTwoPassBytes;
    // Large batches on bf16 shards: the 256 x 256-tile kernel, first in its
    // sampling form over every wstride-th of ITS tiles (floors and filtered
    // scores are then the same arithmetic bit for bit), then as a filter that
    // appends every score that reaches its query's floor (about k * wstride per
    // query: the floor is the k-th largest of n_tiles / wstride tile maxima); the
    // general kernel is enqueued behind it and runs only if a key list overflowed.
    mtr::WidePlan wide, wide_samp;
    int wstride = 0;
    bool use_wide = false;
    if (big && ix->dtype == MTR_BF16 && b >= mtr::tuning().wide_min_batch && sc.allows_wide()) {
        const long long wtiles = (ix->n + 255) / 256;
        wstride = sample_stride(ix->n, wtiles, k);
        if (wstride >= 2) {
            const long long n_wsamp = (wtiles + wstride - 1) / wstride;
            const int hits = (int)std::min<long long>(1 << 24, (long long)std::max(4 * k, 20) * wstride);
            use_wide = n_wsamp <= mtr::kMaxSampleTiles &&
                       mtr::plan_wide(ix->dtype, ix->n, ix->ld, b, k, hits, ix->num_cus, &wide) == 0 &&
                       mtr::plan_wide(ix->dtype, n_wsamp * 256, ix->ld, b, k, 0, ix->num_cus, &wide_samp) == 0;
        }
    }
    // Otherwise the general kernel samples for itself (the fallback behind the
    // wide kernel starts without floors).
    // The sampling launch works on 128-row tiles whatever the full launch uses.
    const long long stiles = (ix->n + 127) / 128;
    // The form on 16 x 16 x 32 MFMAs (plan.via_regs == 2) rounds a sum like the filter kernel, not like
    // the general kernel's other forms: its floors come from the filter kernel's sampling form
    // (256-row tiles), so that the k rows behind a floor reach it again in the full launch.
    mtr::WidePlan regs_samp;
    int rstride = 0;
    if (big && !use_wide && plan.via_regs == 2) {
        const long long wtiles = (ix->n + 255) / 256;
        rstride = sample_stride(ix->n, wtiles, k, std::min(b, plan.tnq));
        if (rstride >= 2) {
            const long long n_wsamp = (wtiles + rstride - 1) / rstride;
            // (a forced combination of switches the sampling form has no shape for: search without floors)
            if (n_wsamp > mtr::kMaxSampleTiles ||
                mtr::plan_wide(ix->dtype, n_wsamp * 256, ix->ld, b, k, 0, ix->num_cus, &regs_samp) != 0 || regs_samp.n_qtiles != 1)
                rstride = 0;
        } else {
            rstride = 0;
        }
    }
    const int stride = (big && !use_wide && plan.via_regs != 2) ? sample_stride(ix->n, stiles, k, std::min(b, plan.tnq)) : 0;
    mtr::MfmaPlan samp = plan;
    if (stride >= 2) {
        const long long n_samp = (stiles + stride - 1) / stride;
        if (mtr::plan_mfma(ix->dtype, n_samp * 128, ix->ld, b, k, ix->num_cus, 1, &samp) != 0 || samp.tm != 128)
            return fail(MTR_ERR_INVALID, "internal: no launch shape for the sampling launch");
        // (its query tiles may be narrower than the full launch's: both index queries and floors by query number)
    }
    const size_t q_row_bytes = (size_t)ix->ld * dtype_size(ix->dtype);
    // Queries padded to whole tiles: the sampling launch may cut the batch into tiles of another
    // width than the full launch (fewer rows, narrower tiles), and its padded count can be the larger.
    const int padded_q = std::max(plan.n_qtiles * plan.tnq, stride >= 2 ? samp.n_qtiles * samp.tnq : 0);
    const int64_t q_rows = std::max<int64_t>((int64_t)padded_q, use_wide ? (int64_t)wide.n_qtiles * wide.tq : rstride ? regs_samp.tq : 0);
    const size_t qbytes = (size_t)q_rows * q_row_bytes;
    const void* q_old = sc.keeps_query_tiles() ? ix->scratch[kQTiles].p : nullptr;
    if ((rc = sc.need(kQTiles, qbytes)) != MTR_OK) return rc;
    if ((rc = sc.need(kMfmaCand, plan.cand_bytes)) != MTR_OK) return rc;
    if ((rc = sc.need(kCandS, (size_t)b * plan.n_ranges * k * 4)) != MTR_OK) return rc;
    if ((rc = sc.need(kCandI, (size_t)b * plan.n_ranges * k * 4)) != MTR_OK) return rc;
    const int floor_words = std::max(padded_q, use_wide ? wide.n_qtiles * wide.tq : 0);
    if ((rc = sc.need(kMfmaFloor, (size_t)floor_words * 4)) != MTR_OK) return rc;
    size_t tile_max_bytes = 0;
    if (use_wide) {
        if ((rc = sc.need(kWideKeys, wide.key_bytes)) != MTR_OK) return rc;
        if ((rc = sc.need(kWideCounts, (wide.count_words + 1) * 4)) != MTR_OK) return rc;
        tile_max_bytes = (size_t)wide.n_qtiles * wide.tq * wide_samp.n_tiles * 4;
    }
    if (stride >= 2) tile_max_bytes = std::max<size_t>(tile_max_bytes, (size_t)floor_words * (size_t)samp.n_tiles * 4);
    if (rstride) tile_max_bytes = std::max<size_t>(tile_max_bytes, (size_t)regs_samp.tq * regs_samp.n_tiles * 4);
    if (tile_max_bytes && (rc = sc.need(kTileMax, tile_max_bytes)) != MTR_OK) return rc;
    if ((rc = sc.ready()) != MTR_OK) return rc;
    // A sampling launch's floor kernel sets every real query's floor; otherwise they start from zero.
    if (stride < 2 && !rstride) HIP_TRY(mtr::launch_fill_u32(sc.p[kMfmaFloor], 0u, (size_t)floor_words, stream));
    // Query rows past b must be zero.
    char* qbuf = static_cast<char*>(sc.p[kQTiles]);
    if (sc.keeps_query_tiles()) {
        if (sc.p[kQTiles] != q_old) ix->q_init_rows = ix->q_data_rows = 0;
        if (q_rows > ix->q_init_rows) {
            HIP_TRY(mtr::launch_fill_u32(qbuf + (size_t)ix->q_init_rows * q_row_bytes, 0u,
                                         (size_t)(q_rows - ix->q_init_rows) * q_row_bytes / 4, stream));
            ix->q_init_rows = q_rows;
        }
        if (ix->q_data_rows > b)
            HIP_TRY(mtr::launch_fill_u32(qbuf + (size_t)b * q_row_bytes, 0u, (size_t)(ix->q_data_rows - b) * q_row_bytes / 4, stream));
        ix->q_data_rows = b;
    } else if (q_rows > b) {
        HIP_TRY(mtr::launch_fill_u32(qbuf + (size_t)b * q_row_bytes, 0u, (size_t)(q_rows - b) * q_row_bytes / 4, stream));
    }
    HIP_TRY(mtr::launch_load_rows(dev_queries, q_dtype, b, ix->d, qbuf, ix->dtype, ix->ld,
                                  metric == MTR_COSINE ? 1 : 0, nf, stream));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const bool profiling = ix->profiling && sc.may_profile();
    if (profiling) {
        if ((rc = next_event_pair(ix, &e0, &e1)) != MTR_OK) return rc;
        HIP_TRY(hipEventRecord(e0, stream));
    }
    unsigned int* floors = static_cast<unsigned int*>(sc.p[kMfmaFloor]);
    float* lists_s = static_cast<float*>(sc.p[kCandS]);
    int* lists_i = static_cast<int*>(sc.p[kCandI]);
    if (stride >= 2) {
        unsigned int* tile_max = static_cast<unsigned int*>(sc.p[kTileMax]);
        const mtr::MfmaPass sampling{stride, tile_max, nullptr, floor_words};
        HIP_TRY(mtr::launch_score_mfma(ix->rows, ix->dtype, ix->n, ix->ld, qbuf, b, k, samp, sampling,
                                       sc.p[kMfmaCand], floors, lists_s, lists_i, stream));
        HIP_TRY(mtr::launch_floor_from_tile_max(tile_max, (int)samp.n_tiles, b, k, floors, stream));
    }
    if (rstride) {
        unsigned int* tile_max = static_cast<unsigned int*>(sc.p[kTileMax]);
        // the sampling form raises tile_max with atomic maxima: every word starts at zero (below any score)
        HIP_TRY(mtr::launch_fill_u32(tile_max, 0u, (size_t)regs_samp.tq * regs_samp.n_tiles, stream));
        HIP_TRY(mtr::launch_score_wide(ix->rows, ix->n, ix->ld, qbuf, b, regs_samp, nullptr, nullptr, nullptr, rstride, tile_max, stream));
        HIP_TRY(mtr::launch_floor_from_tile_max(tile_max, (int)regs_samp.n_tiles, b, k, floors, stream));
    }
    int* overflow = nullptr;
    if (use_wide) {
        unsigned int* counts = static_cast<unsigned int*>(sc.p[kWideCounts]);
        unsigned long long* keys = static_cast<unsigned long long*>(sc.p[kWideKeys]);
        unsigned int* tile_max = static_cast<unsigned int*>(sc.p[kTileMax]);
        overflow = reinterpret_cast<int*>(counts + wide.count_words);
        HIP_TRY(mtr::launch_fill_u32(overflow, 0u, 1, stream));
        // the sampling form raises tile_max with atomic maxima: every word starts at zero (below any score)
        HIP_TRY(mtr::launch_fill_u32(tile_max, 0u, (size_t)wide.n_qtiles * wide.tq * wide_samp.n_tiles, stream));
        HIP_TRY(mtr::launch_score_wide(ix->rows, ix->n, ix->ld, qbuf, b, wide_samp, nullptr, nullptr, nullptr, wstride,
                                       tile_max, stream));
        HIP_TRY(mtr::launch_floor_from_tile_max(tile_max, (int)wide_samp.n_tiles, b, k, floors, stream));
        HIP_TRY(mtr::launch_score_wide(ix->rows, ix->n, ix->ld, qbuf, b, wide, floors, keys, counts, 1, nullptr, stream));
        if (profiling) HIP_TRY(hipEventRecord(e1, stream));
        HIP_TRY(mtr::launch_select_keys(keys, counts, wide.n_lists * mtr::kWideSubs, mtr::kWideSubs, wide.tq, wide.cap, (int)ix->n, b, k, ix->row_offset, dev_out_idx,
                                        dev_out_score, overflow, stream));
        HIP_TRY(mtr::launch_reset_floors_if(floors, floor_words, overflow, stream));
    }
    ix->last_kernel = use_wide ? "score_filter_wide_kernel" : "score_topk_mfma_kernel";
    mtr::MfmaPass all{1, nullptr, overflow, floor_words};
    all.sync = scan_sync(ix);
    HIP_TRY(mtr::launch_score_mfma(ix->rows, ix->dtype, ix->n, ix->ld, qbuf, b, k, plan, all, sc.p[kMfmaCand],
                                   floors, lists_s, lists_i, stream));
    if (profiling && !use_wide) HIP_TRY(hipEventRecord(e1, stream));
    HIP_TRY(mtr::launch_merge_topk(lists_s, lists_i, nullptr, plan.n_ranges, (long long)k, (long long)k,
                                   (long long)plan.n_ranges * k, k, b, k, ix->row_offset, dev_out_idx, dev_out_score,
                                   stream, overflow));
    return MTR_OK;
}

// The next value of the handle's non-finite word (see mtr::NonfiniteFlag).  Once
// in two billion calls the word starts over: everything enqueued on the handle
// is waited for first (never reached by the allocation-free entry point, which
// does not use epochs).
int next_epoch(mtr_index* ix, hipStream_t stream, mtr::NonfiniteFlag* nf) {
    if (ix->epoch == 0x7fffffff) {
        if (ix->used) HIP_TRY(hipEventSynchronize(ix->last_use));
        HIP_TRY(hipStreamSynchronize(stream));
        HIP_TRY(hipMemset(ix->flag, 0, 4));
        ix->epoch = 0;
    }
    *nf = mtr::NonfiniteFlag{ix->flag, ++ix->epoch};
    return MTR_OK;
}

// Scoring + merge of one batch, enqueued on `stream`.  dev_queries: b x d in
// q_dtype on the shard's device.
template <typename S>
int enqueue_search(mtr_index* ix, S& sc, const void* dev_queries, int q_dtype, int b, int k, int metric,
                   mtr::NonfiniteFlag nf, long long* dev_out_idx, float* dev_out_score, hipStream_t stream) {
    if (metric == MTR_COSINE && !ix->normalized)
        return fail(MTR_ERR_INVALID, "cosine search needs a shard built with normalize=1");
    if (b >= mfma_min_batch(ix->dtype)) {
        mtr::MfmaPlan mplan;
        if (mtr::plan_mfma(ix->dtype, ix->n, ix->ld, b, k, ix->num_cus, 0, &mplan) == 0)
            return enqueue_mfma(ix, sc, mplan, dev_queries, q_dtype, b, k, metric, nf, dev_out_idx, dev_out_score, stream);
    }
    mtr::ScanPlan plan;   // with candidate lists of k entries
    if (mtr::plan_scan(ix->dtype, ix->n, ix->d, ix->ld, b, k, ix->num_cus, &plan) != 0)
        return fail(MTR_ERR_INVALID, "unsupported search shape (d=%d, k=%d)", ix->d, k);
    // Long result lists: the rows are dealt evenly to the workgroups, so a workgroup's
    // share of a query's k best rows is about k / grid, and keeping k candidates per
    // workgroup (every wave inserting into a list of k) is what makes the scan slow for
    // large k (10M x 1024 bf16: k = 1024 20 ms against 2.85 ms at k = 10).  The scan
    // keeps lists of list_k < k instead (eight times the even share, at least 16), a
    // check kernel proves afterwards that no list dropped a row of the result, and only
    // if it cannot (rows of one neighbourhood that all match: the flag) the search runs
    // again with full lists, enqueued behind it and a no-op otherwise.
    mtr::ScanPlan splan = plan;
    int list_k = k;
    {
        const int forced = mtr::tuning().scan_short_lists;
        if (forced != 0 && k > 16) {
            int c = forced;
            if (forced < 0) {
                const int share = 8 * ((k + plan.grid - 1) / plan.grid);
                c = 16;
                while (c < share) c *= 2;
            }
            mtr::ScanPlan t;
            if (c >= 1 && c < k && mtr::plan_scan(ix->dtype, ix->n, ix->d, ix->ld, b, c, ix->num_cus, &t) == 0 &&
                (forced > 0 || ix->n / t.grid >= 16ll * c)) {
                splan = t;
                list_k = c;
            }
        }
    }
    const bool short_lists = list_k < k;
    int rc;
    const int max_qb = std::max(plan.qb, splan.qb);
    if ((rc = sc.need(kQF32, (size_t)max_qb * plan.qstride * 4)) != MTR_OK) return rc;
    const size_t cand_cells = std::max((size_t)splan.qb * splan.grid * list_k, (size_t)plan.qb * plan.grid * k);
    if ((rc = sc.need(kCandS, cand_cells * 4)) != MTR_OK) return rc;
    if ((rc = sc.need(kCandI, cand_cells * 4)) != MTR_OK) return rc;
    if ((rc = sc.ready()) != MTR_OK) return rc;
    ix->last_kernel = splan.stream_sweeps ? "scan_stream_kernel" : "scan_topk_kernel";
    float* q_f32 = static_cast<float*>(sc.p[kQF32]);
    float* cand_s = static_cast<float*>(sc.p[kCandS]);
    int* cand_i = static_cast<int*>(sc.p[kCandI]);
    int* redo = reinterpret_cast<int*>(ix->flag) + 3;   // raised by the check kernel: the lists were too short
    const int qsize = dtype_size(q_dtype);
    const bool profiling = ix->profiling && sc.may_profile();
    const int normalize = metric == MTR_COSINE ? 1 : 0;
    for (int b0 = 0; b0 < b; b0 += splan.qb) {
        const int nb = std::min(splan.qb, b - b0);
        // fp32 queries are normalised and padded inside the scan kernel; other
        // types go through the loader first (it writes all qstride columns of
        // each query it converts; unused slots of a short pass are zeroed).
        const char* qsrc = static_cast<const char*>(dev_queries) + (size_t)b0 * ix->d * qsize;
        const bool raw = q_dtype == MTR_F32;
        if (!raw) {
            if (nb < splan.qb) HIP_TRY(mtr::launch_fill_u32(q_f32, 0u, (size_t)splan.qb * splan.qstride, stream));
            HIP_TRY(mtr::launch_load_rows(qsrc, q_dtype, nb, ix->d, q_f32, MTR_F32, splan.qstride, normalize, nf, stream));
        }
        if (short_lists) HIP_TRY(mtr::launch_fill_u32(redo, 0u, 1, stream));
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (profiling) {
            if ((rc = next_event_pair(ix, &e0, &e1)) != MTR_OK) return rc;
            HIP_TRY(hipEventRecord(e0, stream));
        }
        HIP_TRY(mtr::launch_scan_topk(ix->rows, ix->dtype, ix->n, ix->ld, ix->d, q_f32,
                                      raw ? reinterpret_cast<const float*>(qsrc) : nullptr, nb, normalize, nf, list_k, splan,
                                      cand_s, cand_i, scan_sync(ix), stream));
        if (profiling) HIP_TRY(hipEventRecord(e1, stream));
        // the scan's per-workgroup lists are sorted best first
        long long* out_i = dev_out_idx + (size_t)b0 * k;
        float* out_s = dev_out_score + (size_t)b0 * k;
        HIP_TRY(mtr::launch_merge_sorted_lists(cand_s, cand_i, splan.grid, (long long)list_k, (long long)splan.grid * list_k,
                                               list_k, nb, k, ix->row_offset, out_i, out_s, stream));
        if (!short_lists) continue;
        HIP_TRY(mtr::launch_check_short_lists(cand_s, cand_i, splan.grid, (long long)list_k, (long long)splan.grid * list_k,
                                              list_k, nb, k, ix->row_offset, out_i, out_s, redo, stream));
        // The same queries with full lists, in passes of the full plan's width (it divides
        // this pass's: both are powers of two and longer lists never allow more queries).
        for (int s0 = 0; s0 < nb; s0 += plan.qb) {
            const int nbb = std::min(plan.qb, nb - s0);
            HIP_TRY(mtr::launch_scan_topk(ix->rows, ix->dtype, ix->n, ix->ld, ix->d, q_f32 + (size_t)s0 * plan.qstride,
                                          raw ? reinterpret_cast<const float*>(qsrc) + (size_t)s0 * ix->d : nullptr, nbb,
                                          normalize, nf, k, plan, cand_s, cand_i, scan_sync(ix), stream, redo));
            HIP_TRY(mtr::launch_merge_topk(cand_s, cand_i, nullptr, plan.grid, (long long)k, (long long)k,
                                           (long long)plan.grid * k, k, nbb, k, ix->row_offset, out_i + (size_t)s0 * k,
                                           out_s + (size_t)s0 * k, stream, redo));
        }
    }
    return MTR_OK;
}

int check_search_args(const mtr_index* ix, const void* queries, const void* out_idx, const void* out_score, int q_dtype,
                      int b, int k, int metric) {
    int rc = check_index(ix);
    if (rc != MTR_OK) return rc;
    if (!queries || !out_idx || !out_score) return fail(MTR_ERR_INVALID, "NULL argument");
    if (b < 1) return fail(MTR_ERR_INVALID, "b must be >= 1");
    if (k < 1 || k > MTR_MAX_K) return fail(MTR_ERR_INVALID, "k must be in [1, %d]", MTR_MAX_K);
    if (metric != MTR_COSINE && metric != MTR_IP) return fail(MTR_ERR_INVALID, "bad metric %d", metric);
    if (!dtype_size(q_dtype)) return fail(MTR_ERR_INVALID, "bad q_dtype %d", q_dtype);
    return MTR_OK;
}

}  // namespace

extern "C" {

const char* mtr_last_error(void) { return g_err.c_str(); }

const char* mtr_version(void) { return "medtutor-retrieval-mi355x 0.1 (gfx950)"; }

int mtr_index_alloc(int64_t n, int32_t d, int32_t store_dtype, int32_t normalize, int32_t device_id,
                    int64_t row_offset, mtr_index_t** out) {
    if (!out) return fail(MTR_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (n < 1 || d < 1) return fail(MTR_ERR_INVALID, "n and d must be positive (n=%lld, d=%d)", (long long)n, d);
    if (n > 0x7fffffffll) return fail(MTR_ERR_INVALID, "a shard holds at most 2^31-1 rows");
    if (store_dtype != MTR_F32 && store_dtype != MTR_BF16)
        return fail(MTR_ERR_INVALID, "store_dtype must be MTR_F32 or MTR_BF16");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device_id < 0 || device_id >= ndev) return fail(MTR_ERR_INVALID, "device %d of %d", device_id, ndev);
    DeviceGuard guard(device_id);
    if (!guard.ok) return fail(MTR_ERR_HIP, "hipSetDevice(%d) failed", device_id);
    mtr_index* ix = new (std::nothrow) mtr_index();
    if (!ix) return fail(MTR_ERR_OOM, "host allocation failed");
    ix->device = device_id;
    ix->n = n;
    ix->d = d;
    ix->dtype = store_dtype;
    ix->normalized = normalize ? 1 : 0;
    ix->row_offset = row_offset;
    const int epc = 16 / dtype_size(store_dtype);
    ix->ld = (d + epc - 1) / epc * epc;
    // Rows of a multiple of 128 bytes can go to the matrix-core kernels (whole K steps); other
    // widths are padded up to one when that adds at most an eighth to the row (300 fp32 values:
    // 1200 -> 1280 bytes), else they stay with the scan, which reads only the data columns.
    {
        const int per_step = 128 / dtype_size(store_dtype);
        const int ld_step = (d + per_step - 1) / per_step * per_step;
        if (d >= per_step && (ld_step - d) * 8 <= d) ix->ld = ld_step;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) ix->num_cus = prop.multiProcessorCount;
    hipError_t e = hipMalloc(&ix->rows, (size_t)n * ix->ld * dtype_size(store_dtype));
    if (e != hipSuccess) {
        delete ix;
        return fail(e == hipErrorOutOfMemory ? MTR_ERR_OOM : MTR_ERR_HIP, "hipMalloc of %lld x %d rows: %s",
                    (long long)n, d, hipGetErrorString(e));
    }
    e = hipMalloc(reinterpret_cast<void**>(&ix->flag), 256);
    if (e == hipSuccess) e = hipMemset(ix->flag, 0, 256);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ix->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ix->last_use, hipEventDisableTiming);
    if (e != hipSuccess) {
        mtr_index_destroy(ix);
        return fail(MTR_ERR_HIP, "shard set-up: %s", hipGetErrorString(e));
    }
    // The shard's GPU is current (guard): its address of the static scratch array.
    // Only the allocation-free entry point needs it, so a failed lookup does not
    // fail the shard: that entry point then reports the array as unresolved.
    if (static_scratch_for(device_id, &ix->static_scratch) != MTR_OK) ix->static_scratch = nullptr;
    *out = ix;
    return MTR_OK;
}

namespace {

// Host rows -> shard through page-locked staging: the host copy of chunk i + 1
// into its pinned buffer runs while chunk i crosses PCIe and is converted.
int upload_host_rows(mtr_index* ix, int64_t row_start, const char* rows, int64_t n_rows, int src_dtype, int normalize,
                     mtr::NonfiniteFlag nf) {
    int rc;
    const size_t esize = dtype_size(ix->dtype);
    const size_t src_row = (size_t)ix->d * dtype_size(src_dtype);
    const int64_t chunk_rows = std::max<int64_t>(1, (int64_t)(kStageBytes / src_row));
    const size_t chunk_bytes = (size_t)std::min(chunk_rows, n_rows) * src_row;
    const int n_buf = n_rows > chunk_rows ? 2 : 1;
    if (ix->pinned_cap < chunk_bytes) {
        for (int i = 0; i < 2; ++i) {
            if (ix->pinned[i]) (void)hipHostFree(ix->pinned[i]);
            ix->pinned[i] = nullptr;
        }
        ix->pinned_cap = 0;
    }
    for (int i = 0; i < n_buf; ++i) {
        if ((rc = ix->stage[i].reserve(chunk_bytes)) != MTR_OK) return rc;
        if (!ix->pinned[i]) HIP_TRY(hipHostMalloc(&ix->pinned[i], std::max(chunk_bytes, ix->pinned_cap), hipHostMallocDefault));
        if (!ix->stage_done[i]) HIP_TRY(hipEventCreateWithFlags(&ix->stage_done[i], hipEventDisableTiming));
    }
    ix->pinned_cap = std::max(chunk_bytes, ix->pinned_cap);
    int64_t chunk = 0;
    for (int64_t r0 = 0; r0 < n_rows; r0 += chunk_rows, ++chunk) {
        const int i = (int)(chunk & 1);
        const int64_t nr = std::min(chunk_rows, n_rows - r0);
        if (chunk >= 2) HIP_TRY(hipEventSynchronize(ix->stage_done[i]));   // the buffers' previous chunk is in the shard
        memcpy(ix->pinned[i], rows + (size_t)r0 * src_row, (size_t)nr * src_row);
        HIP_TRY(hipMemcpyAsync(ix->stage[i].p, ix->pinned[i], (size_t)nr * src_row, hipMemcpyHostToDevice, ix->stream));
        char* dst = static_cast<char*>(ix->rows) + (size_t)(row_start + r0) * ix->ld * esize;
        HIP_TRY(mtr::launch_load_rows(ix->stage[i].p, src_dtype, nr, ix->d, dst, ix->dtype, ix->ld, normalize, nf,
                                      ix->stream));
        HIP_TRY(hipEventRecord(ix->stage_done[i], ix->stream));
    }
    return MTR_OK;
}

}  // namespace

int mtr_index_load_rows(mtr_index_t* ix, int64_t row_start, const void* rows, int64_t n_rows, int32_t src_dtype,
                        int32_t rows_on_device) {
    int rc = check_index(ix);
    if (rc != MTR_OK) return rc;
    if (!rows && n_rows > 0) return fail(MTR_ERR_INVALID, "rows is NULL");
    const int normalize = (src_dtype & MTR_ROWS_AS_STORED) ? 0 : ix->normalized;
    src_dtype &= ~MTR_ROWS_AS_STORED;
    const int ssize = dtype_size(src_dtype);
    if (!ssize) return fail(MTR_ERR_INVALID, "bad src_dtype %d", src_dtype);
    if (row_start < 0 || n_rows < 0 || row_start + n_rows > ix->n)
        return fail(MTR_ERR_INVALID, "rows [%lld, %lld) outside the shard of %lld rows", (long long)row_start,
                    (long long)(row_start + n_rows), (long long)ix->n);
    std::lock_guard<std::mutex> lock(ix->mu);
    DeviceGuard guard(ix->device);
    if ((rc = begin_use(ix, ix->stream)) != MTR_OK) return rc;
    mtr::NonfiniteFlag nf;
    if ((rc = next_epoch(ix, ix->stream, &nf)) != MTR_OK) return rc;
    if (n_rows == 0) {
        // nothing to convert
    } else if (rows_on_device) {
        char* dst = static_cast<char*>(ix->rows) + (size_t)row_start * ix->ld * dtype_size(ix->dtype);
        HIP_TRY(mtr::launch_load_rows(rows, src_dtype, n_rows, ix->d, dst, ix->dtype, ix->ld, normalize, nf, ix->stream));
    } else if ((rc = upload_host_rows(ix, row_start, static_cast<const char*>(rows), n_rows, src_dtype, normalize, nf)) != MTR_OK) {
        (void)hipStreamSynchronize(ix->stream);
        return rc;
    }
    int flag = 0;
    HIP_TRY(hipMemcpyAsync(&flag, ix->flag, 4, hipMemcpyDeviceToHost, ix->stream));
    HIP_TRY(hipStreamSynchronize(ix->stream));
    ix->used = false;   // everything enqueued on the handle has finished: the next call has nothing to wait for
    if (flag == nf.value) return fail(MTR_ERR_NONFINITE, "Input contains NaN or infinity.");
    return MTR_OK;
}

int mtr_index_create(const void* host_rows, int64_t n, int32_t d, int32_t src_dtype, int32_t store_dtype,
                     int32_t normalize, int32_t device_id, int64_t row_offset, mtr_index_t** out) {
    int rc = mtr_index_alloc(n, d, store_dtype, normalize, device_id, row_offset, out);
    if (rc != MTR_OK) return rc;
    rc = mtr_index_load_rows(*out, 0, host_rows, n, src_dtype, 0);
    if (rc != MTR_OK) {
        std::string keep = g_err;
        mtr_index_destroy(*out);
        *out = nullptr;
        g_err = keep;
    }
    return rc;
}

int mtr_index_search(mtr_index_t* ix, const void* queries, int32_t q_dtype, int32_t queries_on_device, int32_t b,
                     int32_t k, int32_t metric, int64_t* out_idx, float* out_score) {
    int rc = check_search_args(ix, queries, out_idx, out_score, q_dtype, b, k, metric);
    if (rc != MTR_OK) return rc;
    std::lock_guard<std::mutex> lock(ix->mu);
    DeviceGuard guard(ix->device);
    if ((rc = begin_use(ix, ix->stream)) != MTR_OK) return rc;
    const void* dq = queries;
    if (!queries_on_device) {
        const size_t bytes = (size_t)b * ix->d * dtype_size(q_dtype);
        if ((rc = ix->q_in.reserve(bytes)) != MTR_OK) return rc;
        HIP_TRY(hipMemcpyAsync(ix->q_in.p, queries, bytes, hipMemcpyHostToDevice, ix->stream));
        dq = ix->q_in.p;
    }
    if ((rc = ix->out_i.reserve((size_t)b * k * 8)) != MTR_OK) return rc;
    if ((rc = ix->out_s.reserve((size_t)b * k * 4)) != MTR_OK) return rc;
    Scratch sc;
    sc.ix = ix;
    mtr::NonfiniteFlag nf;
    if ((rc = next_epoch(ix, ix->stream, &nf)) != MTR_OK) return rc;
    const int epoch = nf.value;
    rc = enqueue_search(ix, sc, dq, q_dtype, b, k, metric, nf, static_cast<long long*>(ix->out_i.p),
                        static_cast<float*>(ix->out_s.p), ix->stream);
    if (rc != MTR_OK) {
        std::string keep = g_err;
        (void)hipStreamSynchronize(ix->stream);
        ix->used = false;
        g_err = keep;
        return rc;
    }
    int flag = 0;
    HIP_TRY(hipMemcpyAsync(out_idx, ix->out_i.p, (size_t)b * k * 8, hipMemcpyDeviceToHost, ix->stream));
    HIP_TRY(hipMemcpyAsync(out_score, ix->out_s.p, (size_t)b * k * 4, hipMemcpyDeviceToHost, ix->stream));
    HIP_TRY(hipMemcpyAsync(&flag, ix->flag, 4, hipMemcpyDeviceToHost, ix->stream));
    // The call waits for its own work, so it leaves nothing for a later call on
    // another stream to wait for: no event needs recording.
    HIP_TRY(hipStreamSynchronize(ix->stream));
    ix->used = false;
    if (flag == epoch) return fail(MTR_ERR_NONFINITE, "Input contains NaN or infinity.");
    return MTR_OK;
}

int mtr_index_search_device(mtr_index_t* ix, const void* dev_queries, int32_t q_dtype, int32_t b, int32_t k,
                            int32_t metric, int64_t* dev_out_idx, float* dev_out_score, void* stream) {
    int rc = check_search_args(ix, dev_queries, dev_out_idx, dev_out_score, q_dtype, b, k, metric);
    if (rc != MTR_OK) return rc;
    std::lock_guard<std::mutex> lock(ix->mu);
    DeviceGuard guard(ix->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if ((rc = begin_use(ix, s)) != MTR_OK) return rc;
    Scratch sc;
    sc.ix = ix;
    mtr::NonfiniteFlag nf;
    if ((rc = next_epoch(ix, s, &nf)) != MTR_OK) return rc;
    ix->device_epoch = nf.value;
    ix->last_device_noalloc = false;
    rc = enqueue_search(ix, sc, dev_queries, q_dtype, b, k, metric, nf, reinterpret_cast<long long*>(dev_out_idx),
                        dev_out_score, s);
    std::string keep = g_err;
    const int rc2 = end_use(ix, s);
    if (rc != MTR_OK) g_err = keep;
    return rc != MTR_OK ? rc : rc2;
}

int mtr_index_search_device_noalloc(mtr_index_t* ix, const void* dev_queries, int32_t q_dtype, int32_t b, int32_t k,
                                    int32_t metric, int64_t* dev_out_idx, float* dev_out_score, void* stream) {
    int rc = check_search_args(ix, dev_queries, dev_out_idx, dev_out_score, q_dtype, b, k, metric);
    if (rc != MTR_OK) return rc;
    std::lock_guard<std::mutex> lock(ix->mu);
    DeviceGuard guard(ix->device);
    if (!ix->static_scratch) return fail(MTR_ERR_HIP, "the static scratch array was not resolved on device %d", ix->device);
    // No events and no cross-stream ordering here: this entry point may run under
    // stream capture, and its callers keep to one stream per GPU.
    StaticScratch sc;
    sc.ix = ix;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // The non-finite report of this entry point: word 1 of the handle's flag
    // block, cleared here (a kernel, so a captured call clears it on every
    // replay) and raised to 1 by the kernels; no epoch, which capture would freeze.
    HIP_TRY(mtr::launch_fill_u32(ix->flag + 1, 0u, 1, s));
    ix->last_device_noalloc = true;
    return enqueue_search(ix, sc, dev_queries, q_dtype, b, k, metric, mtr::NonfiniteFlag{ix->flag + 1, 1},
                          reinterpret_cast<long long*>(dev_out_idx), dev_out_score, s);
}

int mtr_index_nonfinite(mtr_index_t* ix, int32_t* nonfinite) {
    int rc = check_index(ix);
    if (rc != MTR_OK) return rc;
    if (!nonfinite) return fail(MTR_ERR_INVALID, "nonfinite is NULL");
    std::lock_guard<std::mutex> lock(ix->mu);
    DeviceGuard guard(ix->device);
    if (ix->used) HIP_TRY(hipEventSynchronize(ix->last_use));
    int flag[2] = {0, 0};
    HIP_TRY(hipMemcpy(flag, ix->flag, 8, hipMemcpyDeviceToHost));
    if (ix->last_device_noalloc) *nonfinite = flag[1] != 0 ? 1 : 0;
    else *nonfinite = (ix->device_epoch > 0 && flag[0] == ix->device_epoch) ? 1 : 0;
    return MTR_OK;
}

int mtr_merge_topk_device(const int64_t* dev_idx, const float* dev_score, int32_t n_lists, int64_t idx_list_stride,
                          int64_t score_list_stride, int32_t b, int32_t k_in, int32_t k_out, int64_t* dev_out_idx,
                          float* dev_out_score, int32_t device_id, void* stream) {
    if (!dev_idx || !dev_score || !dev_out_idx || !dev_out_score) return fail(MTR_ERR_INVALID, "NULL argument");
    if (n_lists < 1 || b < 1 || k_in < 1 || k_out < 1 || k_out > MTR_MAX_K)
        return fail(MTR_ERR_INVALID, "bad merge shape (lists=%d, b=%d, k_in=%d, k_out=%d)", n_lists, b, k_in, k_out);
    DeviceGuard guard(device_id);
    if (!guard.ok) return fail(MTR_ERR_HIP, "hipSetDevice(%d) failed", device_id);
    const long long dense = (long long)b * k_in;
    if ((idx_list_stride != 0 && idx_list_stride < dense) || (score_list_stride != 0 && score_list_stride < dense))
        return fail(MTR_ERR_INVALID, "list strides must be 0 (dense) or at least b*k_in");
    HIP_TRY(mtr::launch_merge_topk(dev_score, nullptr, reinterpret_cast<const long long*>(dev_idx), n_lists,
                                   idx_list_stride ? idx_list_stride : dense,
                                   score_list_stride ? score_list_stride : dense, (long long)k_in, k_in, b, k_out, 0,
                                   reinterpret_cast<long long*>(dev_out_idx), dev_out_score,
                                   static_cast<hipStream_t>(stream)));
    return MTR_OK;
}

int mtr_pool_normalize(const void* dev_hidden, int32_t dtype, const int64_t* dev_mask, int32_t b, int32_t t,
                       int32_t d, float* dev_out, int32_t device_id, void* stream) {
    if (!dev_hidden || !dev_mask || !dev_out) return fail(MTR_ERR_INVALID, "NULL argument");
    if (dtype != MTR_F32 && dtype != MTR_BF16 && dtype != MTR_F16) return fail(MTR_ERR_INVALID, "bad dtype %d", dtype);
    if (b < 1 || t < 1 || d < 1) return fail(MTR_ERR_INVALID, "bad shape (b=%d, t=%d, d=%d)", b, t, d);
    DeviceGuard guard(device_id);
    if (!guard.ok) return fail(MTR_ERR_HIP, "hipSetDevice(%d) failed", device_id);
    HIP_TRY(mtr::launch_pool_normalize(dev_hidden, dtype, reinterpret_cast<const long long*>(dev_mask), b, t, d, dev_out,
                                       static_cast<hipStream_t>(stream)));
    return MTR_OK;
}

int mtr_index_info(const mtr_index_t* ix, int64_t* n, int32_t* d, int32_t* store_dtype, int32_t* normalized,
                   int32_t* device_id, int64_t* row_offset) {
    int rc = check_index(ix);
    if (rc != MTR_OK) return rc;
    if (n) *n = ix->n;
    if (d) *d = ix->d;
    if (store_dtype) *store_dtype = ix->dtype;
    if (normalized) *normalized = ix->normalized;
    if (device_id) *device_id = ix->device;
    if (row_offset) *row_offset = ix->row_offset;
    return MTR_OK;
}

int mtr_index_read_rows(mtr_index_t* ix, int64_t row_start, int64_t n_rows, float* host_out) {
    int rc = check_index(ix);
    if (rc != MTR_OK) return rc;
    if (!host_out || row_start < 0 || n_rows < 0 || row_start + n_rows > ix->n)
        return fail(MTR_ERR_INVALID, "bad row range");
    if (n_rows == 0) return MTR_OK;
    std::lock_guard<std::mutex> lock(ix->mu);
    DeviceGuard guard(ix->device);
    const size_t bytes = (size_t)n_rows * ix->d * 4;
    if ((rc = ix->stage[0].reserve(bytes)) != MTR_OK) return rc;
    if ((rc = begin_use(ix, ix->stream)) != MTR_OK) return rc;
    const char* src = static_cast<const char*>(ix->rows) + (size_t)row_start * ix->ld * dtype_size(ix->dtype);
    HIP_TRY(mtr::launch_read_rows(src, ix->dtype, n_rows, ix->d, ix->ld, static_cast<float*>(ix->stage[0].p), ix->stream));
    HIP_TRY(hipMemcpyAsync(host_out, ix->stage[0].p, bytes, hipMemcpyDeviceToHost, ix->stream));
    HIP_TRY(hipStreamSynchronize(ix->stream));
    ix->used = false;
    return MTR_OK;
}

int mtr_index_profile(mtr_index_t* ix, int32_t enable) {
    int rc = check_index(ix);
    if (rc != MTR_OK) return rc;
    std::lock_guard<std::mutex> lock(ix->mu);
    ix->profiling = enable != 0;
    ix->events_used = 0;
    return MTR_OK;
}

int mtr_index_scan_time(mtr_index_t* ix, int32_t* n_launches, double* total_ms) {
    int rc = check_index(ix);
    if (rc != MTR_OK) return rc;
    std::lock_guard<std::mutex> lock(ix->mu);
    DeviceGuard guard(ix->device);
    double sum = 0.0;
    for (size_t i = 0; i < ix->events_used; ++i) {
        HIP_TRY(hipEventSynchronize(ix->events[i].second));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, ix->events[i].first, ix->events[i].second));
        sum += ms;
    }
    if (n_launches) *n_launches = (int32_t)ix->events_used;
    if (total_ms) *total_ms = sum;
    return MTR_OK;
}

int mtr_index_scan_samples(mtr_index_t* ix, float* out_ms, int32_t cap, int32_t* n_launches) {
    int rc = check_index(ix);
    if (rc != MTR_OK) return rc;
    if (cap < 0 || (cap > 0 && !out_ms)) return fail(MTR_ERR_INVALID, "out_ms is NULL or cap is negative");
    std::lock_guard<std::mutex> lock(ix->mu);
    DeviceGuard guard(ix->device);
    for (size_t i = 0; i < ix->events_used && i < (size_t)cap; ++i) {
        HIP_TRY(hipEventSynchronize(ix->events[i].second));
        HIP_TRY(hipEventElapsedTime(&out_ms[i], ix->events[i].first, ix->events[i].second));
    }
    if (n_launches) *n_launches = (int32_t)ix->events_used;
    return MTR_OK;
}

int mtr_index_read_ceiling(mtr_index_t* ix, int32_t reps, float* avg_ms) {
    int rc = check_index(ix);
    if (rc != MTR_OK) return rc;
    if (reps < 1 || !avg_ms) return fail(MTR_ERR_INVALID, "reps must be >= 1 and avg_ms not NULL");
    std::lock_guard<std::mutex> lock(ix->mu);
    DeviceGuard guard(ix->device);
    mtr::ScanPlan plan;
    if (mtr::plan_scan(ix->dtype, ix->n, ix->d, ix->ld, 1, 1, ix->num_cus, &plan) != 0)
        return fail(MTR_ERR_INVALID, "no scan shape for this shard");
    if ((rc = begin_use(ix, ix->stream)) != MTR_OK) return rc;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIP_TRY(hipEventCreate(&e0));
    hipError_t err = hipEventCreate(&e1);
    unsigned int* sink = reinterpret_cast<unsigned int*>(ix->flag) + 8;   // spare words of the flag block
    double sum = 0.0;
    for (int r = 0; r < reps + 1 && err == hipSuccess; ++r) {   // the first launch warms up
        float ms = 0.f;
        if ((err = hipEventRecord(e0, ix->stream)) != hipSuccess) break;
        if ((err = mtr::launch_scan_read_only(ix->rows, ix->dtype, ix->n, ix->ld, ix->d, plan, sink, scan_sync(ix), ix->stream)) != hipSuccess) break;
        if ((err = hipEventRecord(e1, ix->stream)) != hipSuccess) break;
        if ((err = hipEventSynchronize(e1)) != hipSuccess) break;
        if ((err = hipEventElapsedTime(&ms, e0, e1)) != hipSuccess) break;
        if (r > 0) sum += ms;
    }
    (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (err != hipSuccess) {
        (void)hipStreamSynchronize(ix->stream);
        ix->used = false;
        return fail(MTR_ERR_HIP, "timing the scan's loads: %s", hipGetErrorString(err));
    }
    ix->used = false;   // every launch above was waited for
    *avg_ms = (float)(sum / reps);
    return MTR_OK;
}

int mtr_index_last_kernel(const mtr_index_t* ix, char* buf, int32_t cap) {
    int rc = check_index(ix);
    if (rc != MTR_OK) return rc;
    if (!buf || cap < 1) return fail(MTR_ERR_INVALID, "buf is NULL or cap < 1");
    snprintf(buf, (size_t)cap, "%s", ix->last_kernel);
    return MTR_OK;
}

void mtr_index_destroy(mtr_index_t* ix) {
    if (!ix) return;
    {
        DeviceGuard guard(ix->device);
        if (ix->used) (void)hipEventSynchronize(ix->last_use);
        if (ix->stream) (void)hipStreamSynchronize(ix->stream);
        for (auto& ev : ix->events) {
            (void)hipEventDestroy(ev.first);
            (void)hipEventDestroy(ev.second);
        }
        for (auto& buf : ix->scratch) buf.release();
        ix->q_in.release();
        ix->out_i.release();
        ix->out_s.release();
        for (int i = 0; i < 2; ++i) {
            ix->stage[i].release();
            if (ix->pinned[i]) (void)hipHostFree(ix->pinned[i]);
            if (ix->stage_done[i]) (void)hipEventDestroy(ix->stage_done[i]);
        }
        if (ix->last_use) (void)hipEventDestroy(ix->last_use);
        if (ix->flag) (void)hipFree(ix->flag);
        if (ix->rows) (void)hipFree(ix->rows);
        if (ix->stream) (void)hipStreamDestroy(ix->stream);
    }
    delete ix;
}

// ---- launch-shape switches ---------------------------------------------------

namespace {
struct TuningField {
    const char* name;
    int mtr::Tuning::*field;
};
const TuningField kTuningFields[] = {
    {"mfma_min_batch", &mtr::Tuning::mfma_min_batch},     {"mfma_tnq", &mtr::Tuning::mfma_tnq},
    {"mfma_per_cu", &mtr::Tuning::mfma_per_cu},           {"mfma_cap", &mtr::Tuning::mfma_cap},
    {"mfma_sample_stride", &mtr::Tuning::mfma_sample_stride}, {"mfma_stream_rows", &mtr::Tuning::mfma_stream_rows},
    {"mfma_interleave", &mtr::Tuning::mfma_interleave},   {"mfma_dynamic_tail", &mtr::Tuning::mfma_dynamic_tail}, {"mfma_via_regs", &mtr::Tuning::mfma_via_regs},
    {"mfma_static_permille", &mtr::Tuning::mfma_static_permille},   {"mfma_stages", &mtr::Tuning::mfma_stages},           {"mfma_tile_rows", &mtr::Tuning::mfma_tile_rows},
    {"wide_min_batch", &mtr::Tuning::wide_min_batch},     {"wide_qgroup", &mtr::Tuning::wide_qgroup}, {"wide_tile_queries", &mtr::Tuning::wide_tile_queries}, {"wide_stream_rows", &mtr::Tuning::wide_stream_rows},
    {"scan_short_lists", &mtr::Tuning::scan_short_lists},
    {"scan_wg_per_cu", &mtr::Tuning::scan_wg_per_cu},     {"scan_form", &mtr::Tuning::scan_form},
    {"scan_static_permille", &mtr::Tuning::scan_static_permille}, {"scan_pair_permille", &mtr::Tuning::scan_pair_permille},
    {"scan_chunk_groups", &mtr::Tuning::scan_chunk_groups}, {"scan_late_refill", &mtr::Tuning::scan_late_refill},
    {"scan_mid_permille", &mtr::Tuning::scan_mid_permille},
    {"scan_nontemporal", &mtr::Tuning::scan_nontemporal}, {"scan_max_qb", &mtr::Tuning::scan_max_qb}, {"pool_sliced", &mtr::Tuning::pool_sliced},
#ifdef MTR_TIMING_SWITCHES
    {"timing_mfma", &mtr::Tuning::timing_mfma},           {"timing_wide", &mtr::Tuning::timing_wide},
#endif
};
}  // namespace

int mtr_tuning_set(const char* name, int32_t value) {
    if (!name) return fail(MTR_ERR_INVALID, "name is NULL");
    for (const TuningField& f : kTuningFields)
        if (strcmp(name, f.name) == 0) {
            mtr::tuning().*(f.field) = value;
            return MTR_OK;
        }
    return fail(MTR_ERR_INVALID, "no launch-shape switch named '%s'", name);
}

int mtr_tuning_get(const char* name, int32_t* value) {
    if (!name || !value) return fail(MTR_ERR_INVALID, "NULL argument");
    for (const TuningField& f : kTuningFields)
        if (strcmp(name, f.name) == 0) {
            *value = mtr::tuning().*(f.field);
            return MTR_OK;
        }
    return fail(MTR_ERR_INVALID, "no launch-shape switch named '%s'", name);
}

}  // extern "C"

namespace mtr {
// Initial values: one environment variable per switch (MTR_ + the upper-case
// name), read once.
Tuning& tuning() {
    static Tuning t = [] {
        Tuning v;
        for (const TuningField& f : kTuningFields) {
            std::string env = "MTR_";
            for (const char* c = f.name; *c; ++c) env.push_back((char)toupper((unsigned char)*c));
            if (const char* e = getenv(env.c_str())) v.*(f.field) = atoi(e);
        }
#ifdef MTR_TIMING_SWITCHES
        if (const char* e = getenv("MTR_TIMING_FLOOR")) v.timing_floor = (float)atof(e);
#endif
        return v;
    }();
    return t;
}
}  // namespace mtr
