// Shared device-side helpers and kernel launch prototypes for the retrieval library.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mtr {

typedef unsigned short bf16_t;  // raw bfloat16 bits
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
using f32x4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ float bf16_to_f32(bf16_t h) { return __uint_as_float(((unsigned int)h) << 16); }

// Round-to-nearest-even; callers guarantee a finite input (non-finite inputs
// are rejected before any conversion).
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    unsigned int u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}

// Elements per 16-byte chunk and chunk -> float unpacking for the two storage types.
template <typename T> struct Chunk;
template <> struct Chunk<float> {
    static constexpr int EPC = 4;
    static __device__ __forceinline__ void unpack(const u32x4& v, float (&f)[4]) {
        f[0] = __uint_as_float(v[0]); f[1] = __uint_as_float(v[1]);
        f[2] = __uint_as_float(v[2]); f[3] = __uint_as_float(v[3]);
    }
};
template <> struct Chunk<bf16_t> {
    static constexpr int EPC = 8;
    static __device__ __forceinline__ void unpack(const u32x4& v, float (&f)[8]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[2 * i] = __uint_as_float(v[i] << 16);
            f[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
        }
    }
};

// ---- launch-shape switches ---------------------------------------------------
// Process-wide overrides of the launch-shape rules, for parity tests (which
// force the large-shard kernel forms onto small shards) and timing sweeps.
// Initial values come from the environment (MTR_MFMA_MIN_BATCH, ... one
// variable per field, upper case) when the library is first used;
// mtr_tuning_set changes them afterwards.  No value changes a result beyond
// what the kernels' parity tests allow.
struct Tuning {
    int mfma_min_batch = 0;       // smallest batch on the matrix cores (0: 2 for bf16, 3 for fp32 shards)
    int mfma_tnq = 0;             // query-tile width 32 / 64 / 128 (0: by batch size)
    int mfma_per_cu = 0;          // workgroups per CU (0: measured best per form)
    int mfma_cap = 0;             // candidate slots per (workgroup, query) 256 / 512 (0: by k)
    int mfma_sample_stride = -1;  // -1: sampling launch on shards >= 1 GB with the sqrt rule; 0: never; n: every n-th tile, any shard
    int mfma_stream_rows = -1;    // -1: streaming row loads when one query tile reads a shard beyond the caches; 0 / 1
    int mfma_stages = 0;          // LDS stage buffers 2 / 3 (0: by form)
    int mfma_tile_rows = 0;       // rows per tile 128 / 256 (0: by form)
    int mfma_dynamic_tail = -1;   // -1: the tall ring forms claim their last tiles from a counter; 0 / 1
    int mfma_static_permille = 0; // ... share of the tiles taken by index in that form (0: 930)
    int mfma_interleave = -1;     // -1: ranges interleaved tile by tile when one query tile streams a large shard; 0 / 1
    int mfma_via_regs = -1;       // -1: the tall ring form's 128-query tile, and its 64-query tile where it does not claim its tail, take their rows through registers (on 16 x 16 x 32 MFMAs); 0 never; 1 wherever the shape allows, on 32 x 32 x 16 MFMAs; 2 wherever the shape allows, on 16 x 16 x 32
    int wide_min_batch = 129;     // smallest batch for the 256 x 256-tile kernel
    int wide_qgroup = 0;          // its query tiles per row list on one XCD (0: four)
    int wide_stream_rows = -1;    // -1: its row loads stream (nt) when one query tile reads a shard beyond the caches; 0 / 1
    int wide_tile_queries = 0;    // its query-tile width 128 / 192 / 256 (0: the narrowest that keeps the number of tiles)
    int scan_wg_per_cu = 0;       // scan workgroups per CU (0: one for a single query, occupancy otherwise)
    int scan_max_qb = 0;          // most queries per scan pass 1 / 2 / 4 / 8 (0: four)
    int scan_nontemporal = -1;    // -1: streaming row loads on shards beyond the caches; 0 / 1
    int scan_short_lists = -1;    // scan candidate lists shorter than k, verified afterwards: -1 by shape; 0 never; n: lists of n (k > n)
    int scan_form = 0;            // single-query scan: 0 by shard size; 1 classic form; 2 streaming form wherever the shape allows
    int scan_static_permille = 0; // streaming form: share of the row groups the waves take by index (0: 880) ...
    int scan_chunk_groups = 0;    // ... groups per claimed chunk of a first phase (0: one) ...
    int scan_mid_permille = 0;    // ... which runs up to here (0: no such phase); chunks of two groups follow ...
    int scan_pair_permille = 0;   // ... up to here (0: none either), single groups after
    int scan_late_refill = 0;     // from this share of the groups on, one group per wave, claimed on demand (0: 985; -1: never)
    int pool_sliced = -1;         // pooling kernel form: -1 by shape (two kernels from 256 tokens); 0 one kernel; 1 two kernels
    int timing_mfma = 0;          // MTR_TIMING_SWITCHES builds only: drop candidates / skip loads or MFMAs (wrong results)
    int timing_wide = 0;          // MTR_TIMING_SWITCHES builds only
    float timing_floor = 0.f;     // MTR_TIMING_SWITCHES builds only
};
Tuning& tuning();

// ---- host-callable launchers (defined in the .hip files) -------------------

// Where a kernel reports NaN/inf in its input: it raises *word to `value`
// (atomic max).  The caller hands out ever increasing values, one per call, and
// afterwards asks whether the word equals its value: the word never needs
// clearing between calls, which saves a launch per search.
struct NonfiniteFlag {
    int* word;
    int value;
};

// Row loader: finiteness check, optional L2 normalisation (float64 norm; norms
// below 10*DBL_EPSILON count as 1), cast to the storage type, zero padding of
// columns [d, ld).  src_dtype / dst_dtype use the MTR_F32/F64/BF16 codes.
hipError_t launch_load_rows(const void* src, int src_dtype, int64_t n_rows, int d, void* dst,
                            int dst_dtype, int ld, int normalize, NonfiniteFlag nonfinite_flag,
                            hipStream_t stream);

// dst[0 .. n_words) = value (32-bit words), as a kernel.  Used instead of
// hipMemsetAsync wherever a search clears scratch: memset nodes captured into a
// HIP graph replay with a garbage fill value on this runtime from the second
// replay on (measured: 128 bytes "zeroed" to +inf bit patterns), and searches
// are meant to be capturable.
hipError_t launch_fill_u32(void* dst, unsigned int value, size_t n_words, hipStream_t stream);

// Copies stored rows back as fp32 (n_rows x d).
hipError_t launch_read_rows(const void* rows, int dtype, int64_t n_rows, int d, int ld, float* dst,
                            hipStream_t stream);

struct ScanPlan {
    int qb;          // queries per pass (1, 2, 4 or 8)
    int nit;         // sweeps per row (16 lanes x 16 bytes each)
    int qstride;     // floats per padded query
    int grid;        // workgroups
    size_t lds;      // dynamic LDS bytes
    int nontemporal; // stream rows past the caches
    int stream_sweeps; // 0: classic form; else the streaming single-query form (four waves, sixteen-row groups,
                     // two register sets, claimed tail) with this many sweeps per unit; grid and lds are then its own
};

// Chooses the pass width and launch shape for the row scan.
int plan_scan(int dtype, int64_t n, int d, int ld, int b, int k, int num_cus, ScanPlan* plan);

// Row scan + fused per-workgroup top-k.  q: plan.qb padded fp32 queries
// ([qb][qstride]); outputs [qb][grid][k] candidate (score, local row) lists,
// each sorted, padded with (-inf, -1).
// run_if != NULL: the kernel does nothing while that device word is 0.
// sync: 32 device words, zero between launches, that the streaming form's waves claim rows
// through (a shard's launches must not overlap on them); unused by the classic form.
// Instead of prepared queries (q), the first nb raw fp32 queries ([nb][d],
// q_raw != NULL) can be given: every workgroup then normalises (if asked) and
// pads them itself and reports NaN/inf through nonfinite_flag.
hipError_t launch_scan_topk(const void* rows, int dtype, int64_t n, int ld, int d, const float* q,
                            const float* q_raw, int nb, int normalize, NonfiniteFlag nonfinite_flag, int k,
                            const ScanPlan& plan, float* cand_s, int* cand_i, unsigned int* sync, hipStream_t stream,
                            const int* run_if = nullptr);

// Candidate merge: for each of b queries, the best k_out of n_lists * k_in
// candidates.  Candidate (list l, query q, slot j) sits at
// l*list_stride + q*query_stride + j (list_stride_s for the scores).  idx32 != NULL: 32-bit local rows, to
// which idx_offset is added; else idx64.  Entries with id < 0 are padding.
// run_if != NULL: the kernel does nothing while that device word is 0.
hipError_t launch_merge_topk(const float* cand_s, const int* idx32, const long long* idx64,
                             int n_lists, long long list_stride, long long list_stride_s,
                             long long query_stride, int k_in,
                             int b, int k_out, long long idx_offset, long long* out_idx,
                             float* out_score, hipStream_t stream, const int* run_if = nullptr);

// Masked mean pooling over tokens + L2 normalisation of encoder outputs:
// hidden [b][t][d] (dtype 0 fp32, 2 bf16, 3 fp16), mask [b][t] int64 -> out [b][d] fp32.
hipError_t launch_pool_normalize(const void* hidden, int dtype, const long long* mask, int b, int t, int d, float* out,
                                 hipStream_t stream);

// The single-query scan's loads without its arithmetic (plan from plan_scan with b = 1): a
// live measurement of what the memory system delivers to the scan's access pattern.
hipError_t launch_scan_read_only(const void* rows, int dtype, int64_t n, int ld, int d, const ScanPlan& plan,
                                 unsigned int* sink, unsigned int* sync, hipStream_t stream);

struct MfmaPlan {
    int cap;                  // candidate slots per (workgroup, query): 256 or 512
    int slack;                // extra keys an in-loop re-selection may keep
    int tnq;                  // queries per tile: 32, 64 or 128 by batch size
    int tm;                   // rows per tile: 128, or 256 (64-query tiles in the ring form, eight waves)
    long long n_tiles;        // row tiles covered by the launch
    int n_qtiles;             // query tiles
    int n_ranges;             // row ranges (= candidate lists per query)
    long long tiles_per_range;
    int grid;
    size_t lds;
    size_t cand_bytes;        // scratch for the workgroups' candidate buffers
    int stream_rows;          // row tiles loaded with the streaming (nt) cache policy
    int stages;               // LDS stage buffers: 2, or 3 (ring with counted waits)
    int interleave;           // row ranges interleaved tile by tile instead of contiguous stretches
    int dynamic_tail;         // ... and the last tiles claimed from a device-wide counter (needs MfmaPass::sync)
    long long static_tiles;   // tiles every range takes by index in that form
    int via_regs;             // tall ring form with the rows in flight in registers instead of LDS stages: 1 on 32 x 32 x 16 MFMAs,
                              // 2 on 16 x 16 x 32 (a sampling launch must then be the filter kernel's: launch_score_wide with tile_max)
};

// Launch shape of the matrix-core scoring kernel; -1 when the shape is not
// covered (row stride not a multiple of 128 bytes or shorter than 256, or k > 256).
// sampling: the shape of a sampling launch (128-row tiles, two buffers, default cache policy).
int plan_mfma(int dtype, int64_t n, int ld, int b, int k, int num_cus, int sampling, MfmaPlan* plan);

// Which row tiles of the matrix one launch covers and what it keeps.  Large
// shards are searched in two launches: a sampling launch over every
// tile_stride-th tile that only records each query's best score per tile
// (tile_max), whose k-th largest per query (launch_floor_from_tile_max) is a
// valid starting floor for the full launch (tile_stride 1, tile_max NULL).
constexpr int kMaxSampleKeys = 1024;    // keys per query the floor kernel selects among ...
constexpr int kMaxSampleTiles = 8192;   // ... and tiles of a sampling launch (taken in groups beyond that many)
struct MfmaPass {
    int tile_stride;
    unsigned int* tile_max;   // [n_qtiles*tnq][plan.n_tiles] order-preserving score bits, or NULL
    const int* run_if;        // not NULL: the launch does nothing while this device word is 0
    int padded_queries;       // queries the caller's query, floor and tile_max buffers hold (the launch needs n_qtiles * tnq)
    unsigned int* sync = nullptr;   // 32 device words, zero between launches, for the form that claims its last tiles (NULL: never that form)
};

// rows x queries on the matrix cores with fused top-k, both in the shard's storage type
// (dtype 0 fp32: exact-fp32 MFMA; 2 bf16).  queries:
// [n_qtiles*tnq][ld], rows past b zero.  shared_floor: n_qtiles*tnq words, zeroed by the
// caller or preset by launch_floor_from_tile_max before every launch.  plan: from plan_mfma for the
// number of rows the pass covers (its tiles x plan.tm).  Outputs (tile_max NULL)
// [b][n_ranges][k] candidate lists (unsorted, padded with (-inf, -1)).
hipError_t launch_score_mfma(const void* rows, int dtype, int64_t n, int ld, const void* queries, int b, int k,
                             const MfmaPlan& plan, const MfmaPass& pass, void* cand_buf, unsigned int* shared_floor,
                             float* out_s, int* out_i, hipStream_t stream);

// shared_floor[q] = k-th largest of tile_max[q][0..n_tiles) for q < b (0 when there are not
// more than k tiles); n_tiles <= kMaxSampleTiles.
hipError_t launch_floor_from_tile_max(const unsigned int* tile_max, int n_tiles, int b, int k,
                                      unsigned int* shared_floor, hipStream_t stream);

constexpr int kWideSubs = 8;  // key lists per (workgroup, query): one per lane that holds the query's scores
struct WidePlan {
    int tq;                   // queries per tile: 128, 192 or 256
    int n_qtiles;             // query tiles
    int n_lists;              // interleaved row lists (workgroups per query tile)
    long long n_tiles;        // row tiles of 256
    int grid;
    size_t lds;
    int cap;                  // key slots per list; every (workgroup, query) pair has kWideSubs lists
    size_t key_bytes;         // [n_qtiles][n_lists][tq][kWideSubs][cap] packed keys
    size_t count_words;       // [n_qtiles][n_lists][tq][kWideSubs]
};

// Launch shape of the 256 x 256-tile filter kernel (bf16 shards, large
// batches); -1 when the shape is not covered.  expected_hits: how many scores
// per query the given floors are expected to let through.
int plan_wide(int dtype, int64_t n, int ld, int b, int k, int expected_hits, int num_cus, WidePlan* plan);

// rows x queries on the matrix cores.  tile_max NULL: every score >= its query's
// floor (floors[q], order-preserving bits as written by
// launch_floor_from_tile_max) is appended as  row | score bits << 32  to
// keys[query tile][list][query of the tile][sub][0..cap) (rows past n in the last tile
// included, as copies of the last row), and counts[...][sub] receives how many
// reached the floor (more than cap: the list is incomplete).  tile_max not
// NULL (sampling form): tile t of the launch is tile t * tile_stride of the
// matrix, and tile_max[q][t] ([n_qtiles*plan.tq][plan.n_tiles], zeroed by the
// caller) receives each query's best score of the tile.  plan: from plan_wide
// for the rows the launch covers (its tiles x 256).
// queries: [n_qtiles*plan.tq][ld] bf16, rows past b zero.
hipError_t launch_score_wide(const void* rows, int64_t n, int ld, const void* queries, int b, const WidePlan& plan,
                             const unsigned int* floors, unsigned long long* keys, unsigned int* counts,
                             int tile_stride, unsigned int* tile_max, hipStream_t stream);

// floors[0..words) = 0 when *flag != 0 (device word); nothing otherwise.
hipError_t launch_reset_floors_if(unsigned int* floors, int words, const int* flag, hipStream_t stream);

// Raises *flag (a device word) when lists of k_in < k_out candidates per query were too
// short for the merged result out_idx / out_score to be certain: some full list's last
// entry beats the merged k_out-th entry.  Lists sorted best first, padded with id -1.
hipError_t launch_check_short_lists(const float* cand_s, const int* cand_i, int n_lists, long long list_stride,
                                    long long query_stride, int k_in, int b, int k_out, long long idx_offset,
                                    const long long* out_idx, const float* out_score, int* flag, hipStream_t stream);

// The same for lists that are each sorted best first (score descending, row
// ascending; short lists padded with id -1 at the end), 32-bit ids: by the
// floor-from-the-heads kernel when the shape is small (k <= 32, all keys in 48 KB),
// through launch_merge_topk otherwise.  Lists of a query are list_stride
// elements apart in both arrays.
hipError_t launch_merge_sorted_lists(const float* cand_s, const int* idx32, int n_lists, long long list_stride,
                                     long long query_stride, int k_in, int b, int k_out, long long idx_offset,
                                     long long* out_idx, float* out_score, hipStream_t stream);

// The best k_out of each query's key lists written by launch_score_wide, in
// (score descending, row ascending) order, ignoring keys of rows >= n_rows (the
// last tile's padding); *overflow is set to 1 when a list of any query was
// incomplete (the outputs of that query are then not valid).
// n_lists lists per query in groups of subs ([query tile][n_lists / subs][tile_queries][subs][cap]).
hipError_t launch_select_keys(const unsigned long long* keys, const unsigned int* counts, int n_lists, int subs, int tile_queries, int cap,
                              int n_rows, int b, int k_out, long long idx_offset, long long* out_idx, float* out_score, int* overflow,
                              hipStream_t stream);

}  // namespace mtr
