/*
 * medtutor_retrieval.h -- C ABI of the MI355X-native local dense retriever.
 *
 * The reference (yale-nlp/medical-rag, "MedTutor") has no FFI of its own for
 * this path: its boundary is the Python surface of pipeline/utils/local_rag.py.
 * Each entry point below names the reference lines whose work it takes over;
 * the Python module medical-rag_amd/local_rag.py binds them with ctypes and
 * keeps the reference's four function names.
 *
 * Conventions
 *  - C linkage, plain pointers and sizes, no exceptions cross the boundary.
 *  - Return value: MTR_OK or a negative MTR_ERR_* code; mtr_last_error()
 *    returns a thread-local message for the last failing call.
 *  - An index handle owns one device-resident row shard on ONE GPU: rows are
 *    stored row-major, L2-normalised (when asked), fp32 or bf16, with the row
 *    stride padded to a multiple of 16 bytes (of 128 bytes when that adds at
 *    most an eighth to the row).  Row ids reported by a search
 *    are  row_offset + local row  (int64), so shards of one corpus on several
 *    GPUs (one process per GPU) report global ids.
 *  - Result rows are ordered by (score descending, row id ascending); slots
 *    beyond the number of rows hold id -1 and score -inf.
 *  - Threads and streams.  Host calls on one handle are serialised by an
 *    internal mutex.  A handle's scratch buffers are shared by all its searches:
 *    every call records an event behind the work it enqueued, and a call that
 *    enqueues on a different stream first makes that stream wait for the event,
 *    so searches on one handle run one after the other on the GPU whatever
 *    streams they were given.
 */
#ifndef MEDTUTOR_RETRIEVAL_H
#define MEDTUTOR_RETRIEVAL_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mtr_index mtr_index_t;
typedef struct mtr_comm mtr_comm_t;
typedef struct mtr_multi_index mtr_multi_index_t;
typedef struct mtr_json_index mtr_json_index_t;

enum {
    MTR_OK = 0,
    MTR_ERR_INVALID = -1,   /* bad argument / unsupported size */
    MTR_ERR_NONFINITE = -2, /* NaN or infinity in rows or queries (sklearn raises ValueError) */
    MTR_ERR_HIP = -3,       /* HIP runtime error */
    MTR_ERR_COMM = -4,      /* collective (RCCL) error */
    MTR_ERR_OOM = -5        /* device allocation failed */
};

enum { MTR_F32 = 0, MTR_F64 = 1, MTR_BF16 = 2, MTR_F16 = 3 /* mtr_pool_normalize input only */ };
/* OR-ed into mtr_index_load_rows' src_dtype: the rows are already in the
 * shard's stored form (unit length or zero when the shard is normalised), e.g.
 * read back from a bf16 index file: they are checked for NaN/inf, cast and
 * padded, but not normalised again (a second rounding would change them). */
enum { MTR_ROWS_AS_STORED = 256 };
enum { MTR_COSINE = 0, MTR_IP = 1 };

#define MTR_MAX_K 1024

/* Allocates an empty shard of n rows x d columns on GPU device_id.
 * store_dtype: MTR_F32 or MTR_BF16.  normalize != 0: rows are L2-normalised
 * as they are loaded (norms below 10*DBL_EPSILON count as 1, as scikit-learn's
 * normalize does), which is what lets cosine search run as a pure inner
 * product.  Takes over the device-side half of load_embeddings
 * (pipeline/utils/local_rag.py:90-101) and the per-call float64 matrix rebuild
 * of retrieve_top_index (pipeline/utils/local_rag.py:76). */
int mtr_index_alloc(int64_t n, int32_t d, int32_t store_dtype, int32_t normalize,
                    int32_t device_id, int64_t row_offset, mtr_index_t** out);

/* Converts n_rows rows (src_dtype MTR_F32/F64/BF16, dense row-major, d
 * columns) into the shard starting at local row row_start: finiteness check,
 * optional normalisation, cast, pad.  rows_on_device != 0: `rows` is a device
 * pointer on the shard's GPU; otherwise a host pointer (copied in chunks).
 * Returns MTR_ERR_NONFINITE when a NaN/inf was seen (rows are then unusable).
 * Replaces the row normalisation inside sklearn cosine_similarity as called at
 * pipeline/utils/local_rag.py:77, done once instead of per query. */
int mtr_index_load_rows(mtr_index_t* index, int64_t row_start, const void* rows, int64_t n_rows,
                        int32_t src_dtype, int32_t rows_on_device);

/* mtr_index_alloc + mtr_index_load_rows of all n rows from host memory. */
int mtr_index_create(const void* host_rows, int64_t n, int32_t d, int32_t src_dtype,
                     int32_t store_dtype, int32_t normalize, int32_t device_id,
                     int64_t row_offset, mtr_index_t** out);

/* Exact top-k of b queries against the shard; blocks until out_idx[b*k] and
 * out_score[b*k] (host memory) are filled.  queries: b x d dense row-major,
 * q_dtype MTR_F32/F64/BF16, host pointer or (queries_on_device != 0) device
 * pointer.  metric MTR_COSINE normalises each query (the shard must have been
 * built with normalize != 0); MTR_IP uses the queries as given.
 * Replaces retrieve_top_index (pipeline/utils/local_rag.py:71-78): column 0 of
 * the result is that function's return value; columns 1..k-1 extend its
 * np.argmax tie rule (lowest row first). */
int mtr_index_search(mtr_index_t* index, const void* queries, int32_t q_dtype,
                     int32_t queries_on_device, int32_t b, int32_t k, int32_t metric,
                     int64_t* out_idx, float* out_score);

/* Same search with device-resident queries and outputs, enqueued on `stream`
 * (a hipStream_t; NULL = HIP's default stream) without synchronising.  The
 * caller synchronises the stream before reading dev_out_*.  A NaN/inf in the
 * queries cannot fail a call that does not wait for the GPU: it is recorded, and
 * mtr_index_nonfinite reports it afterwards.  Used by the multi-GPU path (local
 * candidates feed the all-gather) and by benchmarks. */
int mtr_index_search_device(mtr_index_t* index, const void* dev_queries, int32_t q_dtype,
                            int32_t b, int32_t k, int32_t metric,
                            int64_t* dev_out_idx, float* dev_out_score, void* stream);

/* mtr_index_search_device for callers that cannot allocate device memory: code
 * under HIP stream capture, tight serving loops, harnesses that forbid
 * hipMalloc.  No counterpart in the reference.  Same arguments, same kernels and
 * bit-identical results, but every scratch buffer (query tiles, candidate lists,
 * floors) comes from a file-scope device array of MTR_NOALLOC_SCRATCH_BYTES that
 * each GPU holds from library load to process end, and the call makes no
 * hipMalloc / hipFree or other allocating call (no profiling events either).
 * Limits: the search's scratch must fit the array -- every single-query search
 * and every batch with b <= 64 and k <= 128 does; larger shapes are accepted
 * when their launch plan fits and rejected with MTR_ERR_INVALID (the message
 * names the bytes needed) otherwise.  All calls on one GPU, from any handle,
 * share that GPU's array: calls that overlap on one GPU (different streams or
 * threads without ordering between them) race; callers keep to one stream per
 * GPU or order the calls themselves.  The handle's other entry points keep their
 * own buffers and may be mixed in freely. */
#define MTR_NOALLOC_SCRATCH_BYTES (192u << 20)
int mtr_index_search_device_noalloc(mtr_index_t* index, const void* dev_queries, int32_t q_dtype,
                                    int32_t b, int32_t k, int32_t metric,
                                    int64_t* dev_out_idx, float* dev_out_score, void* stream);

/* Whether the queries of the last mtr_index_search_device call on the handle
 * held a NaN or infinity (*nonfinite = 1; its results are then meaningless, and
 * the reference's scorer raises ValueError: pipeline/utils/local_rag.py:77).
 * Waits for that search to finish.  After mtr_index_search_device_noalloc,
 * which records no event, the caller synchronises with its stream first; the
 * report is then that of the last execution of that call, also when it was
 * captured into a HIP graph and replayed (each replay clears and sets a word
 * of its own, no value is frozen into the graph). */
int mtr_index_nonfinite(mtr_index_t* index, int32_t* nonfinite);

/* k-way merge of per-shard candidate lists on GPU device_id: inputs are
 * n_lists blocks of b x k_in (id, score) pairs laid out [list][b][k_in] --
 * exactly what an all-gather of each rank's mtr_index_search_device output
 * produces; ids < 0 are padding.  idx_list_stride / score_list_stride give the
 * distance between consecutive lists in elements of the respective array
 * (0 = dense, b*k_in), so ids and scores may sit in one gathered buffer.  Writes the best k_out per query ordered by
 * (score desc, id asc).  This is the exchange step of the row-sharded search;
 * the reference is single-shard (pipeline/utils/local_rag.py:78 is its whole
 * "merge"). */
int mtr_merge_topk_device(const int64_t* dev_idx, const float* dev_score, int32_t n_lists,
                          int64_t idx_list_stride, int64_t score_list_stride,
                          int32_t b, int32_t k_in, int32_t k_out,
                          int64_t* dev_out_idx, float* dev_out_score,
                          int32_t device_id, void* stream);

/* ---- row-sharded search: one process per GPU, RCCL over xGMI ---------------
 * The reference is single-shard (pipeline/utils/local_rag.py:78 is its whole
 * "merge"); these entry points are what lets any host search a corpus that
 * spans the GPUs of a node.  Each process holds the contiguous row block of its
 * rank (row_offset = first global row) and a communicator; a search is
 * local top-k -> ONE all-gather of b*k*12 bytes per rank -> k-way merge, and
 * every rank ends with the same global result.  RCCL is loaded on first use. */
#define MTR_COMM_ID_BYTES 128
/* Rank 0 makes the id (MTR_COMM_ID_BYTES bytes) and ships it to the other ranks
 * by any means (a file, a socket, torch.distributed's store). */
int mtr_comm_unique_id(void* out_id);
/* Collective: every rank calls it with the same id; blocks until all have. */
int mtr_comm_create(const void* unique_id, int32_t rank, int32_t world, int32_t device_id,
                    mtr_comm_t** out);
int mtr_comm_info(const mtr_comm_t* comm, int32_t* rank, int32_t* world, int32_t* device_id);
void mtr_comm_destroy(mtr_comm_t* comm);
/* Collective, enqueued on `stream` without synchronising: mtr_index_search_device
 * on this rank's shard, the all-gather, the merge; dev_out_* (b*k each) receive
 * the global top-k, identical on every rank.  index may be NULL for a rank that
 * holds no rows (it contributes padding).  Every rank passes the same queries,
 * b, k and metric.  Up to four calls may be in flight on different streams.
 * Failures.  A rank whose own search fails (bad arguments for its shard, no
 * memory for its scratch) still enters the all-gather with a block of padding
 * and runs the merge, so that the other ranks are not left waiting, and returns
 * its error afterwards: the other ranks' results then lack that rank's rows, and
 * it is for the failing rank's caller to stop the job.  A rank that cannot even
 * get its exchange buffers (MTR_ERR_OOM from this call or from
 * mtr_comm_exchange_merge_device) misses the collective: its communicator then
 * fails every later call with MTR_ERR_COMM, and the peers' pending all-gather
 * does not complete until their communicators are destroyed. */
int mtr_sharded_search_device(mtr_index_t* index, mtr_comm_t* comm, const void* dev_queries,
                              int32_t q_dtype, int32_t b, int32_t k, int32_t metric,
                              int64_t* dev_out_idx, float* dev_out_score, void* stream);
/* The exchange step alone, for callers that overlap it with the next search on
 * another stream: this rank's local candidates (b x k_in; both NULL = none) are
 * all-gathered and merged into the global best k_out per query. */
int mtr_comm_exchange_merge_device(mtr_comm_t* comm, const int64_t* dev_local_idx,
                                   const float* dev_local_score, int32_t b, int32_t k_in,
                                   int32_t k_out, int64_t* dev_out_idx, float* dev_out_score,
                                   void* stream);

/* ---- row-sharded search: ONE process, several GPUs --------------------------
 * The reference's caller is one process holding one index object
 * (pipeline/async_main.py:190-195 loads it, pipeline/utils/retrieval.py:70-76
 * searches it inline).  A multi index is that one object for a corpus spread
 * over the GPUs of a node: device_ids lists the GPUs (n_devices entries, 1-64),
 * block i of the rows -- contiguous, n / n_devices rows each, the first
 * n % n_devices blocks one more -- lives as an ordinary shard with global row
 * ids on device_ids[i].  Queries arrive at, and results leave from,
 * device_ids[0].  A search enqueues, from the calling thread and without
 * waiting in between: the queries to every GPU, every block's local top-k on a
 * stream of its own, ONE exchange of the packed b*k*12-byte candidate blocks,
 * and the k-way merge on device_ids[0].
 * exchange: MTR_EXCHANGE_RCCL = a grouped ncclAllGather over the communicators
 * of ncclCommInitAll (xGMI; the devices must be distinct);
 * MTR_EXCHANGE_COPY = one peer copy per block into device_ids[0]'s gather
 * buffer behind an event (no RCCL needed; also what a list that names a device
 * more than once uses, e.g. a test of the block logic on one GPU);
 * MTR_EXCHANGE_AUTO = RCCL when the list holds several distinct devices and
 * RCCL loads, copies otherwise.  Results are those of one shard holding all the
 * rows: same order, same tie rule (lowest row first). */
enum { MTR_EXCHANGE_AUTO = 0, MTR_EXCHANGE_COPY = 1, MTR_EXCHANGE_RCCL = 2 };
int mtr_multi_index_alloc(int64_t n, int32_t d, int32_t store_dtype, int32_t normalize,
                          const int32_t* device_ids, int32_t n_devices, int32_t exchange,
                          mtr_multi_index_t** out);
/* Host rows [row_start, row_start + n_rows) of the corpus into the blocks that
 * hold them (src_dtype as for mtr_index_load_rows, MTR_ROWS_AS_STORED
 * included); the parts of different GPUs are uploaded side by side. */
int mtr_multi_index_load_rows(mtr_multi_index_t* index, int64_t row_start, const void* host_rows,
                              int64_t n_rows, int32_t src_dtype);
/* mtr_multi_index_alloc + mtr_multi_index_load_rows of all n rows: the
 * multi-GPU counterpart of mtr_index_create (load_embeddings,
 * pipeline/utils/local_rag.py:90-101). */
int mtr_multi_index_create(const void* host_rows, int64_t n, int32_t d, int32_t src_dtype,
                           int32_t store_dtype, int32_t normalize, const int32_t* device_ids,
                           int32_t n_devices, int32_t exchange, mtr_multi_index_t** out);
/* Exact top-k of b host queries over all blocks; blocks until out_idx[b*k] and
 * out_score[b*k] (host) are filled.  Same contract as mtr_index_search
 * (retrieve_top_index, pipeline/utils/local_rag.py:71-78, is b = k = 1). */
int mtr_multi_index_search(mtr_multi_index_t* index, const void* queries, int32_t q_dtype,
                           int32_t b, int32_t k, int32_t metric, int64_t* out_idx, float* out_score);
/* The same with queries and outputs resident on device_ids[0], enqueued behind
 * `stream` (a stream of that GPU) without synchronising: the caller's stream
 * continues behind the merge.  Non-finite queries are reported by
 * mtr_multi_index_nonfinite afterwards, as for mtr_index_search_device. */
int mtr_multi_index_search_device(mtr_multi_index_t* index, const void* dev_queries, int32_t q_dtype,
                                  int32_t b, int32_t k, int32_t metric, int64_t* dev_out_idx,
                                  float* dev_out_score, void* stream);
int mtr_multi_index_nonfinite(mtr_multi_index_t* index, int32_t* nonfinite);
/* Shape, storage, number of blocks and the exchange form in use (never AUTO).
 * Any out pointer may be NULL. */
int mtr_multi_index_info(const mtr_multi_index_t* index, int64_t* n, int32_t* d, int32_t* store_dtype,
                         int32_t* normalized, int32_t* n_blocks, int32_t* exchange);
/* Block i: its shard (borrowed; NULL when the block holds no rows: fewer rows
 * than blocks), GPU and row range [row_lo, row_hi). */
int mtr_multi_index_block(mtr_multi_index_t* index, int32_t i, mtr_index_t** shard,
                          int32_t* device_id, int64_t* row_lo, int64_t* row_hi);
/* Rows [row_start, row_start + n_rows) as stored, fp32 (n_rows x d), host. */
int mtr_multi_index_read_rows(mtr_multi_index_t* index, int64_t row_start, int64_t n_rows,
                              float* host_out);
void mtr_multi_index_destroy(mtr_multi_index_t* index);

/* Encoder epilogue on GPU device_id: masked mean over the t tokens of each of
 * b sequences, then L2 normalisation (eps 1e-12), written as fp32 rows that
 * mtr_index_search_device / mtr_index_load_rows accept directly.  dev_hidden:
 * [b][t][d] in MTR_F32, MTR_BF16 or MTR_F16; dev_mask: [b][t] int64 (the
 * tokenizer's attention_mask).  Enqueued on `stream`, no synchronisation.
 * Replaces the pooling arithmetic at pipeline/utils/local_rag.py:61-65 and
 * pipeline/utils/embed.py:68-72; the encoder forward stays in PyTorch. */
int mtr_pool_normalize(const void* dev_hidden, int32_t dtype, const int64_t* dev_mask, int32_t b,
                       int32_t t, int32_t d, float* dev_out, int32_t device_id, void* stream);

/* Reader of the reference's index file, host side only (no GPU): one JSON
 * list of page objects, each with an "embedding" array of d numbers
 * (written at pipeline/utils/embed.py:74-79,110-113).  Takes over the json.load
 * of pipeline/utils/local_rag.py:96-98 for the vectors, which never become
 * Python lists: mtr_json_index_scan walks `text` (len bytes of UTF-8, which
 * must stay mapped until mtr_json_index_free) and finds every page's array;
 * mtr_json_index_info gives the page count, d and the byte length of the
 * "rest" -- the same text with each of those arrays replaced by null, for the
 * caller's own JSON parser to turn into the page dicts; mtr_json_index_read
 * converts the arrays into `matrix` (n_pages x d, MTR_F64 or MTR_F32, split
 * over `threads` host threads) and copies the rest into `rest` (rest_len
 * bytes, no terminator; NULL to skip).  Numbers follow Python's json module
 * (JSON plus NaN, Infinity, -Infinity) with correct rounding, so an MTR_F64
 * matrix equals numpy.array(json.load(...)) value for value.  A file that is
 * anything else than such a list (or has rows of different lengths) is
 * declined with MTR_ERR_INVALID, by either call: use a general parser then. */
int mtr_json_index_scan(const char* text, int64_t len, mtr_json_index_t** out);
int mtr_json_index_info(const mtr_json_index_t* parsed, int64_t* n_pages, int32_t* d, int64_t* rest_len);
int mtr_json_index_read(const mtr_json_index_t* parsed, void* matrix, int32_t dtype, char* rest, int32_t threads);
void mtr_json_index_free(mtr_json_index_t* parsed);

/* Shape and storage of a shard. Any out pointer may be NULL. */
int mtr_index_info(const mtr_index_t* index, int64_t* n, int32_t* d, int32_t* store_dtype,
                   int32_t* normalized, int32_t* device_id, int64_t* row_offset);

/* Copies rows [row_start, row_start+n_rows) back to host as fp32 (n_rows x d),
 * as stored (normalised, rounded to the storage type).  For parity tests. */
int mtr_index_read_rows(mtr_index_t* index, int64_t row_start, int64_t n_rows, float* host_out);

/* Per-launch HIP-event timing of the scoring kernel (the scan over the rows),
 * recorded on the stream it is launched on.  enable != 0 starts recording (and
 * clears earlier samples); mtr_index_scan_time synchronises the recorded
 * events and returns their count and total milliseconds. */
int mtr_index_profile(mtr_index_t* index, int32_t enable);
int mtr_index_scan_time(mtr_index_t* index, int32_t* n_launches, double* total_ms);
/* The same samples one by one: up to cap durations (milliseconds, in launch
 * order) into out_ms, their full count into n_launches. */
int mtr_index_scan_samples(mtr_index_t* index, float* out_ms, int32_t cap, int32_t* n_launches);

/* Measurement aid: runs the single-query scan's loads WITHOUT its arithmetic
 * over the shard (same rows per wave, sweeps, cache policy and launch shape)
 * reps times and returns the average milliseconds by HIP events: what the
 * memory system delivers to the scan's access pattern on this GPU, the ceiling
 * beside which the scan's own time is reported.  No counterpart in the
 * reference. */
int mtr_index_read_ceiling(mtr_index_t* index, int32_t reps, float* avg_ms);

/* The name of the scoring kernel the last search enqueued on the shard took
 * ("scan_topk_kernel", "scan_stream_kernel", "score_topk_mfma_kernel" or
 * "score_filter_wide_kernel"; "" before the first search), NUL-terminated
 * into buf (cap bytes).  For benchmark reports and profiles, which otherwise
 * would have to restate the library's launch rules.  No counterpart in the
 * reference. */
int mtr_index_last_kernel(const mtr_index_t* index, char* buf, int32_t cap);

void mtr_index_destroy(mtr_index_t* index);

/* Process-wide launch-shape switches, for parity tests (which force the
 * kernel forms of large shards onto small ones) and timing sweeps; not needed
 * for normal use.  Names: mfma_min_batch, mfma_tnq, mfma_per_cu, mfma_cap,
 * mfma_sample_stride, mfma_stream_rows, mfma_stages, mfma_tile_rows,
 * mfma_interleave, mfma_dynamic_tail, mfma_static_permille, mfma_via_regs,
 * wide_min_batch, wide_qgroup, wide_tile_queries, wide_stream_rows,
 * scan_wg_per_cu, scan_nontemporal, scan_max_qb, scan_short_lists,
 * scan_form, scan_static_permille, scan_chunk_groups, scan_mid_permille,
 * scan_pair_permille, scan_late_refill,
 * pool_sliced.  Each starts from the environment variable MTR_<NAME> if set.  Unknown name: MTR_ERR_INVALID.
 * No counterpart in the reference. */
int mtr_tuning_set(const char* name, int32_t value);
int mtr_tuning_get(const char* name, int32_t* value);

/* Thread-local message of the last failing call on this thread ("" if none). */
const char* mtr_last_error(void);

/* Library version string. */
const char* mtr_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MEDTUTOR_RETRIEVAL_H */
