// Candidate merge: the best k of several sorted-or-not candidate lists per
// query, ordered by (score descending, row id ascending).
//
// Used twice on the path: (1) after the row scan, to merge the per-workgroup
// lists of one shard, and (2) after the all-gather of the row-sharded search,
// to merge the per-GPU lists.  The reference's counterpart is the single
// np.argmax at pipeline/utils/local_rag.py:78.
//
// One workgroup per query, three kernels.  merge_heads_kernel (few short sorted
// lists with 32-bit ids, k <= 32: what the row scan leaves) takes the k-th
// largest list head as a floor and ranks the few keys that reach it.  merge_select_kernel (32-bit local row
// ids, k <= 256: the per-workgroup and per-range lists of one shard) selects
// the k-th largest packed key of each wave's slice with a radix walk and folds
// the survivors; it does not need sorted input.  merge_topk_kernel (64-bit ids
// or larger k: the per-GPU lists after the all-gather) streams the candidates
// through per-wave top-k lists (wave_topk.h); its producers write sorted lists,
// which it uses to bound the k-th best from the list heads first.
#include "mtr_device.h"
#include "wave_select.h"
#include "wave_topk.h"

namespace mtr {

namespace {

struct MergeArgs {
    const float* cs;
    const int* ci32;
    const long long* ci64;
    int n_lists;
    long long list_stride;    // elements between lists in the id array
    long long list_stride_s;  // ... and in the score array
    long long query_stride;
    int k_in;
    int k_out;
    long long idx_offset;
    long long* out_i;
    float* out_s;
    int heads_first;  // lists are sorted: use the list heads to bound the k-th best first
    // Key-list form (merge_select_kernel only): list l of query q holds
    // min(key_counts[...], k_in) entries  row | score bits << 32  instead of (score, id) pairs.
    // Lists come in groups of key_subs (one group per producing workgroup).
    const unsigned long long* keys;   // [query tile][n_lists / key_subs][key_tile][key_subs][k_in]
    const unsigned int* key_counts;   // [query tile][n_lists / key_subs][key_tile][key_subs]
    int key_subs;
    int key_tile;                     // queries per tile
    int* overflow;                    // set to 1 when a count exceeds k_in
    int key_rows;                     // keys of rows >= key_rows (tile padding) are dropped
    const int* run_if;                // not NULL: the kernel does nothing while *run_if == 0
};

struct Cand {
    float s;
    long long id;
    bool ok;
};

__device__ __forceinline__ Cand load_cand(const MergeArgs& a, long long qbase, long long l, long long j, bool in_range) {
    Cand c{0.f, -1, false};
    if (in_range) {
        const long long off = l * a.list_stride + qbase + j;
        c.s = a.cs[l * a.list_stride_s + qbase + j];
        c.id = a.ci32 ? (long long)a.ci32[off] : a.ci64[off];
        c.ok = c.id >= 0;
        if (a.ci32) c.id += a.idx_offset;
    }
    return c;
}

// Folds the lists of waves 1.. into wave 0's (called by wave 0 after a barrier).
__device__ __forceinline__ void fold_wave_lists(WaveTopK<long long>& tk, const float* list_s, const long long* list_i,
                                                const int* cnt_lds, int nwaves, int k, int lane) {
    for (int w = 1; w < nwaves; ++w) {
        const int c = cnt_lds[w];
        const float* ws = list_s + (size_t)w * k;
        const long long* wi = list_i + (size_t)w * k;
        for (int j0 = 0; j0 < c; j0 += 64) {
            const int j = j0 + lane;
            const bool ok = j < c;
            tk.push(ok ? ws[j] : 0.f, ok ? wi[j] : 0ll, ok);
        }
    }
}

constexpr int kQueueCap = 512;  // survivors staged in LDS for wave 0

__global__ void merge_topk_kernel(MergeArgs a) {
    if (a.run_if && *a.run_if == 0) return;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nwaves = blockDim.x >> 6;
    long long* list_i = reinterpret_cast<long long*>(smem);
    long long* queue_i = list_i + (size_t)nwaves * a.k_out;
    float* list_s = reinterpret_cast<float*>(queue_i + kQueueCap);
    float* queue_s = list_s + (size_t)nwaves * a.k_out;
    int* cnt_lds = reinterpret_cast<int*>(queue_s + kQueueCap);
    float* floor_lds = reinterpret_cast<float*>(cnt_lds + nwaves);
    int* qcount = reinterpret_cast<int*>(floor_lds + 1);

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int q = blockIdx.x;
    const long long qbase = (long long)q * a.query_stride;
    constexpr int kUnroll = 8;

    WaveTopK<long long> tk;
    tk.init(list_s + (size_t)wave * a.k_out, list_i + (size_t)wave * a.k_out, a.k_out);
    if (threadIdx.x == 0) {
        qcount[0] = 0;
        floor_lds[0] = -__builtin_huge_valf();
    }
    __syncthreads();

    // Pass 1 (many lists only): the k-th best of the list HEADS is a floor for
    // the k-th best overall; with sorted lists almost nothing else clears it,
    // so pass 2 keeps a handful of candidates instead of thousands.  One wave
    // does this: folding per-wave head lists would cost more than it saves.
    if (a.heads_first) {
        if (wave == 0) {
            for (long long l0 = 0; l0 < a.n_lists; l0 += 64 * kUnroll) {
                Cand c[kUnroll];
#pragma unroll
                for (int u = 0; u < kUnroll; ++u) {
                    const long long l = l0 + u * 64 + lane;
                    c[u] = load_cand(a, qbase, l, 0, l < a.n_lists);
                }
#pragma unroll
                for (int u = 0; u < kUnroll; ++u) tk.push(c[u].s, c[u].id, c[u].ok);
            }
            if (lane == 0 && tk.cnt == a.k_out) floor_lds[0] = tk.thr_s;
            tk.init(list_s, list_i, a.k_out);
        }
        __syncthreads();
    }
    const float floor_s = floor_lds[0];

    // Pass 2: candidates that reach the floor go to the LDS queue (or, if it is
    // full, through this wave's own list).  Loads are issued kUnroll at a time.
    const long long total = (long long)a.n_lists * a.k_in;
    const long long stride = (long long)nwaves * 64;
    for (long long m0 = (long long)wave * 64; m0 < total; m0 += stride * kUnroll) {
        Cand c[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const long long m = m0 + u * stride + lane;
            const long long l = m / a.k_in;
            c[u] = load_cand(a, qbase, l, m - l * a.k_in, m < total);
        }
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            bool keep = c[u].ok && c[u].s >= floor_s;
            if (keep) {
                const int pos = atomicAdd(qcount, 1);
                if (pos < kQueueCap) {
                    queue_s[pos] = c[u].s;
                    queue_i[pos] = c[u].id;
                    keep = false;
                }
            }
            tk.push(c[u].s, c[u].id, keep);  // queue overflow only
        }
    }

    if (lane == 0) cnt_lds[wave] = tk.cnt;
    __syncthreads();
    if (wave == 0) {
        const int nq = qcount[0] < kQueueCap ? qcount[0] : kQueueCap;
        for (int j0 = 0; j0 < nq; j0 += 64) {
            const int j = j0 + lane;
            const bool ok = j < nq;
            tk.push(ok ? queue_s[j] : 0.f, ok ? queue_i[j] : 0ll, ok);
        }
        fold_wave_lists(tk, list_s, list_i, cnt_lds, nwaves, a.k_out, lane);
        float* os = a.out_s + (size_t)q * a.k_out;
        long long* oi = a.out_i + (size_t)q * a.k_out;
        for (int j = lane; j < a.k_out; j += 64) {
            const bool ok = j < tk.cnt;
            os[j] = ok ? tk.ls[j] : -__builtin_huge_valf();
            oi[j] = ok ? tk.li[j] : -1ll;
        }
    }
}


// ---------------------------------------------------------------------------
// Selection-based merge for 32-bit row ids and k <= 256.  Instead of inserting
// candidates one by one, every wave packs its slice into 64-bit keys
// (wave_select.h), finds the k-th largest with a radix walk and keeps the
// survivors; the waves' survivor lists are then folded in rounds of
// kFanIn-or-fewer lists per wave, and the last <= k keys are rank-sorted.
// ---------------------------------------------------------------------------
constexpr int kSelE = 8;                 // keys per lane: 512 slots per wave
constexpr int kSelSlots = kSelE * 64;

// Keeps the k largest of key[] (exact), writes them compacted to list[0..),
// returns how many.
__device__ __forceinline__ int select_into(const unsigned long long (&key)[kSelE], int k, unsigned long long* list,
                                           unsigned int* hist, int lane) {
    int passes = 0;
    const unsigned long long cut = wave_kth_largest_hist<kSelE>(key, k, 0, hist, lane, &passes);
    int base = 0;
#pragma unroll
    for (int e = 0; e < kSelE; ++e) {
        const bool keep = key[e] != 0ull && key[e] >= cut;
        const unsigned long long m = __ballot(keep);
        if (keep) list[base + __popcll(m & ((1ull << lane) - 1ull))] = key[e];
        base += __popcll(m);
    }
    return base;
}

__global__ void merge_select_kernel(MergeArgs a) {
    if (a.run_if && *a.run_if == 0) return;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nwaves = blockDim.x >> 6;
    const int k = a.k_out;
    // [nwaves][k]; key-list form: [nwaves][kSelSlots], new keys staged behind the k survivors
    const int lstride = a.keys ? kSelSlots : k;
    unsigned long long* lists = reinterpret_cast<unsigned long long*>(smem);
    unsigned int* hist_all = reinterpret_cast<unsigned int*>(lists + (size_t)nwaves * lstride);  // [nwaves][256]
    int* cnt_lds = reinterpret_cast<int*>(hist_all + nwaves * 256);

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int q = blockIdx.x;
    const long long qbase = (long long)q * a.query_stride;
    unsigned long long* my_list = lists + (size_t)wave * lstride;
    unsigned int* hist = hist_all + wave * 256;

    // Level 0: each wave streams a contiguous slice of the candidates through
    // "survivors so far + a chunk of new ones -> select k".
    const long long total = (long long)a.n_lists * a.k_in;
    const long long per_wave = (total + nwaves - 1) / nwaves;
    const long long lo = wave * per_wave;
    const long long hi = lo + per_wave < total ? lo + per_wave : total;
    const int chunk = kSelSlots - k;  // new candidates per round (k <= 256, so >= 256)
    int cnt = 0;
    if (a.keys) {
        // Key lists: wave w takes lists w, w + nwaves, ... and copies their keys
        // behind its survivors in LDS; whenever `chunk` new keys are staged (and
        // at the end) survivors + staged keys -> select k.
        const int groups = a.n_lists / a.key_subs;
        int staged = 0;
        auto select_round = [&]() {
            unsigned long long key[kSelE];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int e = 0; e < kSelE; ++e) {
                const int slot = e * 64 + lane;
                key[e] = (slot < k ? slot < cnt : slot - k < staged) ? my_list[slot] : 0ull;
            }
            __builtin_amdgcn_wave_barrier();
            cnt = select_into(key, k, my_list, hist, lane);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            staged = 0;
        };
        // Most lists are empty or hold a key or two (a query's hits are spread over
        // thousands of lists), so the lists are taken 64 at a time, one per lane: the
        // counts with one load, the first two keys of every list with one load each,
        // compacted behind the survivors; only longer lists are then walked one by one.
        auto slot_of = [&](int l) {
            return (((size_t)(q / a.key_tile) * groups + l / a.key_subs) * a.key_tile + q % a.key_tile) * a.key_subs + l % a.key_subs;
        };
        auto stage_key = [&](unsigned long long raw) {
            const int row = (int)(unsigned int)raw;
            return row < a.key_rows ? pack_key(__uint_as_float((unsigned int)(raw >> 32)), row) : 0ull;
        };
        constexpr int kDirect = 2;
        for (int l0 = wave; l0 < a.n_lists; l0 += nwaves * 64) {
            const int my_l = l0 + lane * nwaves;
            unsigned int my_have = 0;
            size_t my_slot0 = 0;
            if (my_l < a.n_lists) {
                my_slot0 = slot_of(my_l);
                my_have = a.key_counts[my_slot0];
            }
            if (my_have > (unsigned int)a.k_in) {
                *a.overflow = 1;
                my_have = (unsigned int)a.k_in;
            }
            if (__ballot(my_have != 0) == 0ull) continue;
#pragma unroll
            for (int t = 0; t < kDirect; ++t) {
                const bool has = my_have > (unsigned int)t;
                const unsigned long long mask = __ballot(has);
                if (mask == 0ull) break;
                if (staged + 64 > chunk) select_round();
                if (has) my_list[k + staged + __popcll(mask & ((1ull << lane) - 1ull))] = stage_key(a.keys[my_slot0 * a.k_in + t]);
                staged += __popcll(mask);
            }
            unsigned long long longer = __ballot(my_have > (unsigned int)kDirect);
            while (longer) {
                const int j = __ffsll((long long)longer) - 1;
                longer &= longer - 1;
                const int have = __builtin_amdgcn_readlane((int)my_have, j);
                const unsigned long long* src = a.keys + slot_of(l0 + j * nwaves) * a.k_in;
                int done = kDirect;
                while (done < have) {
                    if (staged == chunk) select_round();
                    const int take = have - done < chunk - staged ? have - done : chunk - staged;
                    for (int i = lane; i < take; i += 64) my_list[k + staged + i] = stage_key(src[done + i]);
                    staged += take;
                    done += take;
                }
            }
        }
        if (staged > 0) select_round();
    } else
    for (long long c0 = lo; c0 < hi; c0 += chunk) {
        unsigned long long key[kSelE];
#pragma unroll
        for (int e = 0; e < kSelE; ++e) {
            const int slot = e * 64 + lane;
            unsigned long long v = 0ull;
            if (slot < k) {
                if (slot < cnt) v = my_list[slot];
            } else {
                const long long m = c0 + (slot - k);
                if (m < hi) {
                    const long long l = m / a.k_in;
                    const long long j = m - l * a.k_in;
                    const int id = a.ci32[l * a.list_stride + qbase + j];
                    if (id >= 0) v = pack_key(a.cs[l * a.list_stride_s + qbase + j], id);
                }
            }
            key[e] = v;
        }
        __builtin_amdgcn_wave_barrier();
        cnt = select_into(key, k, my_list, hist, lane);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) cnt_lds[wave] = cnt;
    __syncthreads();

    // Fold rounds: wave g takes lists [g*fan, g*fan + fan), all loads before any write.
    const int fan = kSelSlots / k < 2 ? 2 : kSelSlots / k;
    int n = nwaves;
    while (n > 1) {
        const int groups = (n + fan - 1) / fan;
        unsigned long long key[kSelE];
        const bool active = wave < groups;
        if (active) {
#pragma unroll
            for (int e = 0; e < kSelE; ++e) {
                const int slot = e * 64 + lane;
                const int j = slot / k;          // which list of my group
                const int i = slot - j * k;
                const int src = wave * fan + j;
                key[e] = (j < fan && src < n && i < cnt_lds[src]) ? lists[(size_t)src * lstride + i] : 0ull;
            }
        }
        __syncthreads();
        if (active) {
            const int c = select_into(key, k, my_list, hist, lane);
            if (lane == 0) cnt_lds[wave] = c;
        }
        __syncthreads();
        n = groups;
    }

    // Rank sort of the final <= k keys (wave 0) and output.
    if (wave == 0) {
        const int c = cnt_lds[0];
        float* os = a.out_s + (size_t)q * k;
        long long* oi = a.out_i + (size_t)q * k;
        for (int i0 = 0; i0 < k; i0 += 64) {
            const int i = i0 + lane;
            const unsigned long long mine = i < c ? lists[i] : 0ull;
            int rank = 0;
            for (int j = 0; j < c; ++j) rank += lists[j] > mine ? 1 : 0;  // LDS broadcast reads
            if (i < c) {
                os[rank] = key_score(mine);
                oi[rank] = (long long)key_row(mine) + a.idx_offset;
            }
        }
        for (int i = c + lane; i < k; i += 64) {
            os[i] = -__builtin_huge_valf();
            oi[i] = -1ll;
        }
    }
}

// Few short SORTED lists (the row scan's per-workgroup lists, best first), by
// counting instead of selecting: the k-th largest list HEAD is a floor for the
// k-th best overall, only the k lists whose heads reach it can hold anything
// above it (at most k * k_in keys), and those survivors are ordered by rank.
// Ranks come from LDS broadcast reads; there is no cross-lane dependency chain.
constexpr int kHeadsThreads = 256;
__global__ __launch_bounds__(kHeadsThreads) void merge_heads_kernel(MergeArgs a) {
    if (a.run_if && *a.run_if == 0) return;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int total = a.n_lists * a.k_in;
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem);  // [n_lists][k_in], 0 = empty
    unsigned long long* surv = keys + total;                                   // [k_out * k_in]
    unsigned long long* floor_lds = surv + (size_t)a.k_out * a.k_in;
    int* count_lds = reinterpret_cast<int*>(floor_lds + 1);
    const int tid = threadIdx.x;
    const int q = blockIdx.x;
    const long long qbase = (long long)q * a.query_stride;
    // All of a thread's loads are issued before the first is used: taken one at
    // a time they cost a memory round trip each, which was most of the kernel.
    constexpr int kBatch = 8;
    for (int i0 = tid; i0 < total; i0 += kHeadsThreads * kBatch) {
        int id[kBatch];
        float sc[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            const int i = i0 + u * kHeadsThreads;
            const int ic = i < total ? i : 0;
            const int l = ic / a.k_in, j = ic - l * a.k_in;
            id[u] = a.ci32[l * a.list_stride + qbase + j];
            sc[u] = a.cs[l * a.list_stride_s + qbase + j];
        }
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            const int i = i0 + u * kHeadsThreads;
            if (i < total) keys[i] = id[u] >= 0 ? pack_key(sc[u], id[u]) : 0ull;
        }
    }
    if (tid == 0) {
        floor_lds[0] = 0ull;   // fewer than k_out non-empty lists: everything survives
        count_lds[0] = 0;
    }
    __syncthreads();
    // The head that k_out - 1 other heads beat is the floor (heads are unique or empty).
    for (int l = tid; l < a.n_lists; l += kHeadsThreads) {
        const unsigned long long mine = keys[(size_t)l * a.k_in];
        if (mine == 0ull) continue;
        int rank = 0;
        for (int o = 0; o < a.n_lists; ++o) rank += keys[(size_t)o * a.k_in] > mine ? 1 : 0;
        if (rank == a.k_out - 1) floor_lds[0] = mine;
    }
    __syncthreads();
    const unsigned long long floor = floor_lds[0];
    for (int l = tid; l < a.n_lists; l += kHeadsThreads) {
        for (int j = 0; j < a.k_in; ++j) {
            const unsigned long long key = keys[(size_t)l * a.k_in + j];
            if (key == 0ull || key < floor) break;   // sorted: nothing further down the list reaches the floor
            surv[atomicAdd(&count_lds[0], 1)] = key;
        }
    }
    __syncthreads();
    const int c = count_lds[0];
    float* os = a.out_s + (size_t)q * a.k_out;
    long long* oi = a.out_i + (size_t)q * a.k_out;
    for (int i = tid; i < c; i += kHeadsThreads) {
        const unsigned long long mine = surv[i];
        int rank = 0;
        for (int o = 0; o < c; ++o) rank += surv[o] > mine ? 1 : 0;
        if (rank < a.k_out) {
            os[rank] = key_score(mine);
            oi[rank] = (long long)key_row(mine) + a.idx_offset;
        }
    }
    for (int i = c + tid; i < a.k_out; i += kHeadsThreads) {
        os[i] = -__builtin_huge_valf();
        oi[i] = -1ll;
    }
}


// ---------------------------------------------------------------------------
// Long result lists (256 < k <= 1024) from up to 16 K candidates with 32-bit
// ids: one workgroup of 1024 threads per query holds every candidate key in
// registers, finds the k-th largest with a radix walk over the whole
// workgroup (eight 8-bit digits, an LDS histogram per digit), compacts the k
// survivors into LDS and orders them by rank counting over LDS broadcast
// reads.  The streaming merge above inserts nearly every candidate into lists
// of k entries (0.4 ms for 8 K candidates and k = 1024); this takes 20 us.
// ---------------------------------------------------------------------------
constexpr int kRadixThreads = 1024;

template <int E>
__global__ __launch_bounds__(kRadixThreads) void merge_radix_kernel(MergeArgs a) {
    if (a.run_if && *a.run_if == 0) return;
    __shared__ unsigned int hist[256];
    __shared__ unsigned int state[4];               // [0] bin picked, [1] candidates still wanted from it, [2] survivors
    __shared__ unsigned long long surv[1024];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int q = blockIdx.x;
    const long long qbase = (long long)q * a.query_stride;
    const int total = a.n_lists * a.k_in;
    const int k = a.k_out;

    unsigned long long key[E];
    bool act[E];
    int mine = 0;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = e * kRadixThreads + tid;
        key[e] = 0ull;
        if (i < total) {
            const int l = i / a.k_in, j = i - l * a.k_in;
            const int id = a.ci32[l * a.list_stride + qbase + j];
            if (id >= 0) key[e] = pack_key(a.cs[l * a.list_stride_s + qbase + j], id);
        }
        act[e] = key[e] != 0ull;
        mine += act[e] ? 1 : 0;
    }
    if (tid < 4) state[tid] = 0u;
    __syncthreads();
    if (mine) atomicAdd(&state[2], (unsigned int)mine);
    __syncthreads();
    const int have = (int)state[2];
    __syncthreads();

    unsigned long long cut = 0ull;   // have <= k: every candidate survives
    if (have > k) {
        int need = k;
        unsigned long long prefix = 0ull;
        for (int pass = 0; pass < 8; ++pass) {
            const int shift = 56 - 8 * pass;
            if (tid < 256) hist[tid] = 0u;
            __syncthreads();
#pragma unroll
            for (int e = 0; e < E; ++e)
                if (act[e]) atomicAdd(&hist[(unsigned int)(key[e] >> shift) & 255u], 1u);
            __syncthreads();
            if (tid < 64) {   // wave 0: lane l looks after bins 4l .. 4l+3; higher bins hold larger keys
                const unsigned int h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
                const unsigned int sum = h0 + h1 + h2 + h3;
                unsigned int incl = sum;   // inclusive sum over lanes >= this one
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const unsigned int t = __shfl_down(incl, off, 64);
                    if (lane + off < 64) incl += t;
                }
                const unsigned int a3 = incl - sum, a2 = a3 + h3, a1 = a2 + h2, a0 = a1 + h1;
                const unsigned int un = (unsigned int)need;
                if (a3 < un && un <= a3 + h3) { state[0] = 4 * lane + 3; state[1] = un - a3; }
                else if (a2 < un && un <= a2 + h2) { state[0] = 4 * lane + 2; state[1] = un - a2; }
                else if (a1 < un && un <= a1 + h1) { state[0] = 4 * lane + 1; state[1] = un - a1; }
                else if (a0 < un && un <= a0 + h0) { state[0] = 4 * lane; state[1] = un - a0; }
            }
            __syncthreads();
            const unsigned int bin = state[0];
            need = (int)state[1];
            prefix |= (unsigned long long)bin << shift;
#pragma unroll
            for (int e = 0; e < E; ++e) act[e] = act[e] && (((unsigned int)(key[e] >> shift) & 255u) == bin);
            __syncthreads();
        }
        cut = prefix;   // keys are unique: the walk ends on the k-th largest key itself
    }
    if (tid == 0) state[2] = 0u;
    __syncthreads();
#pragma unroll
    for (int e = 0; e < E; ++e)
        if (key[e] != 0ull && key[e] >= cut) {
            const unsigned int slot = atomicAdd(&state[2], 1u);
            if (slot < 1024u) surv[slot] = key[e];
        }
    __syncthreads();
    const int c = (int)(state[2] < 1024u ? state[2] : 1024u);
    float* os = a.out_s + (size_t)q * k;
    long long* oi = a.out_i + (size_t)q * k;
    if (tid < c) {
        const unsigned long long me = surv[tid];
        int rank = 0;
        for (int o = 0; o < c; ++o) rank += surv[o] > me ? 1 : 0;   // LDS broadcast reads
        if (rank < k) {
            os[rank] = key_score(me);
            oi[rank] = (long long)key_row(me) + a.idx_offset;
        }
    }
    for (int i = c + tid; i < k; i += kRadixThreads) {
        os[i] = -__builtin_huge_valf();
        oi[i] = -1ll;
    }
}

// Were the lists long enough?  A scan may keep fewer than k_out candidates per
// list (k_in < k_out: a list's share of a query's k_out best rows is small when
// the rows are dealt evenly).  The merged result is then exact unless some full
// list's LAST kept entry beats the merged k_out-th entry: only then can a row
// the list dropped (all of them rank below its last kept entry) belong to the
// result.  Raises *flag in that case; the caller re-runs the search with full
// lists behind it.
__global__ void check_short_lists_kernel(const float* cand_s, const int* cand_i, int n_lists, long long list_stride,
                                         long long query_stride, int k_in, int k_out, long long idx_offset,
                                         const long long* out_idx, const float* out_score, int* flag) {
    const int q = blockIdx.x;
    const long long gk_i = out_idx[(size_t)q * k_out + k_out - 1];
    const float gk_s = out_score[(size_t)q * k_out + k_out - 1];
    for (int l = threadIdx.x; l < n_lists; l += blockDim.x) {
        const long long at = (long long)l * list_stride + (long long)q * query_stride + k_in - 1;
        const int id = cand_i[at];
        if (id < 0) continue;   // the list is not full: it dropped nothing
        const float s = cand_s[at];
        if (gk_i < 0 || s > gk_s || (s == gk_s && (long long)id + idx_offset < gk_i)) *flag = 1;
    }
}

}  // namespace

hipError_t launch_merge_sorted_lists(const float* cand_s, const int* idx32, int n_lists, long long list_stride,
                                     long long query_stride, int k_in, int b, int k_out, long long idx_offset,
                                     long long* out_idx, float* out_score, hipStream_t stream) {
    const size_t lds = ((size_t)n_lists * k_in + (size_t)k_out * k_in + 1) * 8 + 16;
    if (k_out > 32 || k_in > 32 || n_lists > 1024 || lds > 48 * 1024)  // long merges: the selection kernel scales better
        return launch_merge_topk(cand_s, idx32, nullptr, n_lists, list_stride, list_stride, query_stride, k_in, b, k_out,
                                 idx_offset, out_idx, out_score, stream, nullptr);
    if (b < 1 || k_out < 1 || k_in < 1 || n_lists < 1) return hipErrorInvalidValue;
    MergeArgs a{cand_s, idx32, nullptr, n_lists, list_stride, list_stride, query_stride, k_in, k_out, idx_offset, out_idx, out_score,
                0, nullptr, nullptr, 1, 256, nullptr, 0, nullptr};
    hipLaunchKernelGGL(merge_heads_kernel, dim3(b), dim3(kHeadsThreads), lds, stream, a);
    return hipGetLastError();
}

hipError_t launch_merge_topk(const float* cand_s, const int* idx32, const long long* idx64, int n_lists,
                             long long list_stride, long long list_stride_s, long long query_stride, int k_in, int b, int k_out,
                             long long idx_offset, long long* out_idx, float* out_score, hipStream_t stream,
                             const int* run_if) {
    if (b < 1 || k_out < 1 || k_in < 1 || n_lists < 1) return hipErrorInvalidValue;
    MergeArgs a{cand_s, idx32, idx64, n_lists, list_stride, list_stride_s, query_stride, k_in, k_out, idx_offset, out_idx, out_score,
                (n_lists >= 4 * k_out && n_lists >= 128) ? 1 : 0, nullptr, nullptr, 1, 256, nullptr, 0, run_if};
    const long long total = (long long)n_lists * k_in;
    if (idx32 && k_out <= 256 && total > 64) {
        // one wave per 512-slot chunk of candidates, at most 16
        long long want = (total + (kSelSlots - k_out) - 1) / (kSelSlots - k_out);
        const int nw = (int)(want < 1 ? 1 : want > 16 ? 16 : want);
        const size_t lds_sel = (size_t)nw * k_out * 8 + (size_t)nw * 1024 + (size_t)nw * 4 + 16;
        hipLaunchKernelGGL(merge_select_kernel, dim3(b), dim3(nw * 64), lds_sel, stream, a);
        return hipGetLastError();
    }
    if (idx32 && k_out <= 1024 && total <= 16 * kRadixThreads && total > 2 * k_out) {
        if (total <= 8 * kRadixThreads) hipLaunchKernelGGL(merge_radix_kernel<8>, dim3(b), dim3(kRadixThreads), 0, stream, a);
        else hipLaunchKernelGGL(merge_radix_kernel<16>, dim3(b), dim3(kRadixThreads), 0, stream, a);
        return hipGetLastError();
    }
    int nwaves = 16;
    while (nwaves > 1 && (long long)nwaves * 64 * 2 > total) nwaves /= 2;            // no idle waves on short inputs
    while (nwaves > 1 && (size_t)nwaves * k_out * 12 + kQueueCap * 12 + 128 > 60 * 1024) nwaves /= 2;  // LDS budget
    const size_t lds = (size_t)nwaves * k_out * 12 + (size_t)kQueueCap * 12 + (size_t)nwaves * 4 + 16;
    hipLaunchKernelGGL(merge_topk_kernel, dim3(b), dim3(nwaves * 64), lds, stream, a);
    return hipGetLastError();
}

hipError_t launch_check_short_lists(const float* cand_s, const int* cand_i, int n_lists, long long list_stride,
                                    long long query_stride, int k_in, int b, int k_out, long long idx_offset,
                                    const long long* out_idx, const float* out_score, int* flag, hipStream_t stream) {
    if (b < 1 || n_lists < 1 || k_in < 1 || k_out < 1 || !flag) return hipErrorInvalidValue;
    hipLaunchKernelGGL(check_short_lists_kernel, dim3(b), dim3(256), 0, stream, cand_s, cand_i, n_lists, list_stride,
                       query_stride, k_in, k_out, idx_offset, out_idx, out_score, flag);
    return hipGetLastError();
}

hipError_t launch_select_keys(const unsigned long long* keys, const unsigned int* counts, int n_lists, int subs, int tile_queries, int cap,
                              int n_rows, int b, int k_out, long long idx_offset, long long* out_idx, float* out_score,
                              int* overflow, hipStream_t stream) {
    if (b < 1 || k_out < 1 || k_out > 256 || cap < 1 || n_lists < 1 || subs < 1 || n_lists % subs != 0 || tile_queries < 1)
        return hipErrorInvalidValue;
    MergeArgs a{nullptr, nullptr, nullptr, n_lists, 0, 0, 0, cap, k_out, idx_offset, out_idx, out_score, 0,
                keys, counts, subs, tile_queries, overflow, n_rows, nullptr};
    const int nw = n_lists < 8 ? n_lists : 8;
    const size_t lds_sel = (size_t)nw * kSelSlots * 8 + (size_t)nw * 1024 + (size_t)nw * 4 + 16;
    hipLaunchKernelGGL(merge_select_kernel, dim3(b), dim3(nw * 64), lds_sel, stream, a);
    return hipGetLastError();
}

}  // namespace mtr
