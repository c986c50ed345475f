// Row scan for small query batches: an HBM-bound sweep over the stored rows
// with the top-k selection fused in, so no score row is ever written.
//
// Replaces the per-query work of retrieve_top_index
// (pipeline/utils/local_rag.py:76-78): the float64 matrix rebuild, sklearn's
// re-normalisation of all rows, the dgemv and the argmax.
//
// Layout of the work
//   - Sixteen lanes share a row: one 64-lane load instruction fetches 256
//     contiguous bytes of each of FOUR consecutive rows (lane l takes the
//     16-byte chunk  s*16 + l%16  of row  l/16  in sweep s), straight into
//     registers.  A wave owns groups of 8 consecutive rows (two such row
//     quads, grid-strided, so a wave sees its rows in increasing order) and
//     keeps 8 loads -- four sweeps of both quads, 1 KiB of each row -- in flight.
//   - The QB queries of the pass live in LDS as zero-padded fp32; a lane reads
//     the chunk that matches its row chunk, so chunks past the end of a row
//     multiply by zero instead of being masked.
//   - Each lane accumulates one partial dot product per (quad, query); the
//     sixteen partial sums of a row are added with four DPP steps (lane
//     permutes inside the VALU: no LDS crossbar traffic and no wait, which is
//     what a 64-lane shuffle butterfly cost: 1 % of the scan's time).
//   - Totals go to the wave's per-query top-k lists (wave_topk.h); the eight
//     waves of a workgroup fold their lists pairwise
//     at the end and the workgroup writes one sorted candidate list per query.
//
//
// Single queries on shards beyond the caches take the streaming form
// (scan_stream_kernel): four waves per workgroup, groups of sixteen rows, two
// register sets so that the next batch of loads is requested before the
// current one is consumed, and a dynamic split of the last rows between the
// waves (see there).
//
// Algorithmic bytes per launch: n * ld * sizeof(T) (each stored row byte is
// read once); the query and candidate traffic is negligible beside it.
#include <stdlib.h>

#include <algorithm>

#include "mtr_device.h"
#include "wave_topk.h"

namespace mtr {

namespace {

// Waves per workgroup.  Eight waves, with ONE workgroup per CU for a single
// query, read fastest: a bare read loop of this access pattern peaks there too
// (7.17 TB/s against 7.08 with sixteen waves, profiles/micro/read_patterns.hip),
// and few workgroups keep the number of candidate lists (one each) small.
constexpr int waves_per_group(int) { return 8; }
constexpr int kLanesPerRow = 16;   // lanes that share a row: 256 contiguous bytes per row and load
constexpr int kQuads = 2;          // row quads per wave group: 8 rows
constexpr int kSweepsPerBatch = 4; // sweeps of both quads requested together: 8 loads in flight per wave
constexpr int kGroupRows = 4 * kQuads;

template <bool NT>
__device__ __forceinline__ u32x4 load_chunk(const u32x4* p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}

// Sum over the sixteen lanes of a DPP row, left in every one of them: two
// quad permutes (xor 1, xor 2), then rotations by 4 and by 8 lanes.
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_move<0xB1>(v);    // quad_perm [1, 0, 3, 2]
    v += dpp_move<0x4E>(v);    // quad_perm [2, 3, 0, 1]
    v += dpp_move<0x124>(v);   // row_ror 4
    v += dpp_move<0x128>(v);   // row_ror 8
    return v;
}

struct ScanArgs {
    const char* rows;
    long long n;
    long long row_bytes;
    int nchunks;   // 16-byte chunks per row that hold data
    int nit;       // sweeps per row = ceil(nchunks / 16)
    const float* q;       // prepared queries [QB][qstride] (used when q_raw is null)
    const float* q_raw;   // or: nb raw fp32 queries [nb][d], normalised and padded by every workgroup itself
    int d;
    int nb;
    int normalize;
    NonfiniteFlag nonfinite_flag;  // raised when a raw query holds NaN/inf
    int qstride;   // floats per padded query = nit * 16 * EPC
    int k;
    float* cand_s; // [QB][gridDim.x][k]
    int* cand_i;
    const int* run_if;   // not NULL: the kernel does nothing while this device word is 0
};

template <typename T, int QB, bool NT>
__global__ __launch_bounds__(waves_per_group(QB) * 64) void scan_topk_kernel(ScanArgs a) {
    constexpr int kWaves = waves_per_group(QB);
    constexpr int EPC = Chunk<T>::EPC;
    if (a.run_if && *a.run_if == 0) return;
    static_assert(kQuads * QB <= kLanesPerRow, "one lane of a row's sixteen per (quad, query) total");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* q_lds = reinterpret_cast<float*>(smem);
    float* list_s = q_lds + (size_t)QB * a.qstride;
    int* list_i = reinterpret_cast<int*>(list_s + kWaves * QB * a.k);
    int* cnt_lds = list_i + kWaves * QB * a.k;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;

    const long long ngroups = (a.n + kGroupRows - 1) / kGroupRows;
    const long long gstride = (long long)gridDim.x * kWaves;
    const int col = lane & (kLanesPerRow - 1);   // this lane's chunk within a sweep
    const int lrow = lane >> 4;                  // ... and its row within a quad
    const long long g_first = (long long)blockIdx.x * kWaves + wave;

    // The rows do not depend on the queries: the wave's first batch of row loads
    // goes out before the query set-up below (two dependent rounds of global
    // loads and a barrier, about 5 us during which the memory system would idle).
    auto request_batch = [&](long long g, int s0, u32x4 (&raw)[kSweepsPerBatch][kQuads]) {
        const char* rp[kQuads];
#pragma unroll
        for (int qd = 0; qd < kQuads; ++qd) {
            long long row = g * kGroupRows + qd * 4 + lrow;
            row = row < a.n ? row : a.n - 1;
            rp[qd] = a.rows + row * a.row_bytes;
        }
#pragma unroll
        for (int u = 0; u < kSweepsPerBatch; ++u) {
            if (s0 + u < a.nit) {   // wave-uniform
                const int chunk = (s0 + u) * kLanesPerRow + col;
                // Chunks past the row's data read chunk 0 instead (always in
                // bounds); their query values are zero.
                const int chunk_c = chunk < a.nchunks ? chunk : 0;
#pragma unroll
                for (int qd = 0; qd < kQuads; ++qd)
                    raw[u][qd] = load_chunk<NT>(reinterpret_cast<const u32x4*>(rp[qd] + (size_t)chunk_c * 16));
            }
        }
    };
    u32x4 raw[kSweepsPerBatch][kQuads];
    bool requested = g_first < ngroups;   // wave-uniform: raw holds (or awaits) batch 0 of group g_first
    if (requested) request_batch(g_first, 0, raw);

    if (a.q_raw) {
        // fp32 queries straight from the caller: float64 norm (zero norm counts as
        // 1, as for the rows), scale, zero-pad -- redone by every workgroup, which
        // is cheaper than a kernel boundary.
        // One wave per query (waves take queries round-robin), so no workgroup
        // barrier is needed until all queries are in place.
        for (int b = wave; b < QB; b += kWaves) {
            float* dst = q_lds + (size_t)b * a.qstride;
            if (b >= a.nb) {
                for (int e = lane; e < a.qstride; e += 64) dst[e] = 0.f;
                continue;
            }
            const float* src = a.q_raw + (size_t)b * a.d;
            double ss = 0.0;
            bool bad = false;
            // A lane's loads go out sixteen at a time (one at a time they cost a
            // memory round trip each, ten microseconds before the first row is read).
            constexpr int kBatch = 16;
            for (int e0 = lane; e0 < a.d; e0 += 64 * kBatch) {
                float v[kBatch];
#pragma unroll
                for (int u = 0; u < kBatch; ++u) {
                    const int e = e0 + 64 * u;
                    v[u] = src[e < a.d ? e : lane];   // lane < d whenever this loop runs
                }
#pragma unroll
                for (int u = 0; u < kBatch; ++u) {
                    if (e0 + 64 * u < a.d) {
                        const double x = (double)v[u];
                        bad |= !(fabs(x) <= 1.7976931348623157e308);
                        ss += x * x;
                    }
                }
            }
            if (__ballot(bad) != 0ull && lane == 0 && blockIdx.x == 0) atomicMax(a.nonfinite_flag.word, a.nonfinite_flag.value);
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) {
                int lo = __double2loint(ss), hi = __double2hiint(ss);
                lo = __shfl_xor(lo, m, 64);
                hi = __shfl_xor(hi, m, 64);
                ss += __hiloint2double(hi, lo);
            }
            double norm = 1.0;
            if (a.normalize) {
                norm = sqrt(ss);
                if (!(norm >= 10.0 * 2.220446049250313e-16)) norm = 1.0;
            }
            for (int e0 = lane; e0 < a.qstride; e0 += 64 * kBatch) {
                float v[kBatch];
#pragma unroll
                for (int u = 0; u < kBatch; ++u) {
                    const int e = e0 + 64 * u;
                    v[u] = src[e < a.d ? e : 0];
                }
#pragma unroll
                for (int u = 0; u < kBatch; ++u) {
                    const int e = e0 + 64 * u;
                    if (e < a.qstride) dst[e] = e < a.d ? (float)((double)v[u] / norm) : 0.f;
                }
            }
        }
    } else {
        const int n4 = QB * a.qstride / 4;
        const f32x4* src = reinterpret_cast<const f32x4*>(a.q);
        f32x4* dst = reinterpret_cast<f32x4*>(q_lds);
        for (int i = tid; i < n4; i += kWaves * 64) dst[i] = src[i];
    }
    __syncthreads();

    WaveTopK<int> tk[QB];
#pragma unroll
    for (int b = 0; b < QB; ++b) tk[b].init(list_s + (wave * QB + b) * a.k, list_i + (wave * QB + b) * a.k, a.k);

    // The lane that reports the total of (quad, query): col = quad * QB + query.
    const int my_quad = col / QB;
    const int my_b = col % QB;
    const bool owner = col < kQuads * QB;

    for (long long g = g_first; g < ngroups; g += gstride) {
        const long long row0 = g * kGroupRows;
        float acc[kQuads][QB];
#pragma unroll
        for (int qd = 0; qd < kQuads; ++qd)
#pragma unroll
            for (int b = 0; b < QB; ++b) acc[qd][b] = 0.f;

        for (int s0 = 0; s0 < a.nit; s0 += kSweepsPerBatch) {
            if (!requested) request_batch(g, s0, raw);
            requested = false;
#pragma unroll
            for (int u = 0; u < kSweepsPerBatch; ++u) {
                if (s0 + u < a.nit) {
                    const int chunk = (s0 + u) * kLanesPerRow + col;
                    float qv[QB][EPC];
#pragma unroll
                    for (int b = 0; b < QB; ++b) {
                        const f32x4* qp = reinterpret_cast<const f32x4*>(q_lds + (size_t)b * a.qstride + (size_t)chunk * EPC);
#pragma unroll
                        for (int e4 = 0; e4 < EPC / 4; ++e4) {
                            const f32x4 t = qp[e4];
                            qv[b][4 * e4 + 0] = t[0]; qv[b][4 * e4 + 1] = t[1];
                            qv[b][4 * e4 + 2] = t[2]; qv[b][4 * e4 + 3] = t[3];
                        }
                    }
#pragma unroll
                    for (int qd = 0; qd < kQuads; ++qd) {
                        float f[EPC];
                        Chunk<T>::unpack(raw[u][qd], f);
#pragma unroll
                        for (int b = 0; b < QB; ++b) {
#pragma unroll
                            for (int e = 0; e < EPC; ++e) acc[qd][b] = __builtin_fmaf(f[e], qv[b][e], acc[qd][b]);
                        }
                    }
                }
            }
        }

        float total = 0.f;
#pragma unroll
        for (int qd = 0; qd < kQuads; ++qd)
#pragma unroll
            for (int b = 0; b < QB; ++b) {
                const float t = row16_sum(acc[qd][b]);
                if (my_quad == qd && my_b == b) total = t;
            }
        const long long row = row0 + my_quad * 4 + lrow;
        const bool valid = owner && row < a.n;
#pragma unroll
        for (int b = 0; b < QB; ++b) tk[b].push(total, (int)row, valid && my_b == b);
    }

    // Fold the waves' lists pairwise (wave w takes wave w + half, halving the
    // number of live lists per round), then wave 0 writes the workgroup's lists.
#pragma unroll
    for (int half = kWaves / 2; half >= 1; half >>= 1) {
        if (lane == 0 && wave < 2 * half) {
#pragma unroll
            for (int b = 0; b < QB; ++b) cnt_lds[wave * QB + b] = tk[b].cnt;
        }
        __syncthreads();
        if (wave < half) {
            const int w = wave + half;
#pragma unroll
            for (int b = 0; b < QB; ++b) {
                const int c = cnt_lds[w * QB + b];
                const float* ws = list_s + (w * QB + b) * a.k;
                const int* wi = list_i + (w * QB + b) * a.k;
                for (int j0 = 0; j0 < c; j0 += 64) {
                    const int j = j0 + lane;
                    const bool ok = j < c;
                    const float s = ok ? ws[j] : 0.f;
                    const int i = ok ? wi[j] : 0;
                    tk[b].push(s, i, ok);
                }
            }
        }
        __syncthreads();
    }
    if (wave == 0) {
#pragma unroll
        for (int b = 0; b < QB; ++b) {
            float* os = a.cand_s + ((size_t)b * gridDim.x + blockIdx.x) * a.k;
            int* oi = a.cand_i + ((size_t)b * gridDim.x + blockIdx.x) * a.k;
            for (int j = lane; j < a.k; j += 64) {
                const bool ok = j < tk[b].cnt;
                os[j] = ok ? tk[b].ls[j] : -__builtin_huge_valf();
                oi[j] = ok ? tk[b].li[j] : -1;
            }
        }
    }
}

// The scan's loads and nothing else (same rows per wave, same sweeps, same
// cache policy; the loaded words are only xor-ed together): what the memory
// system delivers to this access pattern on this shard, i.e. the ceiling the
// scan's time is to be compared with.
template <bool NT>
__global__ __launch_bounds__(waves_per_group(1) * 64) void scan_read_only_kernel(const char* rows, long long n,
                                                                                 long long row_bytes, int nchunks, int nit,
                                                                                 unsigned int* sink) {
    constexpr int kWaves = waves_per_group(1);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & (kLanesPerRow - 1), lrow = lane >> 4;
    const long long ngroups = (n + kGroupRows - 1) / kGroupRows;
    const long long gstride = (long long)gridDim.x * kWaves;
    unsigned int x = 0;
    for (long long g = (long long)blockIdx.x * kWaves + wave; g < ngroups; g += gstride) {
        const char* rp[kQuads];
#pragma unroll
        for (int qd = 0; qd < kQuads; ++qd) {
            long long row = g * kGroupRows + qd * 4 + lrow;
            row = row < n ? row : n - 1;
            rp[qd] = rows + row * row_bytes;
        }
        for (int s0 = 0; s0 < nit; s0 += kSweepsPerBatch) {
            u32x4 raw[kSweepsPerBatch][kQuads];
#pragma unroll
            for (int u = 0; u < kSweepsPerBatch; ++u)
                if (s0 + u < nit) {
                    const int chunk = (s0 + u) * kLanesPerRow + col;
                    const int chunk_c = chunk < nchunks ? chunk : 0;
#pragma unroll
                    for (int qd = 0; qd < kQuads; ++qd)
                        raw[u][qd] = load_chunk<NT>(reinterpret_cast<const u32x4*>(rp[qd] + (size_t)chunk_c * 16));
                }
#pragma unroll
            for (int u = 0; u < kSweepsPerBatch; ++u)
                if (s0 + u < nit)
#pragma unroll
                    for (int qd = 0; qd < kQuads; ++qd) x ^= raw[u][qd][0] ^ raw[u][qd][1] ^ raw[u][qd][2] ^ raw[u][qd][3];
        }
    }
    if (x == 0x9e3779b9u) sink[0] = x;   // never on real data in practice; keeps the loads alive
}

// ---- streaming form: one query, shards beyond the caches ----------------------
//
// What the memory system delivers to a CU differs between CUs by several per
// cent (with the rows split evenly, the first workgroups of a 2.85 ms scan
// finish at 2.6 ms), so an even split ends when the slowest CU does.  Here a
// wave takes a fixed share of the groups first (most of them: wave w of W takes
// groups w, w + W, ...), and the rest is handed out through a device-wide
// counter, group by group.  The claims are few (the
// counter takes about one claim per 5 ns) and are made by a fifth wave of the
// workgroup, which does nothing else: an atomic with return sits in its wave's
// queue of loads, which return in order, so in a scanning wave a slow claim
// would hold back the rows requested behind it.  The claimer keeps up to two
// claimed chunks per scanning wave in LDS boxes; a wave takes them in turn and
// stops once all hold a slot past the end.  Every claim costs the whole chip
// about 1.5 ns (all claims go to one address, so to one memory channel, which
// every scanning wave needs as well), so the claimer refills boxes four at a
// time with one atomic, or at once for a wave that has none left.  Deeper boxes
// are no better (three or four per wave, refilled eight or sixteen at a time:
// +0.1 to +0.7 % of the scan's time): groups claimed long before they are read
// widen the stretch of memory the chip reads at any one time.
//
// The scanning loop takes two units (a unit = SWEEPS sweeps of a group's four
// row quads) per trip, one per register set, and requests a unit before it
// consumes the previous one; every wait on the loads is counted.  That needs:
//   - no branch around a load or its consumption (the compiler cannot count
//     loads in flight across a join and waits for all of them): a request
//     past the wave's last unit re-reads that unit, and its sums are dropped;
//   - nothing in flight at the loop's entry that the loop does not know of (the
//     query set-up's loads are waited for explicitly before it);
//   - the wave index in a scalar register (readfirstlane), so that unit
//     counters and loop tests are scalar;
//   - __launch_bounds__ of the real workgroup size, or the second register set
//     is spilled (the compiler assumes 1024 threads and 128 registers).
constexpr int kStreamWaves = 4;                     // scanning waves; one more wave claims chunks for them
constexpr int kStreamThreads = (kStreamWaves + 1) * 64;
constexpr unsigned int kBoxEmpty = 0xffffffffu;
constexpr unsigned int kBoxPastEnd = 0xfffffffeu;   // a slot past the end of any shard
constexpr unsigned long long kStreamGiveUpTicks = 2000000000ull;   // 20 s of the 100 MHz wall clock
constexpr int kBoxDepth = 2;                        // claimed chunks kept ready per scanning wave
constexpr int kLowReady = 0;                        // a wave with this many chunks left (or fewer) is refilled at once
constexpr int kRefillEmpties = kStreamWaves * (kBoxDepth > 2 ? kBoxDepth - 2 : 1);   // empty boxes that make a refill worth an atomic
constexpr int kStreamQuads = 4;                     // 16 rows per group
constexpr int kStreamGroupRows = 4 * kStreamQuads;

constexpr int kStreamPhases = 3;
struct StreamSplit {
    long long ngroups;          // groups of 16 rows
    long long static_groups;    // groups every wave takes by its index (wave w: w, w + W, ...)
    long long dyn_first;        // first group of the claimed part = static_groups * W
    // The claimed part: phase p hands out slots[p] chunks of chunk[p] groups; the
    // chunks shrink towards the end (the last phase is single groups, as many as are left).
    long long slots[kStreamPhases];
    int chunk[kStreamPhases];
    long long late_from;        // from this slot on: one box per wave, refilled on demand
    unsigned int* sync;         // [0] claim counter, [16] workgroups whose claimer has finished; both zero between launches
};

__device__ __forceinline__ bool stream_chunk(const StreamSplit& sp, long long slot, long long& g0, long long& g1) {
    long long first = sp.dyn_first;
#pragma unroll
    for (int p = 0; p < kStreamPhases - 1; ++p) {
        if (slot < sp.slots[p]) { g0 = first + sp.chunk[p] * slot; g1 = g0 + sp.chunk[p]; return true; }
        slot -= sp.slots[p];
        first += sp.chunk[p] * sp.slots[p];
    }
    g0 = first + slot;
    g1 = g0 + 1;
    return g0 < sp.ngroups;
}

struct StreamUnit {
    long long g;
    int s0;
};

template <bool NT, int SWEEPS>
__device__ __forceinline__ void stream_request(const ScanArgs& a, StreamUnit u, int col, int lrow,
                                               u32x4 (&raw)[SWEEPS][kStreamQuads]) {
    const char* rp[kStreamQuads];
#pragma unroll
    for (int qd = 0; qd < kStreamQuads; ++qd) {
        long long row = u.g * kStreamGroupRows + qd * 4 + lrow;
        row = row < a.n ? row : a.n - 1;
        rp[qd] = a.rows + row * a.row_bytes;
    }
#pragma unroll
    for (int s = 0; s < SWEEPS; ++s) {
        const int chunk = (u.s0 + s) * kLanesPerRow + col;
        const int chunk_c = chunk < a.nchunks ? chunk : 0;   // past the row's data: chunk 0, times a zero query value
#pragma unroll
        for (int qd = 0; qd < kStreamQuads; ++qd)
            raw[s][qd] = load_chunk<NT>(reinterpret_cast<const u32x4*>(rp[qd] + (size_t)chunk_c * 16));
    }
}

template <typename T, int SWEEPS, bool ARITH>
__device__ __forceinline__ void stream_consume(const float* q_lds, StreamUnit u, int col,
                                               const u32x4 (&raw)[SWEEPS][kStreamQuads], float (&acc)[kStreamQuads],
                                               unsigned int& x) {
    constexpr int EPC = Chunk<T>::EPC;
#pragma unroll
    for (int s = 0; s < SWEEPS; ++s) {
        if constexpr (ARITH) {
            const int chunk = (u.s0 + s) * kLanesPerRow + col;
            float qv[EPC];
            const f32x4* qp = reinterpret_cast<const f32x4*>(q_lds + (size_t)chunk * EPC);
#pragma unroll
            for (int e4 = 0; e4 < EPC / 4; ++e4) {
                const f32x4 t = qp[e4];
                qv[4 * e4 + 0] = t[0]; qv[4 * e4 + 1] = t[1];
                qv[4 * e4 + 2] = t[2]; qv[4 * e4 + 3] = t[3];
            }
#pragma unroll
            for (int qd = 0; qd < kStreamQuads; ++qd) {
                float f[EPC];
                Chunk<T>::unpack(raw[s][qd], f);
#pragma unroll
                for (int e = 0; e < EPC; ++e) acc[qd] = __builtin_fmaf(f[e], qv[e], acc[qd]);
            }
        } else {
#pragma unroll
            for (int qd = 0; qd < kStreamQuads; ++qd) x ^= raw[s][qd][0] ^ raw[s][qd][1] ^ raw[s][qd][2] ^ raw[s][qd][3];
        }
    }
}

// One raw fp32 query -> normalised, zero-padded fp32 in LDS, by one wave
// (float64 norm; a norm below 10 * DBL_EPSILON counts as 1, as for the rows).
__device__ __forceinline__ void stage_raw_query(const ScanArgs& a, const float* src, float* dst, int lane, bool report) {
    double ss = 0.0;
    bool bad = false;
    constexpr int kBatch = 16;
    for (int e0 = lane; e0 < a.d; e0 += 64 * kBatch) {
        float v[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            const int e = e0 + 64 * u;
            v[u] = src[e < a.d ? e : lane];
        }
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            if (e0 + 64 * u < a.d) {
                const double x = (double)v[u];
                bad |= !(fabs(x) <= 1.7976931348623157e308);
                ss += x * x;
            }
        }
    }
    if (__ballot(bad) != 0ull && lane == 0 && report) atomicMax(a.nonfinite_flag.word, a.nonfinite_flag.value);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        int lo = __double2loint(ss), hi = __double2hiint(ss);
        lo = __shfl_xor(lo, m, 64);
        hi = __shfl_xor(hi, m, 64);
        ss += __hiloint2double(hi, lo);
    }
    double norm = 1.0;
    if (a.normalize) {
        norm = sqrt(ss);
        if (!(norm >= 10.0 * 2.220446049250313e-16)) norm = 1.0;
    }
    for (int e0 = lane; e0 < a.qstride; e0 += 64 * kBatch) {
        float v[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            const int e = e0 + 64 * u;
            v[u] = src[e < a.d ? e : 0];
        }
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            const int e = e0 + 64 * u;
            if (e < a.qstride) dst[e] = e < a.d ? (float)((double)v[u] / norm) : 0.f;
        }
    }
}

// ARITH false: the same loads, claims and waits without the arithmetic and the
// top-k (the ceiling measurement of this form).
template <typename T, bool NT, int SWEEPS, bool ARITH>
__global__ __launch_bounds__(kStreamThreads) void scan_stream_kernel(ScanArgs a, StreamSplit sp) {
    if (a.run_if && *a.run_if == 0) return;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* q_lds = reinterpret_cast<float*>(smem);
    float* list_s = q_lds + a.qstride;
    int* list_i = reinterpret_cast<int*>(list_s + kStreamWaves * a.k);
    int* cnt_lds = list_i + kStreamWaves * a.k;
    unsigned int* box = reinterpret_cast<unsigned int*>(cnt_lds + kStreamWaves);   // [wave][kBoxDepth] claimed slots

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int kFoldBarriers = ARITH ? 4 : 0;   // two per halving round of the final fold

    if (wave == kStreamWaves) {
        // ---- the claimer: lane l keeps box l (wave l / kBoxDepth) filled ----
        constexpr int kBoxes = kStreamWaves * kBoxDepth;
        if (lane < kBoxes) box[lane] = kBoxEmpty;
        __syncthreads();
        bool finished = lane >= kBoxes;   // ... until it has put a slot past the end into it
        const int my_wave_shift = (lane / kBoxDepth) * kBoxDepth;
        constexpr unsigned long long kWaveBoxes = (1ull << kBoxDepth) - 1;
        const unsigned long long give_up_at = wall_clock64() + kStreamGiveUpTicks;
        bool late = false;   // wave-uniform: the single groups at the end are being handed out
        while (__ballot(!finished) != 0ull) {
            const bool empty =
                !finished && __hip_atomic_load(&box[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == kBoxEmpty;
            const unsigned long long empties = __ballot(empty);
            const unsigned long long ready = __ballot(!empty && !finished);   // boxes that hold a chunk
            const int my_ready = __popcll((ready >> my_wave_shift) & kWaveBoxes);
            const unsigned long long my_empties = (empties >> my_wave_shift) & kWaveBoxes;
            bool fill;
            if (!late) {
                // Boxes are refilled four at a time, or at once for a wave with at most one chunk left.
                const bool low = lane < kBoxes && my_ready <= kLowReady && my_empties != 0;
                fill = empty && (__popcll(empties) >= kRefillEmpties || __ballot(low) != 0ull);
            } else {
                // Near the end a chunk queued behind a slow wave is a chunk a fast wave could have
                // scanned: one box per wave, refilled when the wave starts on its chunk.
                fill = empty && my_ready == 0 && lane - my_wave_shift == __ffsll((long long)my_empties) - 1;
            }
            if (fill) {
                // (one atomic for all the lanes that claim: the compiler adds their count and hands out the slots)
                const unsigned int slot = __hip_atomic_fetch_add(sp.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&box[lane], slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                long long c0, c1;
                finished = !stream_chunk(sp, (long long)slot, c0, c1);
                if ((long long)slot >= sp.late_from) late = true;
            }
            late = __ballot(late) != 0ull;
            __builtin_amdgcn_s_sleep(8);
            if (wall_clock64() > give_up_at) {   // never in a working launch: keeps a fault elsewhere from hanging the GPU
                if (lane < kBoxes) box[lane] = kBoxPastEnd;
                break;
            }
        }
        // This workgroup claims nothing more; the last claimer of the launch to get
        // here leaves both counters at zero for the next launch.
        if (lane == 0) {
            const unsigned int before = __hip_atomic_fetch_add(sp.sync + 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (before + 1u == gridDim.x) {
                __hip_atomic_store(sp.sync, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(sp.sync + 16, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
#pragma unroll
        for (int i = 0; i < kFoldBarriers; ++i) __syncthreads();
        return;
    }

    const int col = lane & (kLanesPerRow - 1);
    const int lrow = lane >> 4;
    const long long W = (long long)gridDim.x * kStreamWaves;
    const long long wid = (long long)blockIdx.x * kStreamWaves + wave;

    // Iterator over the wave's units, kept one unit ahead of the consumption.
    long long g = wid, g_end = wid + 1, j_static = 0;
    int s0 = 0;
    bool valid = sp.static_groups > 0;   // the launcher only picks this form for shards with many groups per wave
    int box_turn = 0;
    auto advance = [&]() {
        if (!valid) return;
        s0 += SWEEPS;
        if (s0 < a.nit) return;
        s0 = 0;
        g += 1;
        if (g < g_end) return;
        if (++j_static < sp.static_groups) { g = wid + j_static * W; g_end = g + 1; return; }
        // The next claimed chunk from this wave's boxes: the first one, from the box
        // after the last one taken, that holds a chunk.  The wave is done when every
        // box holds a slot past the end; while some box is empty it waits for the
        // claimer (which refills at once when a wave has nothing left).
        const unsigned long long give_up_at = wall_clock64() + kStreamGiveUpTicks;
        for (;;) {
            int past_end = 0;
            for (int t = 0; t < kBoxDepth; ++t) {
                const int bi = (box_turn + t) % kBoxDepth;
                unsigned int* my_box = box + wave * kBoxDepth + bi;
                const unsigned int slot = (unsigned int)__builtin_amdgcn_readfirstlane(
                    (int)__hip_atomic_load(my_box, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                if (slot == kBoxEmpty) continue;
                long long c0, c1;
                if (stream_chunk(sp, (long long)slot, c0, c1)) {
                    g = c0; g_end = c1;
                    box_turn = (bi + 1) % kBoxDepth;
                    if (lane == 0) __hip_atomic_store(my_box, kBoxEmpty, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    return;
                }
                ++past_end;
            }
            if (past_end == kBoxDepth || wall_clock64() > give_up_at) break;
            __builtin_amdgcn_s_sleep(2);
        }
        valid = false;
    };

    u32x4 ra[SWEEPS][kStreamQuads], rb[SWEEPS][kStreamQuads];
    StreamUnit cur{g, s0};
    bool cur_ok = valid;
    // The rows do not depend on the query: the first unit's loads go out before
    // the query set-up (two dependent rounds of global loads and a barrier).
    if (cur_ok) stream_request<NT, SWEEPS>(a, cur, col, lrow, ra);

    if constexpr (ARITH) {
        if (a.q_raw) {
            if (wave == 0) stage_raw_query(a, a.q_raw, q_lds, lane, blockIdx.x == 0);
        } else {
            const int n4 = a.qstride / 4;
            const f32x4* src = reinterpret_cast<const f32x4*>(a.q);
            f32x4* dst = reinterpret_cast<f32x4*>(q_lds);
            for (int i = tid; i < n4; i += kStreamWaves * 64) dst[i] = src[i];
        }
    }
    __syncthreads();
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): nothing the loop does not know of is in flight when it starts

    WaveTopK<int> tk;
    tk.init(list_s + wave * a.k, list_i + wave * a.k, a.k);
    float acc[kStreamQuads];
#pragma unroll
    for (int qd = 0; qd < kStreamQuads; ++qd) acc[qd] = 0.f;
    unsigned int x = 0;

    auto finish_group = [&](long long grp) {
        if constexpr (ARITH) {
            float total = 0.f;
#pragma unroll
            for (int qd = 0; qd < kStreamQuads; ++qd) {
                const float t = row16_sum(acc[qd]);
                if (col == qd) total = t;
                acc[qd] = 0.f;
            }
            const long long row = grp * kStreamGroupRows + col * 4 + lrow;
            tk.push(total, (int)row, col < kStreamQuads && row < a.n);
        }
    };

    while (cur_ok) {
        advance();
        const bool nx_ok = valid;
        const StreamUnit nx = nx_ok ? StreamUnit{g, s0} : cur;
        stream_request<NT, SWEEPS>(a, nx, col, lrow, rb);
        __builtin_amdgcn_sched_barrier(0);   // or the scheduler moves the consumption in front of the requests
        stream_consume<T, SWEEPS, ARITH>(q_lds, cur, col, ra, acc, x);
        if (cur.s0 + SWEEPS >= a.nit) finish_group(cur.g);
        advance();
        const bool c2_ok = valid;
        const StreamUnit c2 = c2_ok ? StreamUnit{g, s0} : cur;
        stream_request<NT, SWEEPS>(a, c2, col, lrow, ra);
        __builtin_amdgcn_sched_barrier(0);
        // consumed even without a next unit (a copy of the last one; nobody reads its sums)
        stream_consume<T, SWEEPS, ARITH>(q_lds, nx, col, rb, acc, x);
        if (nx_ok && nx.s0 + SWEEPS >= a.nit) finish_group(nx.g);
        cur = c2;
        cur_ok = c2_ok;
    }

    if constexpr (!ARITH) {
        if (x == 0x9e3779b9u) a.cand_i[0] = (int)x;   // never on real data in practice; keeps the loads alive
    } else {
#pragma unroll
        for (int half = kStreamWaves / 2; half >= 1; half >>= 1) {
            if (lane == 0 && wave < 2 * half) cnt_lds[wave] = tk.cnt;
            __syncthreads();
            if (wave < half) {
                const int w = wave + half;
                const int c = cnt_lds[w];
                const float* ws = list_s + w * a.k;
                const int* wi = list_i + w * a.k;
                for (int j0 = 0; j0 < c; j0 += 64) {
                    const int j = j0 + lane;
                    const bool ok = j < c;
                    const float s = ok ? ws[j] : 0.f;
                    const int i = ok ? wi[j] : 0;
                    tk.push(s, i, ok);
                }
            }
            __syncthreads();
        }
        if (wave == 0) {
            float* os = a.cand_s + (size_t)blockIdx.x * a.k;
            int* oi = a.cand_i + (size_t)blockIdx.x * a.k;
            for (int j = lane; j < a.k; j += 64) {
                const bool ok = j < tk.cnt;
                os[j] = ok ? tk.ls[j] : -__builtin_huge_valf();
                oi[j] = ok ? tk.li[j] : -1;
            }
        }
    }
}

size_t stream_lds_bytes(int qstride, int k) {
    return (size_t)qstride * 4 + (size_t)kStreamWaves * k * 8 + (size_t)kStreamWaves * 4 + (size_t)kStreamWaves * kBoxDepth * 4;
}

// Sweeps per unit: the largest of 4, 3, 2 that divides the sweeps of a row (a
// unit never crosses a group); 0 when none does.
int stream_sweeps(int nit) { return nit % 4 == 0 ? 4 : nit % 3 == 0 ? 3 : nit % 2 == 0 ? 2 : 0; }

StreamSplit stream_split(int64_t n, int grid, unsigned int* sync) {
    StreamSplit sp;
    const long long W = (long long)grid * kStreamWaves;
    sp.ngroups = (n + kStreamGroupRows - 1) / kStreamGroupRows;
    // 88 % of the groups by index, the rest claimed group by group, the last 1.5 % on demand.  (Larger chunks would
    // save claims, but a wave that streams a stretch of its own reads slower than waves
    // that sweep neighbouring groups together: 10M x 2048 B, a quarter of the rows claimed,
    // chunks of 4 / 8 / 16 groups: 2.868 / 2.873 / 2.896 ms.  The phases remain for sweeps:
    // tuning scan_static_permille, scan_chunk_groups up to scan_mid_permille, pairs up to
    // scan_pair_permille.)
    const double f_static = tuning().scan_static_permille > 0 ? tuning().scan_static_permille * 1e-3 : 0.88;
    const double f_mid = tuning().scan_mid_permille > 0 ? tuning().scan_mid_permille * 1e-3 : 0.0;
    const double f_pairs = tuning().scan_pair_permille > 0 ? tuning().scan_pair_permille * 1e-3 : 0.0;
    sp.static_groups = (long long)(f_static * (double)sp.ngroups) / W;
    if (sp.static_groups < 1) sp.static_groups = 1;
    sp.dyn_first = sp.static_groups * W;
    sp.chunk[0] = tuning().scan_chunk_groups > 0 ? tuning().scan_chunk_groups : 1;
    sp.chunk[1] = 2;
    sp.chunk[2] = 1;
    long long at = sp.dyn_first;
    const long long mid_end = (long long)(f_mid * (double)sp.ngroups);
    sp.slots[0] = mid_end > at ? (mid_end - at) / sp.chunk[0] : 0;
    at += sp.slots[0] * sp.chunk[0];
    const long long pair_end = (long long)(f_pairs * (double)sp.ngroups);
    sp.slots[1] = pair_end > at ? (pair_end - at) / sp.chunk[1] : 0;
    sp.slots[2] = 0;   // the last phase runs to the end
    // scan_late_refill: the share of the groups (per mille) from which on a wave is handed one
    // group at a time, on demand (0: 985; -1: never)
    sp.late_from = 0x7fffffffll;
    const int late_permille = tuning().scan_late_refill > 0 ? tuning().scan_late_refill : tuning().scan_late_refill < 0 ? 0 : 985;
    if (late_permille > 0) {
        const long long late_group = (long long)(late_permille * 1e-3 * (double)sp.ngroups);
        // slots before late_group, with the phases' chunk sizes
        long long slot = 0, first = sp.dyn_first;
        for (int p = 0; p < kStreamPhases; ++p) {
            const long long n_p = p < kStreamPhases - 1 ? sp.slots[p] : sp.ngroups;
            const long long last = first + n_p * sp.chunk[p];
            if (late_group < last) { slot += late_group > first ? (late_group - first) / sp.chunk[p] : 0; break; }
            slot += n_p;
            first = last;
        }
        sp.late_from = slot;
    }
    sp.sync = sync;
    return sp;
}

template <typename T, bool NT, bool ARITH>
hipError_t launch_stream(const ScanArgs& a, const ScanPlan& plan, const StreamSplit& sp, hipStream_t stream) {
    const size_t lds = ARITH ? plan.lds : stream_lds_bytes(plan.qstride, 1);
#define MTR_STREAM_CASE(S)                                                                                             \
    case S: {                                                                                                          \
        if (lds > 48 * 1024) {                                                                                         \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&scan_stream_kernel<T, NT, S, ARITH>),    \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                  \
            if (e != hipSuccess) return e;                                                                             \
        }                                                                                                              \
        hipLaunchKernelGGL((scan_stream_kernel<T, NT, S, ARITH>), dim3(plan.grid), dim3(kStreamThreads), lds,         \
                           stream, a, sp);                                                                             \
        break;                                                                                                         \
    }
    switch (plan.stream_sweeps) {
        MTR_STREAM_CASE(2)
        MTR_STREAM_CASE(3)
        MTR_STREAM_CASE(4)
        default: return hipErrorInvalidValue;
    }
#undef MTR_STREAM_CASE
    return hipGetLastError();
}

size_t scan_lds_bytes(int qb, int qstride, int k) {
    const int kWaves = waves_per_group(qb);
    return (size_t)qb * qstride * 4 + (size_t)kWaves * qb * k * 8 + (size_t)kWaves * qb * 4;
}

template <typename T, int QB, bool NT>
hipError_t launch_one(const ScanArgs& a, const ScanPlan& plan, hipStream_t stream) {
    if (plan.lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&scan_topk_kernel<T, QB, NT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)plan.lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((scan_topk_kernel<T, QB, NT>), dim3(plan.grid), dim3(waves_per_group(QB) * 64), plan.lds, stream, a);
    return hipGetLastError();
}

template <typename T, int QB, bool NT>
int occupancy_of(size_t lds) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, scan_topk_kernel<T, QB, NT>, waves_per_group(QB) * 64, lds) != hipSuccess)
        nb = 1;
    return nb < 1 ? 1 : nb;
}

template <typename T, bool NT>
int occupancy_qb(int qb, size_t lds) {
    switch (qb) {
        case 1: return occupancy_of<T, 1, NT>(lds);
        case 2: return occupancy_of<T, 2, NT>(lds);
        case 4: return occupancy_of<T, 4, NT>(lds);
        default: return occupancy_of<T, 8, NT>(lds);
    }
}

template <typename T, bool NT>
hipError_t launch_qb(const ScanArgs& a, const ScanPlan& plan, hipStream_t stream) {
    switch (plan.qb) {
        case 1: return launch_one<T, 1, NT>(a, plan, stream);
        case 2: return launch_one<T, 2, NT>(a, plan, stream);
        case 4: return launch_one<T, 4, NT>(a, plan, stream);
        case 8: return launch_one<T, 8, NT>(a, plan, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace

int plan_scan(int dtype, int64_t n, int d, int ld, int b, int k, int num_cus, ScanPlan* plan) {
    const int esize = dtype == 0 ? 4 : 2;
    const int epc = 16 / esize;
    if ((ld * esize) % 16 != 0 || ld < d || k < 1 || n < 1 || b < 1) return -1;
    const int nchunks = (d * esize + 15) / 16;
    plan->nit = (nchunks + kLanesPerRow - 1) / kLanesPerRow;
    plan->qstride = plan->nit * kLanesPerRow * epc;
    int qb = 1;
    // Queries per pass: four on bf16 shards (which hand batches from two on to the matrix cores
    // anyway), eight on fp32 shards, where a pass of eight (1M x 768: 0.57 ms) still beats the
    // fp32 matrix instruction's 32-query tile (0.59 ms).
    const int max_qb = tuning().scan_max_qb > 0 ? tuning().scan_max_qb : (dtype == 0 ? 8 : 4);
    while (qb < b && qb < max_qb && qb < 8) qb *= 2;
    constexpr size_t kLdsBudget = 150 * 1024;  // of the CU's 160 KiB
    while (qb > 1 && scan_lds_bytes(qb, plan->qstride, k) > kLdsBudget) qb /= 2;
    if (scan_lds_bytes(qb, plan->qstride, k) > kLdsBudget) return -1;
    plan->qb = qb;
    plan->lds = scan_lds_bytes(qb, plan->qstride, k);
    // Rows larger than the Infinity Cache gain nothing from being cached.
    plan->nontemporal = ((double)n * ld * esize > 192.0 * 1024 * 1024) ? 1 : 0;
    if (tuning().scan_nontemporal >= 0) plan->nontemporal = tuning().scan_nontemporal != 0;
    int occ;
    if (dtype == 0)
        occ = plan->nontemporal ? occupancy_qb<float, true>(qb, plan->lds) : occupancy_qb<float, false>(qb, plan->lds);
    else
        occ = plan->nontemporal ? occupancy_qb<bf16_t, true>(qb, plan->lds) : occupancy_qb<bf16_t, false>(qb, plan->lds);
    const long long ngroups = (n + kGroupRows - 1) / kGroupRows;
    const int kWaves = waves_per_group(qb);
    long long want = (ngroups + kWaves - 1) / kWaves;
    // Single query: one workgroup per CU measured best at every shard size (10M x
    // 1024 bf16: 2.914 ms against 2.93-2.98 ms with 2-8 per CU; 1M x 768 fp32:
    // 0.457 against 0.464 ms with two).  Several queries per pass need the
    // occupancy to hide their longer dependent chains.
    long long cap = (long long)num_cus * (qb == 1 ? 1 : occ);
    if (tuning().scan_wg_per_cu > 0) cap = (long long)num_cus * tuning().scan_wg_per_cu;
    plan->grid = (int)(want < cap ? want : cap);
    if (plan->grid < 1) plan->grid = 1;
    // One query per pass on a shard beyond the caches: the streaming form, one
    // workgroup of four waves per CU, when a row's sweeps split into whole units
    // and every wave has groups to take by index (tuning: scan_form 1 keeps
    // the classic form, 2 takes the streaming form whenever the shape allows it).
    plan->stream_sweeps = 0;
    const int sweeps = stream_sweeps(plan->nit);
    const int sgrid = num_cus * (tuning().scan_wg_per_cu > 0 ? tuning().scan_wg_per_cu : 1);
    const long long sgroups = (n + kStreamGroupRows - 1) / kStreamGroupRows;
    const bool can_stream = qb == 1 && sweeps != 0 && sgroups >= 2ll * sgrid * kStreamWaves &&
                            stream_lds_bytes(plan->qstride, k) <= kLdsBudget;
    const bool want_stream = tuning().scan_form == 2 || (tuning().scan_form != 1 && plan->nontemporal &&
                                                         sgroups >= 16ll * sgrid * kStreamWaves);
    if (can_stream && want_stream) {
        plan->stream_sweeps = sweeps;
        plan->grid = sgrid;
        plan->lds = stream_lds_bytes(plan->qstride, k);
    }
    return 0;
}

hipError_t launch_scan_read_only(const void* rows, int dtype, int64_t n, int ld, int d, const ScanPlan& plan,
                                 unsigned int* sink, unsigned int* sync, hipStream_t stream) {
    const int esize = dtype == 0 ? 4 : 2;
    const int nchunks = (d * esize + 15) / 16;
    if (plan.stream_sweeps) {
        if (!sync) return hipErrorInvalidValue;
        ScanArgs a = {};
        a.rows = static_cast<const char*>(rows);
        a.n = n;
        a.row_bytes = (long long)ld * esize;
        a.nchunks = nchunks;
        a.nit = plan.nit;
        a.qstride = plan.qstride;
        a.k = 1;
        a.cand_i = reinterpret_cast<int*>(sink);
        const StreamSplit sp = stream_split(n, plan.grid, sync);
        return plan.nontemporal ? launch_stream<bf16_t, true, false>(a, plan, sp, stream)
                                : launch_stream<bf16_t, false, false>(a, plan, sp, stream);
    }
    if (plan.nontemporal)
        hipLaunchKernelGGL(scan_read_only_kernel<true>, dim3(plan.grid), dim3(waves_per_group(1) * 64), 0, stream,
                           static_cast<const char*>(rows), (long long)n, (long long)ld * esize, nchunks, plan.nit, sink);
    else
        hipLaunchKernelGGL(scan_read_only_kernel<false>, dim3(plan.grid), dim3(waves_per_group(1) * 64), 0, stream,
                           static_cast<const char*>(rows), (long long)n, (long long)ld * esize, nchunks, plan.nit, sink);
    return hipGetLastError();
}

hipError_t launch_scan_topk(const void* rows, int dtype, int64_t n, int ld, int d, const float* q,
                            const float* q_raw, int nb, int normalize, NonfiniteFlag nonfinite_flag, int k,
                            const ScanPlan& plan, float* cand_s, int* cand_i, unsigned int* sync, hipStream_t stream,
                            const int* run_if) {
    const int esize = dtype == 0 ? 4 : 2;
    ScanArgs a;
    a.rows = static_cast<const char*>(rows);
    a.n = n;
    a.row_bytes = (long long)ld * esize;
    a.nchunks = (d * esize + 15) / 16;
    a.nit = plan.nit;
    a.q = q;
    a.q_raw = q_raw;
    a.d = d;
    a.nb = nb;
    a.normalize = normalize;
    a.nonfinite_flag = nonfinite_flag;
    a.qstride = plan.qstride;
    a.k = k;
    a.cand_s = cand_s;
    a.cand_i = cand_i;
    a.run_if = run_if;
    if (plan.stream_sweeps) {
        if (!sync || plan.qb != 1) return hipErrorInvalidValue;
        const StreamSplit sp = stream_split(n, plan.grid, sync);
        if (dtype == 0)
            return plan.nontemporal ? launch_stream<float, true, true>(a, plan, sp, stream)
                                    : launch_stream<float, false, true>(a, plan, sp, stream);
        return plan.nontemporal ? launch_stream<bf16_t, true, true>(a, plan, sp, stream)
                                : launch_stream<bf16_t, false, true>(a, plan, sp, stream);
    }
    if (dtype == 0)
        return plan.nontemporal ? launch_qb<float, true>(a, plan, stream) : launch_qb<float, false>(a, plan, stream);
    return plan.nontemporal ? launch_qb<bf16_t, true>(a, plan, stream) : launch_qb<bf16_t, false>(a, plan, stream);
}

}  // namespace mtr
