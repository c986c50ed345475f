// Batched scoring on the matrix cores with the top-k selection fused in: a
// tiled  rows x queries  contraction whose score tiles never leave the
// registers -- v_mfma_f32_32x32x16_bf16 for bf16 shards, v_mfma_f32_32x32x2_f32
// (exact fp32 products, blocked summation) for fp32 shards.
//
// This is the many-queries form of retrieve_top_index
// (pipeline/utils/local_rag.py:76-78): all unique keyword vectors of a run
// (pipeline/async_main.py:193-195) scored in one pass over the page matrix.
//
// Work split.  Queries are cut into tiles of 32, 64 or 128 (by batch size), the
// stored rows into ranges; one workgroup (4 waves) owns one (query tile, row
// range) pair and sweeps the range in tiles of 128 rows.  Workgroups that share
// a query tile are placed on one XCD (consecutive ids after the XCD remap), so
// that tile stays in that XCD's L2 while every XCD streams the rows.
//
// Per tile.  K runs over the row width in steps of 128 bytes: both operands are
// K-contiguous (scores = rows . queries^T), so the row tile and the query tile
// are staged with 16-byte direct-to-LDS loads into two buffers (the next
// step's loads are issued before this step's MFMAs; one barrier per step).
// The LDS image is 128-byte rows whose 16-byte chunks are XOR-swizzled with
// the row number -- applied on the global source address, since a
// direct-to-LDS load writes lane-linear -- which makes the ds_read_b128
// fragment reads conflict-free.  Each wave computes its sub-tile as 32x32 MFMA
// tiles with the rows as the A operand: the accumulator then holds ONE query
// per lane (column = lane & 31) and 16 rows per register set, so the per-query
// threshold test is one compare per lane.
//
// Fused top-k.  Each query keeps in LDS the score of its current k-th best
// (the floor) and a count; scores above the floor are appended, as packed
// 64-bit keys (wave_select.h), to the workgroup's private candidate buffer in
// global memory (L2-resident).  When a query's buffer could overflow on the
// next tile, one wave re-selects its k best with a radix walk, which also
// raises the floor.  Rows arrive in increasing order, so "strictly above the
// floor" never drops a row that would win a tie.  A second, non-strict floor
// per query is shared by all row ranges: it starts from a sampling launch
// (MAX_ONLY below) on large shards and rises as ranges publish their own
// floors.  At the end every query's k best are written as one candidate list
// per row range; merge_topk.hip merges the ranges.
//
// Algorithmic work per launch: 2 * B * n * d flops (fp32 accumulate).
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "mtr_device.h"
#include "wave_select.h"

namespace mtr {

namespace {

// Row tiles are 128 rows; queries per tile are a template parameter TNQ, so
// that small batches do not pay for 128 columns.  Four waves per workgroup:
// TNQ 128: waves 2x2, each 64 rows x 64 queries = 2x2 MFMA tiles; 64: waves 2x2,
// each 64 x 32 = 2x1; 32: waves 4x1, each 32 x 32.
constexpr int kTileRows = 128;   // rows per tile (256 in the tall form of the 64-query tile)
constexpr int kThreads = 256;
constexpr int kStepBytes = 128;  // K bytes of a row per step: 64 bf16 or 32 fp32
constexpr int stage_bytes(int tm, int tnq) { return (tm + tnq) * kStepBytes; }  // row tile then query tile
constexpr int lds_bytes(int tm, int tnq, int stages) { return stages * stage_bytes(tm, tnq) + tnq * 12 + (tm == 256 ? 8 : 4) * 1024; }

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

using f32x4v = __attribute__((ext_vector_type(4))) float;

// Timing-experiment switches (results are then wrong by design) exist only in
// builds with -DMTR_TIMING_SWITCHES; the default build folds them to zero.
#ifdef MTR_TIMING_SWITCHES
#define MTR_TIMING(a) (a).debug_no_candidates
#else
#define MTR_TIMING(a) 0
#endif

struct MfmaArgs {
    const char* rows;        // [n][ld] elements of T
    long long n;
    int ld;                  // elements; ld * sizeof(T) is a multiple of kStepBytes
    const char* queries;     // [n_qtiles*TN][ld] elements of T, zero rows past b
    int b;
    int n_qtiles;
    int n_ranges;
    long long tiles_per_range;
    long long n_tiles;       // tiles this launch covers (numbered 0.. in its own order)
    int tile_stride;         // tile t of the launch is tile t * tile_stride of the matrix
    unsigned int* tile_max;  // not NULL: only record every query's best score per tile, [n_qtiles*TN][n_tiles]
    int k;
    int slack;               // extra keys an in-loop re-selection may keep
    int debug_no_candidates;       // timing experiments only: bit 0 every floor starts at +inf; bit 1 appends wrap, no re-selection; bit 2 count events
    unsigned long long* debug_counters;  // [0] appends [1] re-selections (bit 2)
    float debug_floor;                   // bit 4: every own floor starts here
    unsigned long long* cand_buf;  // [grid][TN][CAP]
    unsigned int* shared_floor;    // [n_qtiles*TN] order-preserving score bits, zeroed before the launch
    float* out_s;            // [b][n_ranges][k]
    int* out_i;
    const int* run_if;       // not NULL: do nothing while *run_if == 0
    int interleave;          // 1: range r takes tiles r, r + n_ranges, ... instead of a contiguous stretch (tile_stride 1 only)
    long long static_tiles;  // DYN forms: tiles every range takes by index (range r: r, r + n_ranges, ...); the rest is claimed
    unsigned int* sync;      // DYN forms: [0] claim counter, [16] workgroups whose claimer has finished; zero between launches
};

template <int AUX, int ROWS, int WAVES>
__device__ __forceinline__ void stage_tile(const char* base, long long row0, long long row_max, long long row_bytes,
                                           int k0_bytes, char* lds, int wave, int lane) {
    // ROWS rows x 128 bytes.  One wave instruction fills 8 rows; lane L writes
    // LDS position (row L/8, slot L%8) and fetches chunk (L%8) ^ (L/8) of it.
    // Wave w takes pieces w, w + WAVES, ...: an equal number each when the pieces
    // divide evenly (the forms with counted waits rely on that, or on "one piece
    // for the first waves only"), else the first waves take one more.
    constexpr int PIECES = ROWS / 8;
    constexpr int ROUNDS = (PIECES + WAVES - 1) / WAVES;
#pragma unroll
    for (int i = 0; i < ROUNDS; ++i) {
        const int t = PIECES % WAVES == 0 ? wave * ROUNDS + i : wave + i * WAVES;
        if (PIECES % WAVES != 0 && t >= PIECES) break;   // wave-uniform
        const int r = t * 8 + (lane >> 3);
        long long row = row0 + r;
        row = row < row_max ? row : row_max;
        const int chunk = (lane & 7) ^ (lane >> 3);
        const char* src = base + row * row_bytes + k0_bytes + chunk * 16;
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(lds + t * 1024), 16, 0, AUX);
    }
}

__device__ __forceinline__ bf16x8 read_frag(const char* lds, int row, int chunk) {
    return *reinterpret_cast<const bf16x8*>(lds + row * 128 + ((chunk ^ (row & 7)) << 4));
}
__device__ __forceinline__ f32x4v read_frag_f32(const char* lds, int row, int chunk) {
    return *reinterpret_cast<const f32x4v*>(lds + row * 128 + ((chunk ^ (row & 7)) << 4));
}

// One K step (128 bytes of every row of the stage) of a wave's 64x64 sub-tile.
// bf16: four v_mfma_f32_32x32x16_bf16 per 2x2 MFMA tile; lane half h supplies k
// elements 8h..8h+7 of each 16-element block.
// fp32: sixteen v_mfma_f32_32x32x2_f32 (exact fp32, an fmaf chain) per MFMA
// tile.  Its K is 2 per instruction, lane half h supplying one of them; the
// order of K inside a step is free as long as both operands agree, so half h
// takes floats 16h..16h+15 of the step, which are four 16-byte reads.
// row_off / q_off: first row / query of the wave's sub-tile inside the tile;
// MT x NT MFMA tiles of 32x32 per wave.
template <int MT, int NT>
__device__ __forceinline__ void mfma_step(bf16_t, const char* sa, const char* sb, int row_off, int q_off, int lane,
                                          f32x16 (&acc)[MT][NT]) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const int chunk = 2 * kk + (lane >> 5);
        bf16x8 fa[MT], fb[NT];
#pragma unroll
        for (int m = 0; m < MT; ++m) fa[m] = read_frag(sa, row_off + m * 32 + (lane & 31), chunk);
#pragma unroll
        for (int n = 0; n < NT; ++n) fb[n] = read_frag(sb, q_off + n * 32 + (lane & 31), chunk);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[m], fb[n], acc[m][n], 0, 0, 0);
    }
}

template <int MT, int NT>
__device__ __forceinline__ void mfma_step(float, const char* sa, const char* sb, int row_off, int q_off, int lane,
                                          f32x16 (&acc)[MT][NT]) {
    // Each step's 32 products are summed in a fresh accumulator and then added
    // to the running one: an fp32 MFMA is a plain fmaf chain, and one chain over
    // all of d (4096 terms) drifts past 1e-6 for near-duplicate vectors; blocks
    // of 32 keep the error near 1e-7.
    f32x4v fb[NT][4];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int c = 0; c < 4; ++c) fb[n][c] = read_frag_f32(sb, q_off + n * 32 + (lane & 31), 4 * (lane >> 5) + c);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        f32x4v fa[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) fa[c] = read_frag_f32(sa, row_off + m * 32 + (lane & 31), 4 * (lane >> 5) + c);
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            f32x16 part;
#pragma unroll
            for (int r = 0; r < 16; ++r) part[r] = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    part = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][j], fb[n][c][j], part, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] += part[r];
        }
    }
}

// One K step of a wave's sub-tile on v_mfma_f32_16x16x32_bf16: MT16 x NT16 tiles of 16 x 16,
// the step's 64 elements as two blocks of 32 (lane l supplies row or query l % 16 and elements
// 8 (l / 16) .. + 7 of a block; it ends up with rows 4 (l / 16) + r, r = 0..3, of query l % 16).
// The same instruction, operand layout and order of blocks as the 256-row filter kernel
// (score_wide.hip), so the two kernels' scores are the same bits -- the floors of this form's
// sampling launch come from that kernel.  The chip holds a higher clock under this
// instruction than under 32 x 32 x 16 (the same matrix-core cycles and LDS reads).
template <int MT16, int NT16>
__device__ __forceinline__ void mfma_step16b(const char* sa, const char* sb, int row_off, int q_off, int lane,
                                             f32x4v (&acc)[MT16][NT16]) {
    const int g = lane >> 4, c16 = lane & 15;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        bf16x8 fb[NT16];
#pragma unroll
        for (int n = 0; n < NT16; ++n) fb[n] = read_frag(sb, q_off + n * 16 + c16, 4 * kb + g);
#pragma unroll
        for (int m = 0; m < MT16; ++m) {
            const bf16x8 fa = read_frag(sa, row_off + m * 16 + c16, 4 * kb + g);
#pragma unroll
            for (int n = 0; n < NT16; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb[n], acc[m][n], 0, 0, 0);
        }
    }
}

// The 16- and 48-query tiles of fp32 shards (batches up to 16, and of 33..48):
// v_mfma_f32_16x16x4_f32, so that a 32- or 64-column tile is not half or a quarter
// padding -- the fp32 matrix rate, not HBM, is what bounds these batches.  A wave
// owns 32 rows x 16 NQ queries as 2 x NQ tiles of 16 x 16.  Lane l supplies row (or query) l % 16 and, for
// the j-th instruction of chunk c, K element 8 (l / 16) + 4 c + j of the step (the
// order of K inside a step is free as long as both operands agree); it ends up with
// rows 4 (l / 16) + r, r = 0..3, of query l % 16.  Blocked summation as above.
template <int NQ>
__device__ __forceinline__ void mfma_step16(const char* sa, const char* sb, int row_off, int lane, f32x4v (&acc)[2][NQ]) {
    const int g = lane >> 4, c16 = lane & 15;
    f32x4v fb[NQ][2];
#pragma unroll
    for (int n = 0; n < NQ; ++n)
#pragma unroll
        for (int c = 0; c < 2; ++c) fb[n][c] = read_frag_f32(sb, n * 16 + c16, 2 * g + c);
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        f32x4v fa[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) fa[c] = read_frag_f32(sa, row_off + m * 16 + c16, 2 * g + c);
#pragma unroll
        for (int n = 0; n < NQ; ++n) {
            f32x4v part = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int j = 0; j < 4; ++j) part = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[c][j], fb[n][c][j], part, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[m][n][r] += part[r];
        }
    }
}

// Re-selection of one query's buffer.
//   load_keys:      c entries (c <= 64*E) into registers, 0 for empty slots.
//   finish_select:  keeps between k and k + slack of them, compacted to the
//                   front, and publishes the new count and floors.
template <int E>
__device__ __forceinline__ void load_keys(const unsigned long long* buf, int c, int lane,
                                          unsigned long long (&key)[E]) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = e * 64 + lane;
        key[e] = i < c ? buf[i] : 0ull;
    }
}

template <int E>
__device__ __forceinline__ void finish_select(unsigned long long* buf, const unsigned long long (&key)[E], int k,
                                              int slack, float* floor_lds, int* cnt_lds, int q, int lane,
                                              unsigned int* shared_floor_q, unsigned int* hist) {
    int passes = 0;
    const unsigned long long cut = wave_kth_largest_hist<E>(key, k, slack, hist, lane, &passes);
    int base = 0;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const bool keep = key[e] != 0ull && key[e] >= cut;
        const unsigned long long m = __ballot(keep);
        const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
        if (keep) buf[pos] = key[e];
        base += __popcll(m);
    }
    if (lane == 0) {
        cnt_lds[q] = base;
        if (cut != 0ull) {
            const unsigned int hi = (unsigned int)(cut >> 32);
            // Own floor (tested with ">"): the exact k-th best score once all four
            // score bytes are decided, else the largest value below the kept bin.
            floor_lds[q] = unorder_f32(passes >= 4 ? hi : hi - 1u);
            // At least k rows of this range score >= hi, so no row below it can be in
            // the query's overall top-k: tell the other row ranges.
            atomicMax(shared_floor_q, hi);
        }
    }
}

template <int E>
__device__ __forceinline__ void compact_query(unsigned long long* buf, int c, int k, int slack, float* floor_lds,
                                              int* cnt_lds, int q, int lane, unsigned int* shared_floor_q,
                                              unsigned int* hist) {
    unsigned long long key[E];
    load_keys<E>(buf, c, lane, key);
    finish_select<E>(buf, key, k, slack, floor_lds, cnt_lds, q, lane, shared_floor_q, hist);
}

// ---- pieces of the fused top-k shared by the tile shapes ---------------------

// LDS stores and the LDS atomic of an append, written as assembly for the kernel form that
// keeps rows in flight in registers.  The compiler orders every LDS store or atomic it can
// see behind ALL direct-to-LDS loads the wave has in flight (s_waitcnt vmcnt(0) in front of
// it: for all it knows the two write the same bytes), and with it behind every register
// load issued before them -- which would drain that form's queue of row loads at every
// step.  The stage buffers, the counts and the floors are different bytes; an instruction
// the compiler cannot look into gets no such wait.  (LDS operations of one wave complete in
// order, and every reader of another wave sits behind `s_waitcnt lgkmcnt(0)` + s_barrier.)
__device__ __forceinline__ unsigned int lds_address(const void* p) { return (unsigned int)(size_t)(lds_ptr_t)p; }
template <int OFFSET>
__device__ __forceinline__ void lds_store_b128(unsigned int addr, const u32x4& v) {
    asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFFSET) : "memory");
}
// A streaming 16-byte load the compiler does not track: it returns at once as far as the
// compiler knows, so the CALLER must not let anything read `dst` before a counted
// s_waitcnt of its own has covered the load (the compiler's own waits for loads it does
// track came out as vmcnt(0) inside the loop that keeps four steps in flight).
// (Address: a wave-uniform base in scalar registers plus a 32-bit lane offset, which costs
// one vector register per load where a 64-bit lane address costs two and an add.)
// A wave-uniform pointer as one the compiler must keep in scalar registers.
__device__ __forceinline__ const char* uniform_pointer(const char* p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)v);
    const unsigned int hi = __builtin_amdgcn_readfirstlane((unsigned int)(v >> 32));
    return reinterpret_cast<const char*>(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ void load_b128_untracked(u32x4& dst, const char* base, unsigned int lane_offset) {
    asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(dst) : "v"(lane_offset), "s"(base) : "memory");
}
// "score > own floor && score >= shared floor" as one strict threshold, from the shared
// floor word as published (order-preserving score bits, 0: none yet): v >= sh is
// v > (largest float below sh); with no floor the bits stand for a NaN, which loses the fmaxf.
__device__ __forceinline__ float strict_threshold(float own, unsigned int known) {
    const unsigned int ord = known | (known == 0u ? 0x007fffffu : 0u);
    return fmaxf(own, unorder_f32(ord - 1u));
}
// The shared floors of up to 64 queries, one per active lane, from global memory (read at
// the device's coherence point: other workgroups raise them with atomics) straight into
// the LDS words dst[lane]: no register result, so nothing the compiler would wait for.
// Whatever a reader finds there, old or new, is a floor some row range has published.
__device__ __forceinline__ void fetch_floors(const unsigned int* src, unsigned int* dst) {
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)dst, 4, 0, 16 /* sc1: device scope */);
}
__device__ __forceinline__ int lds_add_return(int* p, int v) {
    int old;
    asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(old) : "v"(lds_address(p)), "v"(v) : "memory");
    return old;
}

// One 32 x 32 MFMA tile of finished scores (one query per lane column, 16 rows
// per lane: row0 + (r & 3) + 8 * (r >> 2) for register r): appends what is
// above the query's threshold to its candidate buffer and clears the tile.
// The wave does not wait for its stores: whoever reads the keys (a re-selection,
// the final selection) has every wave wait in front of a barrier first.
template <int CAP, bool ASM_ATOMIC = false>
__device__ __forceinline__ void append_above(f32x16& tile, float thr, int ql, long long row0, bool last_tile,
                                             const MfmaArgs& a, int* cnt_lds, unsigned long long* my_buf) {
    // registers 4 g .. 4 g + 3 hold rows row0 + 8 g .. + 3
    float gmax[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) gmax[g] = fmaxf(fmaxf(tile[4 * g], tile[4 * g + 1]), fmaxf(tile[4 * g + 2], tile[4 * g + 3]));
    const float mx = fmaxf(fmaxf(gmax[0], gmax[1]), fmaxf(gmax[2], gmax[3]));
    if (__any(mx > thr)) {
        // As a rule ONE lane holds ONE such score.  It is found behind wave-uniform tests
        // (a compare into the condition mask and a scalar branch each: four groups, then the
        // four registers of a group that holds one) and takes its slot with an LDS atomic of
        // its own.  (Letting every lane walk its own passing rows instead, with a select tree
        // to fetch element r, measured 5-15 % slower at every batch size; counting the lane's
        // scores first, to reserve their slots at once, costs more than the atomics it saves.)
        // Only the last tile of the matrix can hold rows past its end.
        const long long left = a.n - row0;
        const int rows_left = !last_tile || left > 32 ? 32 : left < 0 ? 0 : (int)left;
        unsigned long long* dst = my_buf + (size_t)ql * CAP;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (__any(gmax[g] > thr)) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool hit = tile[4 * g + j] > thr && 8 * g + j < rows_left;
                    if (__any(hit)) {
                        if (hit) {
                            int slot = ASM_ATOMIC ? lds_add_return(&cnt_lds[ql], 1) : atomicAdd(&cnt_lds[ql], 1);
                            if (MTR_TIMING(a) & 4) atomicAdd(a.debug_counters, 1ull);
                            if (MTR_TIMING(a) & 2) slot &= CAP - 1;
                            if (slot < CAP && !(MTR_TIMING(a) & 8)) dst[slot] = pack_key(tile[4 * g + j], (int)(row0 + 8 * g + j));
                        }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) tile[r] = 0.f;
}

// The same for one 16 x 16 tile of the 16-query form: query lane % 16, rows row0 + r.
template <int CAP, bool ASM_ATOMIC = false>
__device__ __forceinline__ void append_above16(f32x4v& tile, float thr, int ql, long long row0, bool last_tile,
                                               const MfmaArgs& a, int* cnt_lds, unsigned long long* my_buf) {
    const float mx = fmaxf(fmaxf(tile[0], tile[1]), fmaxf(tile[2], tile[3]));
    if (__any(mx > thr)) {
        unsigned int pass = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) pass |= (tile[r] > thr) ? (1u << r) : 0u;
        if (last_tile) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (row0 + r >= a.n) pass &= ~(1u << r);
        }
        int slot = 0;
        if (pass) slot = ASM_ATOMIC ? lds_add_return(&cnt_lds[ql], __popc(pass)) : atomicAdd(&cnt_lds[ql], __popc(pass));
        unsigned int any_r = pass;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) any_r |= __shfl_xor(any_r, off, 64);
        any_r = __builtin_amdgcn_readfirstlane(any_r);
        unsigned long long* dst = my_buf + (size_t)ql * CAP;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (any_r & (1u << r)) {
                if (pass & (1u << r)) {
                    if (MTR_TIMING(a) & 2) slot &= CAP - 1;
                    if (slot < CAP) dst[slot] = pack_key(tile[r], (int)(row0 + r));
                    ++slot;
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) tile[r] = 0.f;
}

// s_barrier without the drain of outstanding loads that __syncthreads() adds.
__device__ __forceinline__ void barrier_keeping_loads() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// The calling wave looks after queries [q0, q0 + nq), nq <= 64: re-selects any
// whose buffer could overflow on the next tile (a tile adds at most tile_rows).
template <int CAP>
__device__ __forceinline__ void reselect_where_full(int q0, int nq, int tile_rows, int lane, const MfmaArgs& a,
                                                    float* floor_lds, int* cnt_lds, unsigned long long* my_buf,
                                                    unsigned int* shared_floor, unsigned int* hist_lds) {
    constexpr int E = CAP / 64;
    const int c = cnt_lds[q0 + (lane % nq)];
    unsigned long long need = __ballot(lane < nq && c + tile_rows > CAP);  // k + slack + tile_rows <= CAP by plan
    if (MTR_TIMING(a) & 2) need = 0;
    if ((MTR_TIMING(a) & 4) && lane == 0 && need) atomicAdd(a.debug_counters + 1, (unsigned long long)__popcll(need));
    while (need) {
        const int j = __ffsll((long long)need) - 1;
        need &= need - 1;
        const int q = q0 + j;
        const int cq = __builtin_amdgcn_readfirstlane(cnt_lds[q]);
        compact_query<E>(my_buf + (size_t)q * CAP, cq, a.k, a.slack, floor_lds, cnt_lds, q, lane, shared_floor + q, hist_lds);
    }
}

// True (in every wave alike) when some query's buffer could overflow on the next
// tile.  Called between the barrier behind a tile's appends and the next barrier,
// while nobody changes the counts.
template <int CAP, int TN>
__device__ __forceinline__ bool any_buffer_full(const int* cnt_lds, int tile_rows, int lane) {
    bool full = false;
#pragma unroll
    for (int i = 0; i < (TN + 63) / 64; ++i) {
        const int q = i * 64 + lane;
        full |= q < TN && cnt_lds[q] + tile_rows > CAP;
    }
    return __any(full);
}

// Final selection of queries [q0, q0 + nq) of query tile base qt_base and one
// candidate list per (query, range).
template <int CAP>
__device__ __forceinline__ void write_range_lists(int q0, int nq, int qt_base, int range, int lane, const MfmaArgs& a,
                                                  float* floor_lds, int* cnt_lds, unsigned long long* my_buf,
                                                  unsigned int* shared_floor, unsigned int* hist_lds) {
    constexpr int E = CAP / 64;
    for (int j = 0; j < nq; ++j) {
        const int q = q0 + j;
        const int qg = qt_base + q;
        if (qg >= a.b) continue;  // wave-uniform
        int cq = __builtin_amdgcn_readfirstlane(cnt_lds[q]);
        unsigned long long* buf = my_buf + (size_t)q * CAP;
        if ((MTR_TIMING(a) & 128) && cq > a.k) cq = a.k;
        if (cq > a.k) {
            compact_query<E>(buf, cq, a.k, 0, floor_lds, cnt_lds, q, lane, shared_floor + q, hist_lds);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            cq = __builtin_amdgcn_readfirstlane(cnt_lds[q]);
        }
        float* os = a.out_s + ((size_t)qg * a.n_ranges + range) * a.k;
        int* oi = a.out_i + ((size_t)qg * a.n_ranges + range) * a.k;
        for (int i = lane; i < a.k; i += 64) {
            if (i < cq) {
                const unsigned long long key = buf[i];
                os[i] = key_score(key);
                oi[i] = key_row(key);
            } else {
                os[i] = -__builtin_huge_valf();
                oi[i] = -1;
            }
        }
    }
}

// MAX_ONLY: the sampling form, which keeps nothing but each query's best score
// per tile (a.tile_max).  A compile-time switch: as a run-time branch it made
// the compiler copy all accumulators after every K step.
// ROW_AUX: cache-policy bits of the row-tile loads (2 = nt, streaming).
// NS: LDS stage buffers.  2: the next step is requested before this step's
// MFMAs and waited for at the step's barrier (which drains every load).  3: a
// ring -- step s + 2 is requested as soon as the barrier of step s has shown
// that everybody is done with step s - 1, waits are counted (loads return in
// order, so "at most N outstanding" means all but the newest N have landed) and
// the barriers leave loads in flight, so the memory pipeline never runs empty.
// TM: rows per tile.  256 (64-query tiles in the ring form only): eight waves,
// each 32 rows x 64 queries, one workgroup per CU; the query tile's re-staging
// then adds a quarter to the row bytes that go into LDS instead of a half.
// DYN (ring form, one query tile, 256-row tiles): the ranges take most of the tiles
// by index, side by side (range r: tiles r, r + R, ...), and the last ones are
// claimed tile by tile from a device-wide counter, so that CUs the memory system
// serves faster end up with more rows (with an even split the median workgroup
// is done 40-130 us before the last one of a 2.9 ms launch).  A ninth wave makes
// the claims -- an atomic issued by a loading wave would sit in front of its
// counted waits -- and hands the tile numbers on through LDS; it takes part in
// every barrier of the eight others and does nothing else.
// REGS (tall ring form, bf16, 64- and 128-query tiles, rows of whole 512 bytes): the rows
// travel through registers.  With direct-to-LDS loads the bytes in flight are the LDS
// ring, two K steps of rows (64 KB a CU), and at 128 queries a step's MFMAs and fragment
// reads are long enough that two steps no longer cover a load's way through the memory
// system (10M x 1024: 3.75 ms with, 2.98 ms without the MFMAs).  Here every wave keeps
// kRegDepth K steps of its own row pieces in flight in registers (16 registers a step;
// 128 KB a CU at depth four), writes step s + 1 into one of TWO row stages right behind the
// barrier of step s and asks for step s + 1 + depth in the same breath; only the query
// tile still comes by direct-to-LDS loads, into a ring of depth + 1 stages and as far
// ahead as the rows, because a wave's loads complete in order: waiting for a young query
// piece would wait for every older row load too.  Past its last tile a workgroup keeps
// requesting (from the query matrix, which is in L2) so that no load sits behind a branch.
// What is left between this form and the scan's time at 65..128 queries is the power limit:
// with random queries the chip holds 1.8 GHz under it where the scan runs at 2.4 GHz, and the
// same launch with all-zero queries takes 3.05 instead of 3.45-3.67 ms.
constexpr int kRegDepth = 4;
constexpr int lds_bytes_regs(int tnq) { return 2 * 256 * kStepBytes + (kRegDepth + 1) * tnq * kStepBytes + tnq * 12 + 8 * 1024 + tnq * 4; }

// M16 (with REGS): the sub-tiles as 16 x 16 MFMA tiles (mfma_step16b): 4 % faster at 128 queries,
// and the filter kernel's arithmetic, so its sampling form must provide the floors.
template <typename T, int CAP, int TNQ, bool MAX_ONLY, int ROW_AUX, int NS, int TM = kTileRows, bool DYN = false, bool REGS = false,
          bool M16 = false>
__global__ __launch_bounds__(DYN ? 576 : TM == 256 ? 512 : 256, TM == 256 ? 1 : 2) void score_topk_mfma_kernel(MfmaArgs a) {
    static_assert(!REGS || (NS == 3 && TM == 256 && !MAX_ONLY && !DYN && sizeof(T) == 2 && TNQ % 64 == 0 && ROW_AUX == 2),
                  "rows through registers: tall ring form, bf16, 64 or 128 queries, streaming loads");
    static_assert(!M16 || REGS, "the 16 x 16 bf16 tiles exist in the rows-through-registers form only");
    static_assert(!DYN || (NS == 3 && TM == 256 && !MAX_ONLY), "claimed tiles exist in the tall ring form only");
    static_assert(NS == 2 || (NS == 3 && !MAX_ONLY), "the ring form exists for the full launch only");
    static_assert(TM == 128 || (TM == 256 && NS == 3), "the tall tile exists in the ring form only");
    static_assert(NS == 2 || TNQ % (8 * (TM == 256 ? 8 : 4)) == 0 || TNQ / 8 < (TM == 256 ? 8 : 4),
                  "counted waits: the query tile's 8-row pieces are an equal number per wave, or one each for the first waves");
    constexpr bool N16 = TNQ % 32 != 0 || TNQ == 96;   // the forms on 16 x 16 tiles: 16, 48, 80, 96 and 112 queries (see mfma_step16)
    constexpr int NQ16 = N16 ? TNQ / 16 : 1;
    static_assert(!N16 || (sizeof(T) == 4 && NS == 2 && TM == 128), "the 16 x 16 forms exist for fp32 shards, two buffers, 128-row tiles");
    constexpr int kThreads = TM == 256 ? 512 : 256;
    if (a.run_if && *a.run_if == 0) return;
    constexpr int TN = TNQ;
    constexpr int WAVES = kThreads / 64;
    constexpr int WN = (TNQ <= 32 || N16 || (TM == 256 && TNQ == 64)) ? 1 : 2;    // waves along the queries
    constexpr int WM = WAVES / WN;                          // ... and along the rows
    constexpr int MT = TM / (32 * WM);           // MFMA tiles per wave along the rows
    constexpr int NT = N16 ? 1 : TNQ / (32 * WN);   // ... and along the queries (the 16-query form has its own accumulators)
    constexpr int QPW = TNQ / WAVES;             // queries each wave looks after
    static_assert(CAP >= 2 * TM, "a tile may append TM keys per query on top of k + slack kept ones");
    constexpr int kStageBytes = stage_bytes(TM, TNQ);
    constexpr int BK = kStepBytes / (int)sizeof(T);
    static_assert((CAP & (CAP - 1)) == 0, "CAP must be a power of two");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const stage0 = smem;
    constexpr int kFrontBytes = REGS ? 2 * TM * kStepBytes + (kRegDepth + 1) * TNQ * kStepBytes : NS * kStageBytes;
    float* floor_lds = reinterpret_cast<float*>(smem + kFrontBytes);  // own k-th best score (strict)
    int* cnt_lds = reinterpret_cast<int*>(floor_lds + TN);
    float* shared_lds = reinterpret_cast<float*>(cnt_lds + TN);           // best floor of ANY range (non-strict)
    unsigned int* hist_lds = reinterpret_cast<unsigned int*>(shared_lds + TN) + (threadIdx.x >> 6) * 256;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: the counted waits branch on it
    const int row_off = (wave / WN) * (32 * MT);   // the wave's rows inside the tile
    const int q_off = (wave % WN) * (32 * NT);     // ... and its queries

    // XCD-aware, bijective workgroup remap: ids that land on one XCD become consecutive.
    const int nwg = gridDim.x;
    const int xcd = blockIdx.x % 8;
    const int per = nwg / 8, rem = nwg % 8;
    const int wgid = (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + blockIdx.x / 8;
    const int qt = wgid / a.n_ranges;
    const int range = wgid % a.n_ranges;

    const long long n_tiles = a.n_tiles;
    long long tile_lo = (long long)range * a.tiles_per_range;
    long long tile_hi = tile_lo + a.tiles_per_range;
    if (tile_hi > n_tiles) tile_hi = n_tiles;
    long long my_tiles = tile_hi > tile_lo ? tile_hi - tile_lo : 0;
    if (a.interleave) {   // the ranges sweep the matrix side by side: tiles range, range + n_ranges, ...
        tile_lo = 0;
        my_tiles = range < n_tiles ? (n_tiles - range + a.n_ranges - 1) / a.n_ranges : 0;
    }
    const int ksteps = a.ld / BK;
    // Tile t of this launch is tile phys(t) of the matrix (increasing in t).
    auto phys = [&](long long t) -> long long { return a.interleave ? t * a.n_ranges + range : t * a.tile_stride; };
    unsigned int* best_lds = reinterpret_cast<unsigned int*>(cnt_lds);  // max_only: best score of the tile per query
    const long long row_bytes = (long long)a.ld * (long long)sizeof(T);

    for (int i = tid; i < TN; i += kThreads) {
        const int qg = qt * TN + i;
        // padding queries collect nothing; the timing-only debug mode collects nothing at all
        floor_lds[i] = (qg < a.b && !(MTR_TIMING(a) & 1)) ? ((MTR_TIMING(a) & 16) ? a.debug_floor : -__builtin_huge_valf())
                                                                 : __builtin_huge_valf();
        cnt_lds[i] = 0;
        // floors known before the launch (from a sample of the rows) or already published by faster ranges
        const unsigned int known = __hip_atomic_load(a.shared_floor + (size_t)qt * TN + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        shared_lds[i] = unorder_f32(known | (known == 0 ? 0x007fffffu : 0u));
    }
    unsigned int* shared_floor = a.shared_floor + (size_t)qt * TN;
    unsigned int pending_floor = 0;  // threads < TN: the shared floor word fetched at the start of a tile
    unsigned long long* my_buf = a.cand_buf + (size_t)blockIdx.x * TN * CAP;
    const char* qbase = a.queries + (size_t)qt * TN * row_bytes;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    f32x4v acc16[2][NQ16];   // 16 x 16 forms: rows row_off + 16 m + 4 (lane / 16) + r, query 16 n + lane % 16
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < NQ16; ++n) acc16[m][n] = f32x4v{0.f, 0.f, 0.f, 0.f};

    const long long total_steps = my_tiles * ksteps;
    if constexpr (REGS) {
        constexpr int D = kRegDepth;               // K steps of row loads in flight per wave
        constexpr int QR = D + 1;                  // stages of the query ring
        constexpr int RP = TM / (8 * WAVES);       // 8-row pieces per wave and step (4)
        constexpr int QP = TNQ / (8 * WAVES);      // query pieces per wave and step (1 or 2)
        constexpr int kRowStage = TM * kStepBytes, kQStage = TNQ * kStepBytes;
        static_assert(RP * WAVES * 8 == TM && QP * WAVES * 8 == TNQ && D == 4, "equal shares, four register sets");
        if (my_tiles > 0) {   // workgroup-uniform
            char* const qstage = smem + 2 * kRowStage;
            // this wave's pieces of a row stage: piece i at +i KB, lane-linear as a direct-to-LDS load would write it
            const unsigned int row_dst = lds_address(smem) + (unsigned int)(wave * RP * 1024 + lane * 16);
            // the shared floors as fetched (raw words), behind the eight waves' histograms
            unsigned int* const fetched_lds = reinterpret_cast<unsigned int*>(shared_lds + TN) + WAVES * 256;
            for (int i = tid; i < TN; i += kThreads) fetched_lds[i] = order_f32(shared_lds[i]);
            const int chunk16 = ((lane & 7) ^ (lane >> 3)) * 16;
            u32x4 regs[D][RP];
            // M16: the wave's sub-tile as 16 x 16 tiles: rows row_off + 16 m + 4 (lane / 16) + r, query q_off + 16 n + lane % 16
            constexpr int MT16 = M16 ? 2 * MT : 1, NT16 = M16 ? 2 * NT : 1;
            f32x4v accb[MT16][NT16];
#pragma unroll
            for (int m = 0; m < MT16; ++m)
#pragma unroll
                for (int n = 0; n < NT16; ++n) accb[m][n] = f32x4v{0.f, 0.f, 0.f, 0.f};
            // Source of this wave's pieces of the tile being requested (the workgroup's ld_t-th): a scalar base
            // (the first of the wave's 32 rows) and this lane's offset from it for each piece.
            const char* src_base = nullptr;
            unsigned int src_off[RP];
            long long ld_t = 0;
            int ld_ks = 0;
            auto set_sources = [&](long long t) {
                const bool real = t < my_tiles;
                const long long pt = real ? phys(tile_lo + t) : 0;
                // rows past the end of the matrix are fetched as copies of the last row
                long long first = pt * TM + wave * (RP * 8);
                const long long over = first - (a.n - 1);
                if (over > 0) first = a.n - 1;
                const long long left = a.n - 1 - first;                      // rows of the matrix beyond the first
                const int max_r = left > RP * 8 - 1 ? RP * 8 - 1 : (int)left;
                // past the workgroup's last tile: rows of the query matrix (L2-resident), so that no load sits behind a branch
                src_base = real ? a.rows + first * row_bytes : qbase + (long long)((wave * (RP * 8)) % TN) * row_bytes;
#pragma unroll
                for (int i = 0; i < RP; ++i) {
                    int r = i * 8 + (lane >> 3);
                    if (real) r = r < max_r ? r : max_r;
                    src_off[i] = (unsigned int)r * (unsigned int)row_bytes + (unsigned int)chunk16;
                }
            };
            auto request_rows = [&](u32x4 (&set)[RP]) {
                const char* base = uniform_pointer(src_base + ld_ks * kStepBytes);
                // The base reaches its scalar registers through VALU instructions (v_readfirstlane), and a
                // memory instruction may read a scalar register a VALU instruction wrote only five wait
                // states later.  The compiler keeps that distance for instructions it knows; it does not
                // look into the assembly below, so the distance is made here (without it the first load of
                // a step went out with the previous step's base: a memory fault).
                asm volatile("s_nop 4" : "+s"(base));
#pragma unroll
                for (int i = 0; i < RP; ++i) load_b128_untracked(set[i], base, src_off[i]);
                if (++ld_ks == ksteps) {
                    ld_ks = 0;
                    set_sources(++ld_t);
                }
            };
            int q_ks = 0, q_stage = 0;   // request cursor of the query ring
            auto request_queries = [&]() {
                stage_tile<0, TNQ, WAVES>(qbase, 0, TN - 1, row_bytes, q_ks * kStepBytes, qstage + q_stage * kQStage, wave, lane);
                q_ks = q_ks + 1 == ksteps ? 0 : q_ks + 1;
                q_stage = q_stage + 1 == QR ? 0 : q_stage + 1;
            };
            int q_cur = 0;               // the stage the MFMAs read
            static_assert(RP == 4, "four pieces per wave and step");
            auto store_pieces = [&](auto stage_offset, const u32x4 (&set)[RP]) {
                constexpr int O = decltype(stage_offset)::value;
                lds_store_b128<O>(row_dst, set[0]);
                lds_store_b128<O + 1024>(row_dst, set[1]);
                lds_store_b128<O + 2048>(row_dst, set[2]);
                lds_store_b128<O + 3072>(row_dst, set[3]);
            };
            // One K step; J = step % 4 picks the register set and the row stage at compile time.
            auto step = [&](auto jc) {
                constexpr int J = decltype(jc)::value;
                constexpr int SET = (J + 1) % D;
                // the query pieces of this step have landed (this wave's share; the barrier extends it to
                // all), and so has every row load older than them: those of step s + 1 among them
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * (RP + QP)) : "memory");
                barrier_keeping_loads();
                // every wave is past the MFMAs of step s - 1: its row stage takes step s + 1 ...
                if (!(MTR_TIMING(a) & 64)) {
                    store_pieces(std::integral_constant<int, ((J + 1) & 1) * kRowStage>{}, regs[SET]);
                }
                // ... its registers step s + 1 + D, and the query stage of step s - 1 step s + D
                request_rows(regs[SET]);
                request_queries();
                if (!(MTR_TIMING(a) & 32)) {
                    if constexpr (M16) mfma_step16b<MT16, NT16>(smem + (J & 1) * kRowStage, qstage + q_cur * kQStage, row_off, q_off, lane, accb);
                    else mfma_step<MT, NT>(T{}, smem + (J & 1) * kRowStage, qstage + q_cur * kQStage, row_off, q_off, lane, acc);
                }
                q_cur = q_cur + 1 == QR ? 0 : q_cur + 1;
            };
            // Nothing but the loads below may be in flight when the counted waits start.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            set_sources(0);
            request_rows(regs[0]);
            request_rows(regs[1]);
            request_queries();
            request_rows(regs[2]);
            request_queries();
            request_rows(regs[3]);
            request_queries();
            // step "-1": the first step's rows go into stage 0
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * (RP + QP)) : "memory");
            store_pieces(std::integral_constant<int, 0>{}, regs[0]);
            request_rows(regs[0]);
            request_queries();
            for (long long t = 0; t < my_tiles; ++t) {
                const long long ptile = phys(tile_lo + t);
                // the floors the other ranges have published by now: one load more in this wave's queue,
                // so the counted waits of the next three steps insist on one row load of three steps ago
                if (tid < TN) fetch_floors(shared_floor + tid, fetched_lds + (wave << 6));
                int kg = 0;
                do {
                    step(std::integral_constant<int, 0>{});
                    step(std::integral_constant<int, 1>{});
                    step(std::integral_constant<int, 2>{});
                    step(std::integral_constant<int, 3>{});
                    kg += D;
                } while (kg < ksteps);
                const bool last_tile = (ptile + 1) * TM > a.n;
                if constexpr (M16) {
                    const long long row_base = ptile * TM + row_off + 4 * (lane >> 4);
#pragma unroll
                    for (int n = 0; n < NT16; ++n) {
                        const int ql = q_off + n * 16 + (lane & 15);
                        const float thr = strict_threshold(floor_lds[ql], fetched_lds[ql]);
#pragma unroll
                        for (int m = 0; m < MT16; ++m)
                            append_above16<CAP, true>(accb[m][n], thr, ql, row_base + m * 16, last_tile, a, cnt_lds, my_buf);
                    }
                } else {
                    const long long row_base = ptile * TM + row_off + 4 * (lane >> 5);
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        const int ql = q_off + n * 32 + (lane & 31);
                        const float thr = strict_threshold(floor_lds[ql], fetched_lds[ql]);
#pragma unroll
                        for (int m = 0; m < MT; ++m)
                            append_above<CAP, true>(acc[m][n], thr, ql, row_base + m * 32, last_tile, a, cnt_lds, my_buf);
                    }
                }
                barrier_keeping_loads();  // the counts of this tile are visible
                if (any_buffer_full<CAP, TN>(cnt_lds, TM, lane)) {
                    // only when a buffer has to be re-selected, the keys: every wave waits for its
                    // stores (and with them for everything it has in flight) in front of one more barrier
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    barrier_keeping_loads();
                    reselect_where_full<CAP>(wave * QPW, QPW, TM, lane, a, floor_lds, cnt_lds, my_buf, shared_floor, hist_lds);
                }
            }
            // the requests made past the last tile still land in registers the code below is free to reuse
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        write_range_lists<CAP>(wave * QPW, QPW, qt * TN, range, lane, a, floor_lds, cnt_lds, my_buf, shared_floor, hist_lds);
        return;
    }
    if constexpr (DYN) {
        constexpr int LPS_ROWS = TM / (8 * WAVES);
        constexpr bool kUnevenQ = TNQ < 8 * WAVES;
        constexpr int LPS = LPS_ROWS + (kUnevenQ ? 0 : TNQ / (8 * WAVES));
        // The workgroup's j-th tile: by index while j < S, else what the claimer put into
        // seq[j & 3] (-1: no such tile, nor any later one).  The waves use entries j and
        // j + 1 (the requests run two K steps ahead), the claimer writes entry j + 2.
        int* seq = reinterpret_cast<int*>(smem + NS * kStageBytes + TN * 12 + WAVES * 1024);
        const long long R = a.n_ranges, S = a.static_tiles;
        const long long dyn_first = S * R;
        if (wave == WAVES) {
            // ---- the claimer ----
            long long m0 = -1, m1 = -1, m2 = -1, m3 = -1;   // its copy of seq (scalars: an indexed array would live in scratch)
            bool ended = false;                              // a claim has come back past the end
            unsigned int opaque_zero;
            asm volatile("v_mov_b32 %0, 0" : "=v"(opaque_zero));
            for (long long j = 0;; ++j) {
                const int e = (int)(j & 3);
                const long long pj = j < S ? j * R + range : (e == 0 ? m0 : e == 1 ? m1 : e == 2 ? m2 : m3);
                if (pj < 0) break;
                const bool wanted = j + 2 >= S;
                const bool claims = wanted && !ended;
                // The claim is issued here and read behind the tile's K steps.  (Only lane 0's value
                // is read; the other lanes' stays undefined, and the address is one the compiler cannot
                // prove uniform: either would otherwise make it wait for the result on the spot.)
                unsigned int tick;
                if (claims && lane == 0)
                    tick = __hip_atomic_fetch_add(a.sync + opaque_zero, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (int ks = 0; ks < ksteps; ++ks) barrier_keeping_loads();   // the K steps' barriers
                if (wanted) {
                    long long p = -1;
                    if (claims) {
                        p = dyn_first + (long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)tick);
                        if (p >= n_tiles) { p = -1; ended = true; }
                    }
                    const int w = (int)((j + 2) & 3);
                    if (w == 0) m0 = p; else if (w == 1) m1 = p; else if (w == 2) m2 = p; else m3 = p;
                    if (lane == 0) seq[w] = (int)p;
                }
                barrier_keeping_loads();   // the barrier at the tile's end: entry j + 2 is visible from here on
                // ... and the one in front of a re-selection, which the eight others pass when a buffer is full
                if (MTR_TIMING(a) & 2 ? false : any_buffer_full<CAP, TN>(cnt_lds, TM, lane)) barrier_keeping_loads();
            }
            __syncthreads();               // ... and the one in front of the final selection
            if (lane == 0) {
                const unsigned int before = __hip_atomic_fetch_add(a.sync + 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (before + 1u == gridDim.x) {   // the launch's last claimer leaves the counters at zero
                    __hip_atomic_store(a.sync, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(a.sync + 16, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            return;
        }
        auto tile_of = [&](long long j) -> long long {
            if (j < S) return j * R + range;
            return (long long)__builtin_amdgcn_readfirstlane(seq[j & 3]);
        };
        const bool extra_q = kUnevenQ && wave < TNQ / 8;
        // request cursor: K step ld_ks of the workgroup's ld_j-th tile (ld_ptile, -1: past the last tile)
        long long ld_j = 0, ld_ptile = S > 0 ? (long long)range : -1, queued = -1;
        int ld_ks = 0, ld_stage = 0;
        auto request_step = [&]() {
            char* st = stage0 + ld_stage * kStageBytes;
            stage_tile<ROW_AUX, TM, WAVES>(a.rows, ld_ptile * TM, a.n - 1, row_bytes, ld_ks * kStepBytes, st, wave, lane);
            stage_tile<0, TNQ, WAVES>(qbase, 0, TN - 1, row_bytes, ld_ks * kStepBytes, st + TM * kStepBytes, wave, lane);
            ld_stage = ld_stage == NS - 1 ? 0 : ld_stage + 1;
            if (++ld_ks == ksteps) {
                ld_ks = 0;
                ++ld_j;
                ld_ptile = tile_of(ld_j);
                queued = ld_ptile;   // what the compute cursor moves on to at its tile's end
            }
        };
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // (the launcher only picks this form for S >= 2 and at least four K steps per tile)
        request_step();
        request_step();
        long long ptile = (long long)range;
        int ks = 0, cur = 0;
        while (ptile >= 0) {
            // step (ptile, ks) has landed when all but the next step's loads have (this wave's share; the barrier extends it to all)
            const bool has_next = ks + 1 < ksteps || queued >= 0;
            if (!has_next) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (extra_q) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS + 1) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory");
            barrier_keeping_loads();
            // every wave is past the MFMAs of the previous step: its buffer takes the step after next
            if (ld_ptile >= 0) request_step();
            if (ks == 0 && tid < TN)
                pending_floor = __hip_atomic_load(shared_floor + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const char* st = stage0 + cur * kStageBytes;
            mfma_step<MT, NT>(T{}, st, st + TM * kStepBytes, row_off, q_off, lane, acc);
            cur = cur == NS - 1 ? 0 : cur + 1;
            if (ks + 1 == ksteps) {
                const long long row_base = ptile * TM + row_off + 4 * (lane >> 5);
                const bool last_tile = (ptile + 1) * TM > a.n;
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const int ql = q_off + n * 32 + (lane & 31);
                    const float thr = fmaxf(floor_lds[ql], unorder_f32(order_f32(shared_lds[ql]) - 1u));
#pragma unroll
                    for (int m = 0; m < MT; ++m)
                        append_above<CAP>(acc[m][n], thr, ql, row_base + m * 32, last_tile, a, cnt_lds, my_buf);
                }
                barrier_keeping_loads();  // the counts of this tile are visible
                if (tid < TN) shared_lds[tid] = unorder_f32(pending_floor | (pending_floor == 0 ? 0x007fffffu : 0u));
                if (MTR_TIMING(a) & 2 ? false : any_buffer_full<CAP, TN>(cnt_lds, TM, lane)) {
                    // ... and, only when a buffer has to be re-selected, the keys: every wave waits
                    // for its stores (and with them for its loads in flight, which is why the
                    // appends themselves do not wait) in front of one more barrier.
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    barrier_keeping_loads();
                    reselect_where_full<CAP>(wave * QPW, QPW, TM, lane, a, floor_lds, cnt_lds, my_buf, shared_floor, hist_lds);
                }
                ks = 0;
                ptile = queued;
            } else {
                ++ks;
            }
        }
        __syncthreads();
        write_range_lists<CAP>(wave * QPW, QPW, qt * TN, range, lane, a, floor_lds, cnt_lds, my_buf, shared_floor, hist_lds);
        return;
    }
    if constexpr (NS == 3) {
        // Load instructions per wave and step: the row pieces, plus the wave's query pieces
        // (one for the first TNQ / 8 waves only when the query tile has fewer pieces than waves).
        constexpr int LPS_ROWS = TM / (8 * WAVES);
        constexpr bool kUnevenQ = TNQ < 8 * WAVES;
        constexpr int LPS = LPS_ROWS + (kUnevenQ ? 0 : TNQ / (8 * WAVES));
        const bool extra_q = kUnevenQ && wave < TNQ / 8;
        long long ld_tile = tile_lo, ld_ptile = phys(tile_lo);
        int ld_ks = 0, ld_stage = 0;
        auto request_step = [&]() {
            char* st = stage0 + ld_stage * kStageBytes;
            stage_tile<ROW_AUX, TM, WAVES>(a.rows, ld_ptile * TM, a.n - 1, row_bytes, ld_ks * kStepBytes, st, wave, lane);
            stage_tile<0, TNQ, WAVES>(qbase, 0, TN - 1, row_bytes, ld_ks * kStepBytes, st + TM * kStepBytes, wave, lane);
            ld_stage = ld_stage == NS - 1 ? 0 : ld_stage + 1;
            if (++ld_ks == ksteps) {
                ld_ks = 0;
                ++ld_tile;
                ld_ptile = phys(ld_tile);
            }
        };
        // The floors above were plain loads; nothing but tile requests may be in
        // flight when the counted waits start.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (total_steps > 0) request_step();
        if (total_steps > 1) request_step();
        long long tile = tile_lo, ptile = phys(tile_lo);
        int ks = 0, cur = 0;
        for (long long s = 0; s < total_steps; ++s) {
            // step s has landed (this wave's share; the barrier extends it to all)
            if (s + 1 >= total_steps) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (extra_q) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS + 1) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory");
            barrier_keeping_loads();
            // every wave is past the MFMAs of step s - 1: its buffer takes step s + 2
            if (s + 2 < total_steps) request_step();
            if (ks == 0 && tid < TN)
                pending_floor = __hip_atomic_load(shared_floor + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const char* st = stage0 + cur * kStageBytes;
            if (!(MTR_TIMING(a) & 32)) mfma_step<MT, NT>(T{}, st, st + TM * kStepBytes, row_off, q_off, lane, acc);
            cur = cur == NS - 1 ? 0 : cur + 1;
            if (ks + 1 == ksteps) {
                const long long row_base = ptile * TM + row_off + 4 * (lane >> 5);
                const bool last_tile = (ptile + 1) * TM > a.n;
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const int ql = q_off + n * 32 + (lane & 31);
                    const float thr = fmaxf(floor_lds[ql], unorder_f32(order_f32(shared_lds[ql]) - 1u));
#pragma unroll
                    for (int m = 0; m < MT; ++m)
                        append_above<CAP>(acc[m][n], thr, ql, row_base + m * 32, last_tile, a, cnt_lds, my_buf);
                }
                barrier_keeping_loads();  // the counts of this tile are visible
                if (tid < TN) shared_lds[tid] = unorder_f32(pending_floor | (pending_floor == 0 ? 0x007fffffu : 0u));
                if (MTR_TIMING(a) & 2 ? false : any_buffer_full<CAP, TN>(cnt_lds, TM, lane)) {
                    // ... and, only when a buffer has to be re-selected, the keys: every wave waits
                    // for its stores (and with them for its loads in flight, which is why the
                    // appends themselves do not wait) in front of one more barrier.
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    barrier_keeping_loads();
                    reselect_where_full<CAP>(wave * QPW, QPW, TM, lane, a, floor_lds, cnt_lds, my_buf, shared_floor, hist_lds);
                }
                ks = 0;
                ++tile;
                ptile = phys(tile);
            } else {
                ++ks;
            }
        }
        __syncthreads();
        write_range_lists<CAP>(wave * QPW, QPW, qt * TN, range, lane, a, floor_lds, cnt_lds, my_buf, shared_floor, hist_lds);
        return;
    }
    if (total_steps > 0) {
        stage_tile<ROW_AUX, TM, WAVES>(a.rows, phys(tile_lo) * TM, a.n - 1, row_bytes, 0, stage0, wave, lane);
        stage_tile<0, TNQ, WAVES>(qbase, 0, TN - 1, row_bytes, 0, stage0 + TM * kStepBytes, wave, lane);
    }
    __syncthreads();

    long long tile = tile_lo;
    long long ptile = phys(tile_lo), ptile_next = phys(tile_lo + 1);
    int ks = 0;
    for (long long s = 0; s < total_steps; ++s) {
        const int cur = (int)(s & 1);
        if (ks == 0 && tid < TN)
            pending_floor = __hip_atomic_load(shared_floor + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (s + 1 < total_steps && !(MTR_TIMING(a) & 64)) {
            const int nks = ks + 1 == ksteps ? 0 : ks + 1;
            const long long ntile = ks + 1 == ksteps ? ptile_next : ptile;
            stage_tile<ROW_AUX, TM, WAVES>(a.rows, ntile * TM, a.n - 1, row_bytes, nks * kStepBytes, stage0 + (cur ^ 1) * kStageBytes, wave, lane);
            stage_tile<0, TNQ, WAVES>(qbase, 0, TN - 1, row_bytes, nks * kStepBytes, stage0 + (cur ^ 1) * kStageBytes + TM * kStepBytes, wave, lane);
        }
        if constexpr (N16) {
            mfma_step16<NQ16>(stage0 + cur * kStageBytes, stage0 + cur * kStageBytes + TM * kStepBytes, row_off, lane, acc16);
        } else if (!(MTR_TIMING(a) & 32))
        mfma_step<MT, NT>(T{}, stage0 + cur * kStageBytes, stage0 + cur * kStageBytes + TM * kStepBytes, row_off, q_off,
                          lane, acc);

        const bool tile_done = ks + 1 == ksteps;
        if constexpr (MAX_ONLY) { if (tile_done) {
            // Sampling launch: keep nothing but the tile's best score per query.  Rows
            // past the end of the matrix were staged as copies of the last row, so
            // they cannot raise a maximum.
            if constexpr (N16) {
#pragma unroll
                for (int n = 0; n < NQ16; ++n) {
                    float mx = -__builtin_huge_valf();
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            mx = fmaxf(mx, acc16[m][n][r]);
                            acc16[m][n][r] = 0.f;
                        }
                    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                    if (lane < 16) atomicMax(&best_lds[n * 16 + lane], order_f32(mx));
                }
            } else
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                float mx = -__builtin_huge_valf();
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        mx = fmaxf(mx, acc[m][n][r]);
                        acc[m][n][r] = 0.f;
                    }
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                if (lane < 32) atomicMax(&best_lds[q_off + n * 32 + lane], order_f32(mx));
            }
        } } else if (tile_done) {
            // Scores of the 128-row x TNQ-query tile are complete: append what clears each query's floor.
            const long long row_base = ptile * TM + row_off + 4 * (lane >> 5);
            const bool last_tile = (ptile + 1) * TM > a.n;
            if constexpr (N16) {
#pragma unroll
                for (int n = 0; n < NQ16; ++n) {
                    const int ql = n * 16 + (lane & 15);
                    const float thr = fmaxf(floor_lds[ql], unorder_f32(order_f32(shared_lds[ql]) - 1u));
#pragma unroll
                    for (int m = 0; m < 2; ++m)
                        append_above16<CAP>(acc16[m][n], thr, ql, ptile * TM + row_off + m * 16 + 4 * (lane >> 4), last_tile, a, cnt_lds, my_buf);
                }
            } else
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int ql = q_off + n * 32 + (lane & 31);
                // "v > own floor && v >= shared floor" as ONE strict threshold: v >= sh is
                // v > (largest float below sh); a NaN from sh = -inf loses the fmaxf.
                const float thr = fmaxf(floor_lds[ql], unorder_f32(order_f32(shared_lds[ql]) - 1u));
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    append_above<CAP>(acc[m][n], thr, ql, row_base + m * 32, last_tile, a, cnt_lds, my_buf);
            }
        }
        __syncthreads();
        if (MAX_ONLY && tile_done) {
            if (tid < TN) {
                a.tile_max[((size_t)qt * TN + tid) * a.n_tiles + tile] = best_lds[tid];
                best_lds[tid] = 0u;
            }
            ks = 0;
            ++tile;
            ptile = ptile_next;
            ptile_next = phys(tile + 1);
            __syncthreads();  // the next tile's maxima start from the cleared words
        } else if (tile_done) {
            // Wave w looks after queries [QPW*w, QPW*(w+1)): re-select any whose buffer
            // could overflow on the next tile (a tile adds at most TM entries).
            if (tid < TN) shared_lds[tid] = unorder_f32(pending_floor | (pending_floor == 0 ? 0x007fffffu : 0u));
            reselect_where_full<CAP>(wave * QPW, QPW, TM, lane, a, floor_lds, cnt_lds, my_buf, shared_floor, hist_lds);
            ks = 0;
            ++tile;
            ptile = ptile_next;
            ptile_next = phys(tile + 1);
            if (ksteps == 1) __syncthreads();  // no other barrier separates this from the next tile's appends
        } else {
            ++ks;
        }
    }

    if constexpr (MAX_ONLY) return;
    // Final selection per query and one candidate list per (query, range).
    __syncthreads();
    write_range_lists<CAP>(wave * QPW, QPW, qt * TN, range, lane, a, floor_lds, cnt_lds, my_buf, shared_floor, hist_lds);
}

}  // namespace


int plan_mfma(int dtype, int64_t n, int ld, int b, int k, int num_cus, int sampling, MfmaPlan* plan) {
    const int esize = dtype == 0 ? 4 : 2;
    if ((ld * esize) % kStepBytes != 0 || ld * esize < kStepBytes || k < 1 || k > 256 || b < 1 || n < 1) return -1;
    const bool one_step = ld * esize == kStepBytes;   // rows of 128 bytes: one K step per tile, the two-buffer 128-row forms only
    // Query-tile widths: 32, 64 and 128 (32 x 32 MFMA tiles, both storage types); fp32 shards,
    // whose batches are bound by the fp32 matrix rate, also have 16, 48, 80, 96 and 112
    // (16 x 16 tiles, see mfma_step16) so that a tile is never more than 15 columns of padding.
    // The batch is cut into the fewest tiles of at most 128, and those are as narrow as they can be.
    static const int kWidthsF32[] = {16, 32, 48, 64, 80, 96, 112, 128};
    static const int kWidthsBf16[] = {32, 64, 128};
    const int* widths = dtype == 0 ? kWidthsF32 : kWidthsBf16;
    const int n_widths = dtype == 0 ? 8 : 3;
    int wi = 0;
    {
        const int tiles = (b + 127) / 128, per_tile = (b + tiles - 1) / tiles;
        while (wi + 1 < n_widths && widths[wi] < per_tile) ++wi;
    }
    // A shard of few row tiles cannot fill the CUs with one workgroup per (row
    // tile, query tile) pair: narrower query tiles give more, shorter workgroups
    // (4032 x 4096 fp32, 34 queries: 32 workgroups of 64 queries 0.222 ms, 64 of 32
    // queries 0.130 ms, 96 of 16 queries 0.095 ms; the rows are read once more, out of L2).
    {
        const long long row_tiles = (n + kTileRows - 1) / kTileRows;
        while (wi > 0 && row_tiles * ((b + widths[wi] - 1) / widths[wi]) < num_cus) --wi;
    }
    plan->tnq = widths[wi];
    const Tuning& tn = tuning();
    for (int i = 0; i < n_widths; ++i)
        if (tn.mfma_tnq == widths[i]) plan->tnq = tn.mfma_tnq;
    plan->n_qtiles = (b + plan->tnq - 1) / plan->tnq;
    // Row loads carry the streaming (nt) policy when every row byte is fetched
    // once per launch -- a single query tile -- and the rows exceed the Infinity
    // Cache: 16-byte direct-to-LDS loads then read 7.05 TB/s against 6.2 TB/s
    // under the default policy (profiles/micro/tile_read_patterns.hip).  With
    // several query tiles the later ones find the rows in L2, which streaming
    // lines would leave first.  (A sampling launch reads a small part of the
    // rows, which the full launch reads again: default policy, two buffers.)
    plan->stream_rows = (!sampling && plan->n_qtiles == 1 && (double)n * ld * esize > 192.0 * 1024 * 1024) ? 1 : 0;
    if (tn.mfma_stream_rows >= 0) plan->stream_rows = !sampling && tn.mfma_stream_rows != 0;
    // ... and, for the 32- and 64-query tiles of bf16 shards, the three-buffer
    // ring (12.5M x 768 bf16, 64 queries: 3.16 -> 3.07 ms; 1M x 1024: 0.394 ->
    // 0.375 ms; fp32 shards are bound by the fp32 matrix rate there and lose 1-2 %
    // to the ring's two workgroups per CU).
    // The 128-query tile has the ring only in its 256-row form (below).
    const bool tall_shape = dtype == 2 && k <= 128 && !one_step;   // 256-row tiles (eight waves): bf16, every tile width
    plan->stages = (plan->stream_rows && dtype == 2 && (plan->tnq <= 64 || tall_shape)) ? 3 : 2;
    if (tn.mfma_stages) plan->stages = (!sampling && tn.mfma_stages == 3 && (plan->tnq <= 64 || tall_shape)) ? 3 : 2;
    // ... with 256-row tiles (eight waves) for the 64- and 128-query tiles (bf16, k <= 128).
    int tm = kTileRows;
    const bool tall_ok = plan->stages == 3 && tall_shape;
    if (tall_ok) tm = 256;
    if (tn.mfma_tile_rows) tm = (tn.mfma_tile_rows == 256 && tall_ok) ? 256 : kTileRows;
    if (tm == kTileRows && plan->tnq == 128) plan->stages = 2;
    if (plan->tnq % 32 != 0 || plan->tnq == 96 || one_step) plan->stages = 2;   // the 16 x 16 forms have two buffers
    plan->cap = (k <= 32 && tm == 128) ? 256 : 512;
    if ((tn.mfma_cap == 256 || tn.mfma_cap == 512) && k + tm <= tn.mfma_cap) plan->cap = tn.mfma_cap;
    plan->slack = (plan->cap - tm - k) / 4;
    if (plan->slack > 64) plan->slack = 64;
    plan->tm = tm;
    const long long n_tiles = (n + tm - 1) / tm;
    plan->n_tiles = n_tiles;
    // Workgroups per CU (= row ranges x query tiles / CUs): more, shorter row
    // ranges cost more in top-k selection than they gain in bytes in flight.
    // Measured best with two buffers: 3 for the 32-query tile (four fit), 2 for
    // the 64- and 128-query tiles.  Two ring workgroups fit a CU's LDS, one of
    // the eight-wave 256-row form.
    long long per_cu = tm == 256 ? 1 : plan->stages == 3 ? 2 : plan->tnq <= 32 ? 3 : 2;
    if (tn.mfma_per_cu > 0 && tm != 256) per_cu = tn.mfma_per_cu;
    // rounded down: a grid a few workgroups above per_cu * CUs runs those in a second round
    long long ranges = per_cu * num_cus / plan->n_qtiles;
    if (ranges > n_tiles) ranges = n_tiles;
    if (ranges < 1) ranges = 1;
    plan->tiles_per_range = (n_tiles + ranges - 1) / ranges;
    plan->n_ranges = (int)((n_tiles + plan->tiles_per_range - 1) / plan->tiles_per_range);
    plan->grid = plan->n_qtiles * plan->n_ranges;
    // One query tile streaming a large shard: the workgroups sweep the rows side by side.
    plan->interleave = (plan->stream_rows && plan->n_qtiles == 1) ? 1 : 0;
    if (tn.mfma_interleave >= 0) plan->interleave = !sampling && tn.mfma_interleave != 0;
    plan->lds = lds_bytes(tm, plan->tnq, plan->stages);
    // ... and take the last tiles from a counter (tall ring forms of up to 64 queries, rows of at
    // least four K steps, many tiles per range).  Measured on 10M x 1024 and 12.5M x 768 bf16: 0.5 to
    // 1.5 % off the launch, except the 64-query tile over rows of 2 KB and more, which loses 0.2 %.
    plan->dynamic_tail = 0;
    plan->static_tiles = 0;
    {
        const long long per_range = plan->n_tiles / plan->n_ranges;
        const bool can = plan->interleave && tm == 256 && plan->stages == 3 && plan->tnq <= 64 && ld * esize >= 4 * kStepBytes &&
                         per_range >= 4;
        const bool gains = plan->tnq <= 32 || (long long)ld * esize < 2048;
        const bool want = tn.mfma_dynamic_tail >= 0 ? tn.mfma_dynamic_tail != 0 : per_range >= 16 && gains;
        if (can && want) {
            plan->dynamic_tail = 1;
            const int permille = tn.mfma_static_permille > 0 ? tn.mfma_static_permille : 930;
            plan->static_tiles = (long long)((double)plan->n_tiles * permille * 1e-3) / plan->n_ranges;
            if (plan->static_tiles < 2) plan->static_tiles = 2;
            plan->lds += 16;   // the claimer's four-entry tile ring
        }
    }
    // Rows through registers (64- and 128-query tiles of the tall ring form, rows of whole 512 bytes).
    plan->via_regs = 0;
    {
        const bool can = plan->stream_rows && tm == 256 && plan->stages == 3 && dtype == 2 && (plan->tnq == 64 || plan->tnq == 128) &&
                         ((long long)ld * esize) % (kRegDepth * kStepBytes) == 0;
        // (the 64-query tile gains 0.3 % where it does not claim its tail, nothing where it does)
        const bool want = tn.mfma_via_regs >= 0 ? tn.mfma_via_regs != 0 : plan->tnq == 128 || !plan->dynamic_tail;
        if (can && want) {
            // 2: on 16 x 16 x 32 MFMAs (the chip holds a higher clock under them; floors then come from the
            // filter kernel's sampling form, which multiplies the same way); 1: on 32 x 32 x 16 like the other forms
            plan->via_regs = tn.mfma_via_regs == 1 ? 1 : 2;
            plan->dynamic_tail = 0;
            plan->static_tiles = 0;
            plan->lds = lds_bytes_regs(plan->tnq);
        }
    }
    plan->cand_bytes = (size_t)plan->grid * plan->tnq * plan->cap * 8;
    return 0;
}

hipError_t launch_score_mfma(const void* rows, int dtype, int64_t n, int ld, const void* queries, int b, int k,
                             const MfmaPlan& plan, const MfmaPass& pass, void* cand_buf, unsigned int* shared_floor,
                             float* out_s, int* out_i, hipStream_t stream) {
    // the kernels index queries, floors and per-tile maxima up to n_qtiles * tnq, padding included
    if (plan.n_qtiles * plan.tnq > pass.padded_queries || b > plan.n_qtiles * plan.tnq) return hipErrorInvalidValue;
    MfmaArgs a;
    a.rows = static_cast<const char*>(rows);
    a.n = n;
    a.ld = ld;
    a.queries = static_cast<const char*>(queries);
    a.b = b;
    a.n_qtiles = plan.n_qtiles;
    a.n_ranges = plan.n_ranges;
    a.tiles_per_range = plan.tiles_per_range;
    a.n_tiles = plan.n_tiles;
    a.tile_stride = pass.tile_stride < 1 ? 1 : pass.tile_stride;
    a.tile_max = pass.tile_max;
    a.k = k;
    a.slack = plan.slack;
    a.debug_no_candidates = 0;
    a.debug_counters = nullptr;
    a.debug_floor = 0.f;
#ifdef MTR_TIMING_SWITCHES
    a.debug_no_candidates = tuning().timing_mfma;
    a.debug_floor = tuning().timing_floor;
    if (a.debug_no_candidates & 4) {
        static unsigned long long* counters = nullptr;
        if (!counters && hipMalloc(reinterpret_cast<void**>(&counters), 16) != hipSuccess) return hipErrorOutOfMemory;
        unsigned long long host[2] = {0, 0};
        (void)hipMemcpy(host, counters, 16, hipMemcpyDeviceToHost);
        fprintf(stderr, "[mtr debug] previous launch: appends=%llu re-selections=%llu\n", host[0], host[1]);
        (void)hipMemset(counters, 0, 16);
        a.debug_counters = counters;
    }
#endif
    a.cand_buf = static_cast<unsigned long long*>(cand_buf);
    a.shared_floor = shared_floor;
    a.out_s = out_s;
    a.out_i = out_i;
    a.run_if = pass.run_if;
    a.interleave = plan.interleave && a.tile_stride == 1 && a.tile_max == nullptr;
    const bool dyn = plan.dynamic_tail && a.interleave && pass.sync != nullptr && plan.tm == 256 && plan.stages == 3 && plan.tnq <= 64;
    a.static_tiles = dyn ? plan.static_tiles : 0;
    a.sync = pass.sync;
    auto launch = [&](auto kernel) -> hipError_t {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)plan.lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kernel, dim3(plan.grid), dim3(dyn ? 576 : plan.tm == 256 ? 512 : kThreads), plan.lds, stream, a);
        return hipGetLastError();
    };
#define MTR_MFMA_PICK(T, AUX)                                                                         \
    if (a.tile_max != nullptr) switch (plan.tnq) {                                                    \
        case 32: return launch(&score_topk_mfma_kernel<T, 256, 32, true, 0, 2>);                      \
        case 64: return launch(&score_topk_mfma_kernel<T, 256, 64, true, 0, 2>);                      \
        case 128: return launch(&score_topk_mfma_kernel<T, 256, 128, true, 0, 2>);                    \
        default: return hipErrorInvalidValue;                                                         \
    }                                                                                                 \
    switch (plan.tnq + (plan.cap == 256 ? 0 : 1)) {                                                   \
        case 32: return launch(&score_topk_mfma_kernel<T, 256, 32, false, AUX, 2>);                   \
        case 33: return launch(&score_topk_mfma_kernel<T, 512, 32, false, AUX, 2>);                   \
        case 64: return launch(&score_topk_mfma_kernel<T, 256, 64, false, AUX, 2>);                   \
        case 65: return launch(&score_topk_mfma_kernel<T, 512, 64, false, AUX, 2>);                   \
        case 128: return launch(&score_topk_mfma_kernel<T, 256, 128, false, AUX, 2>);                 \
        default: return launch(&score_topk_mfma_kernel<T, 512, 128, false, AUX, 2>);                  \
    }
#define MTR_MFMA_PICK_RING(T)                                                                         \
    switch (plan.tnq + (plan.cap == 256 ? 0 : 1)) {                                                   \
        case 32: return launch(&score_topk_mfma_kernel<T, 256, 32, false, 2, 3>);                     \
        case 33: return launch(&score_topk_mfma_kernel<T, 512, 32, false, 2, 3>);                     \
        case 64: return launch(&score_topk_mfma_kernel<T, 256, 64, false, 2, 3>);                     \
        case 65: return launch(&score_topk_mfma_kernel<T, 512, 64, false, 2, 3>);                     \
        default: return hipErrorInvalidValue;                                                         \
    }
#define MTR_MFMA_PICK16(TNQ16)                                                                                      \
    {                                                                                                                \
        if (dtype != 0 || plan.stages != 2 || plan.tm != kTileRows) return hipErrorInvalidValue;                     \
        if (a.tile_max != nullptr) return launch(&score_topk_mfma_kernel<float, 256, TNQ16, true, 0, 2>);            \
        if (plan.cap == 256) return plan.stream_rows ? launch(&score_topk_mfma_kernel<float, 256, TNQ16, false, 2, 2>) \
                                                     : launch(&score_topk_mfma_kernel<float, 256, TNQ16, false, 0, 2>); \
        return plan.stream_rows ? launch(&score_topk_mfma_kernel<float, 512, TNQ16, false, 2, 2>)                    \
                                : launch(&score_topk_mfma_kernel<float, 512, TNQ16, false, 0, 2>);                   \
    }
    if (plan.tnq == 16) MTR_MFMA_PICK16(16)
    if (plan.tnq == 48) MTR_MFMA_PICK16(48)
    if (plan.tnq == 80) MTR_MFMA_PICK16(80)
    if (plan.tnq == 96) MTR_MFMA_PICK16(96)
    if (plan.tnq == 112) MTR_MFMA_PICK16(112)
#undef MTR_MFMA_PICK16
    if (plan.tm == 256) {
        if (plan.stages != 3 || plan.cap != 512 || dtype != 2 || a.tile_max != nullptr) return hipErrorInvalidValue;
        if (plan.via_regs == 2 && plan.tnq == 64) return launch(&score_topk_mfma_kernel<bf16_t, 512, 64, false, 2, 3, 256, false, true, true>);
        if (plan.via_regs == 2 && plan.tnq == 128) return launch(&score_topk_mfma_kernel<bf16_t, 512, 128, false, 2, 3, 256, false, true, true>);
        if (plan.via_regs && plan.tnq == 64) return launch(&score_topk_mfma_kernel<bf16_t, 512, 64, false, 2, 3, 256, false, true>);
        if (plan.via_regs && plan.tnq == 128) return launch(&score_topk_mfma_kernel<bf16_t, 512, 128, false, 2, 3, 256, false, true>);
        if (dyn && plan.tnq == 32) return launch(&score_topk_mfma_kernel<bf16_t, 512, 32, false, 2, 3, 256, true>);
        if (dyn && plan.tnq == 64) return launch(&score_topk_mfma_kernel<bf16_t, 512, 64, false, 2, 3, 256, true>);
        if (plan.tnq == 32) return launch(&score_topk_mfma_kernel<bf16_t, 512, 32, false, 2, 3, 256>);
        if (plan.tnq == 64) return launch(&score_topk_mfma_kernel<bf16_t, 512, 64, false, 2, 3, 256>);
        if (plan.tnq == 128) return launch(&score_topk_mfma_kernel<bf16_t, 512, 128, false, 2, 3, 256>);
        return hipErrorInvalidValue;
    }
    if (plan.stages == 3 && a.tile_max == nullptr) {
        if (dtype == 0) { MTR_MFMA_PICK_RING(float) }
        MTR_MFMA_PICK_RING(bf16_t)
    }
    if (plan.stream_rows) {
        if (dtype == 0) { MTR_MFMA_PICK(float, 2) }
        MTR_MFMA_PICK(bf16_t, 2)
    }
    if (dtype == 0) { MTR_MFMA_PICK(float, 0) }
    MTR_MFMA_PICK(bf16_t, 0)
#undef MTR_MFMA_PICK_RING
#undef MTR_MFMA_PICK
}

namespace {
// One wave per query: the k-th largest of its per-tile best scores.  Each of
// those k tiles holds a row scoring at least that much, so no row below it can
// be among the query's k best: it becomes the query's starting floor.  Samples of
// more than kMaxSampleKeys tiles are taken `group` neighbouring tiles at a time
// (the best score of the group: still one row each, so the bound holds).
template <int E>
__global__ __launch_bounds__(64) void floor_from_tile_max_kernel(const unsigned int* tile_max, int n_tiles, int group, int k,
                                                                 unsigned int* shared_floor) {
    __shared__ unsigned int hist[256];
    const int q = blockIdx.x;
    const int lane = threadIdx.x;
    const unsigned int* mine = tile_max + (size_t)q * n_tiles;
    unsigned long long key[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int slot = e * 64 + lane;
        unsigned int best = 0u;
        for (int g = 0; g < group; ++g) {
            const int t = slot * group + g;
            if (t < n_tiles) best = max(best, mine[t]);
        }
        // the slot number makes equal scores distinct keys; 0 marks an empty slot
        key[e] = best != 0u ? ((unsigned long long)best << 32) | (unsigned int)(slot + 1) : 0ull;
    }
    int passes = 0;
    const unsigned long long cut = wave_kth_largest_hist<E>(key, k, 0, hist, lane, &passes);
    // The walk stops once exactly k keys are at or above the cut, which may leave the cut's
    // low bytes open: the k-th largest itself is the smallest of those k.
    unsigned long long kth = ~0ull;
#pragma unroll
    for (int e = 0; e < E; ++e)
        if (key[e] != 0ull && key[e] >= cut && key[e] < kth) kth = key[e];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const unsigned long long o = __shfl_xor(kth, off, 64);
        kth = o < kth ? o : kth;
    }
    if (lane == 0) shared_floor[q] = cut != 0ull ? (unsigned int)(kth >> 32) : 0u;  // at most k tiles: no floor
}
}  // namespace

hipError_t launch_floor_from_tile_max(const unsigned int* tile_max, int n_tiles, int b, int k,
                                      unsigned int* shared_floor, hipStream_t stream) {
    if (n_tiles < 1 || n_tiles > kMaxSampleTiles) return hipErrorInvalidValue;
    const int group = (n_tiles + kMaxSampleKeys - 1) / kMaxSampleKeys;
    const int slots = (n_tiles + group - 1) / group;
    if (slots <= 256)
        hipLaunchKernelGGL(floor_from_tile_max_kernel<4>, dim3(b), dim3(64), 0, stream, tile_max, n_tiles, group, k, shared_floor);
    else
        hipLaunchKernelGGL(floor_from_tile_max_kernel<16>, dim3(b), dim3(64), 0, stream, tile_max, n_tiles, group, k, shared_floor);
    return hipGetLastError();
}

}  // namespace mtr
