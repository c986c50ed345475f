// Large-batch scoring of a bf16 shard on the matrix cores: 256 rows x 256
// queries per tile, eight waves, and nothing in the loop but a threshold test.
//
// Same job as score_mfma.hip (the many-queries form of retrieve_top_index,
// pipeline/utils/local_rag.py:76-78, for the batch built at
// pipeline/async_main.py:193-195), for batches of hundreds of queries and more,
// where the 128 x 128 tile of that kernel is bound by what one CU can move from
// L2 into LDS: a 256 x 256 tile needs half the bytes per flop.
//
// The price of the larger tile is registers: 128 accumulators per lane plus
// the operand fragments leave no room for the in-loop re-selection that
// score_mfma.hip uses to keep its candidate buffers short (the compiler spills,
// and a reload stalls the direct-to-LDS loads behind it).  So this kernel
// keeps no floor of its own.  It takes each query's floor as an input -- the
// k-th largest of the per-tile best scores of a sampling launch, which k
// different rows reach, so no row below it can be among the k best -- and
// appends every score that reaches the floor, as (row, score bits),
// to a list private to the lane (eight lanes of a workgroup hold a query's
// scores: eight lists per workgroup and query), counting in a register.  A
// list that overflows keeps counting; the selection kernel that follows
// (merge_topk.hip, launch_select_keys) sees the count, raises a flag, and the
// caller's already enqueued launch of the general kernel, which does nothing
// unless the flag is set, recomputes the batch.  Results are exact either way.
// The floors come from this kernel's own sampling form (MAX_ONLY: every
// stride-th row tile, only each query's best score per tile is kept), so that
// floors and filtered scores are the same arithmetic to the last bit: the k
// rows behind a floor reach it again in the full launch.
//
// Work split.  One workgroup per (query tile, row list); row list r of R holds
// tiles r, r + R, r + 2R, ...: neighbouring rows are often about the same
// subject (pages of one book), and interleaving spreads a query's hits evenly
// over the lists, which is what sizes the buffers.
//
// Tile.  Waves 2 x 4: wave (wm, wn) owns rows [128 wm, +128) x queries
// [64 wn, +64) as 8 x 4 MFMA tiles of 16 x 16 (v_mfma_f32_16x16x32_bf16: the
// same cycles per flop and LDS reads as 4 x 2 tiles of 32 x 32 x 16, but the
// chip holds a higher clock under it -- 7.28 against 7.84 ms for the bare loop,
// profiles/micro/mfma256_core.hip), K steps of 64 elements.  LDS holds two K steps, each as four half tiles of 128
// rows x 128 bytes (rows 0-127, rows 128-255, queries 0-127, queries 128-255),
// filled by 16-byte direct-to-LDS loads whose 16-byte chunks are XOR-swizzled
// with the row number on the source address.  The two wave groups (wm = 0, 1)
// run one barrier apart, so that one is on the matrix cores while the other
// reads LDS.  Every wave executes the same barrier sequence, three per K step;
// group 1 executes one extra barrier first and group 0 one extra at the end.
// In group-0 barrier slots, step s: slot 3s (then it reads its query and low
// row fragments), 3s+1 (MFMA low, reads high rows), 3s+2 (MFMA high); group 1
// does the same at 3s+1, 3s+2, 3s+3.
//   landed:  the query halves of step s+1 are loaded by all eight waves; each
//            waits for its part before its LAST barrier of step s, i.e. before
//            slot 3s+3 at the latest, and the first reader starts after 3s+3.
//            A row half is loaded and read by ONE group: its waves wait before
//            their first barrier of step s+1.
//   refill:  after its last barrier of step s a wave requests the queries and
//            its group's rows of step s+2 into the buffers step s used: group 0
//            after slot 3s+2, when group 1 finished its query reads (before its
//            second barrier = 3s+2); a group's row reads end before its own
//            last barrier.
// Waits on the loads are counted (s_waitcnt vmcnt(N)).  Loads return in order,
// so "at most N operations outstanding" implies that all but the newest N loads
// have landed, whatever the appends' stores (which may complete in any order
// relative to loads) contribute to the count: they only make a wait longer.
//
// Algorithmic work per launch: 2 * B * n * d flops (fp32 accumulate).
#include <stdio.h>
#include <stdlib.h>

#include "mtr_device.h"
#include "wave_select.h"

namespace mtr {

namespace {

constexpr int kWideThreads = 512;
constexpr int kWideTile = 256;                  // rows per tile, and the widest query tile
constexpr int kHalfBytes = 128 * 128;           // 128 rows x 128 bytes
constexpr int kWideStepBytes = 4 * kHalfBytes;  // rows lo, rows hi, queries lo, queries hi
constexpr size_t kWideLds = 2 * kWideStepBytes;

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

// Timing-experiment switches (results are then wrong by design) exist only in
// builds with -DMTR_TIMING_SWITCHES; the default build folds them to zero.
#ifdef MTR_TIMING_SWITCHES
#define MTR_TIMING(a) (a).debug
#else
#define MTR_TIMING(a) 0
#endif

struct WideArgs {
    const char* rows;         // [n][ld] bf16
    long long n;
    int ld;                   // elements, a multiple of 64
    const char* queries;      // [n_qtiles*TQ][ld] bf16, zero rows past b (TQ: queries per tile, 128 / 192 / 256)
    int b;
    int n_lists;              // row lists (workgroups per query tile)
    int n_qtiles;
    int qgroup;               // query tiles that share a row list on one XCD (see the kernel)
    long long n_tiles;        // tiles this launch covers (numbered 0.. in its own order)
    int tile_stride;          // tile t of the launch is tile t * tile_stride of the matrix
    unsigned int* tile_max;   // sampling form: [n_qtiles*TQ][n_tiles] best score per (query, tile), zeroed by the caller
    const unsigned int* floor;    // [b] order-preserving score bits (0: no floor)
    unsigned long long* keys;     // [n_qtiles][n_lists][TQ][kWideSubs][cap]: row | (unsigned long long)score bits << 32
    unsigned int* counts;         // [n_qtiles][n_lists][TQ][kWideSubs] scores that reached the floor (may exceed cap)
    int cap;
    int debug;                // timing experiments only: bit 0 nothing reaches the floor
};

// Addresses of the tile loads are a wave-uniform base (scalar registers) plus a
// 32-bit lane offset: lane L of a 1 KiB piece fetches 16-byte chunk (L % 8) ^ (L / 8)
// of row L / 8 of the piece's eight rows.  Kept in that form, a request costs no
// 64-bit vector address per piece, which matters once its pieces are spread
// between the MFMAs and their addresses live that long.
// A pointer the compiler must treat as one opaque scalar value (it otherwise
// folds the lane offset into the 64-bit sum and carries a vector address).
__device__ __forceinline__ const char* scalar_pointer(const char* p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)v);
    const unsigned int hi = __builtin_amdgcn_readfirstlane((unsigned int)(v >> 32));
    return reinterpret_cast<const char*>(((unsigned long long)hi << 32) | lo);
}

__device__ __forceinline__ unsigned int lane_piece_offset(int lane, unsigned int row_bytes) {
    return (unsigned int)(lane >> 3) * row_bytes + (unsigned int)(((lane & 7) ^ (lane >> 3)) << 4);
}

// This wave's share (NN of 8 NN instructions) of the query tile of a K step: 64 NN queries,
// eight rows a piece, in tile order behind the two row halves.
// wave: wave-uniform (scalar) wave number.
template <int NN>
__device__ __forceinline__ void request_queries(const char* qbase, long long row_bytes, int k0_bytes, char* dst, int wave,
                                                unsigned int lane_off) {
#pragma unroll
    for (int i = 0; i < NN; ++i) {
        const int t = wave * NN + i;
        const char* base = scalar_pointer(qbase + (long long)(t * 8) * row_bytes + k0_bytes);
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(base + lane_off), (lds_ptr_t)(dst + 2 * kHalfBytes + t * 1024), 16, 0, 0);
    }
}

// ... and of ITS GROUP's row half (16 instructions over the group's 4 waves).
// Rows past the end of the matrix are fetched as copies of the last row.
template <int AUX>
__device__ __forceinline__ void request_rows(const char* rows, long long row0, long long row_max, long long row_bytes,
                                             int k0_bytes, char* dst, int wave_in_group, int lane, unsigned int lane_off) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int t = wave_in_group * 4 + i;
        const long long g0 = row0 + t * 8;                                   // the piece's first row (uniform)
        const long long gbase = g0 < row_max ? g0 : row_max;
        const char* base = scalar_pointer(rows + gbase * row_bytes + k0_bytes);
        // Rows of the piece beyond its first that exist: 7 everywhere but in the last
        // tile, where every lane's row is clamped to the last one (no branch: a
        // branch here would cut the requests off from the MFMAs they are spread between).
        const long long left = row_max - g0;
        const int valid = left < 0 ? 0 : left > 7 ? 7 : (int)left;
        const int r = (lane >> 3) < valid ? (lane >> 3) : valid;
        const unsigned int off = (unsigned int)r * (unsigned int)row_bytes + (unsigned int)(((lane & 7) ^ (lane >> 3)) << 4);
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(base + off), (lds_ptr_t)(dst + t * 1024), 16, 0, AUX);
    }
}

__device__ __forceinline__ bf16x8 read_frag(const char* lds, int row, int chunk) {
    return *reinterpret_cast<const bf16x8*>(lds + row * 128 + ((chunk ^ (row & 7)) << 4));
}

// s_barrier without the drain of outstanding loads that __syncthreads() adds.
__device__ __forceinline__ void barrier_keeping_loads() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// One K step of a wave's low rows (row tiles 0-3): the eight MFMAs of a row
// tile (4 query tiles x two K blocks of 32), then the two LDS reads that replace
// its fragments with those of the high rows, which land while the next row tiles
// multiply.  The order is pinned with sched_group_barrier; left to itself the
// compiler issues all reads after all MFMAs and the wave then waits for them.
// FIRST: the sums start from zero instead of acc.
template <bool FIRST, int NN>
__device__ __forceinline__ void mfma_low_rows(bf16x8 (&fa)[4][2], const bf16x8 (&fb)[NN][2], f32x4 (&acc)[8][NN],
                                              const char* sa, int lane) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
#pragma unroll
        for (int nn = 0; nn < NN; ++nn) {
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            f32x4 c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[m][0], fb[nn][0], FIRST ? zero : acc[m][nn], 0, 0, 0);
            acc[m][nn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[m][1], fb[nn][1], c, 0, 0, 0);
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) fa[m][kk] = read_frag(sa, 64 + m * 16 + (lane & 15), 4 * kk + (lane >> 4));
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2 * NN, 0);  // the MFMAs of a row tile
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);       // two LDS reads
    }
}

// ... and of its high rows (row tiles 4-7).  The caller has just requested the
// NN + 4 1 KiB pieces of a later K step; they are issued one per NN MFMAs
// instead of as a burst in front of them (a piece costs its wave 100-200 cycles
// of issue, which the matrix cores otherwise spend idle).
template <bool FIRST, int NN>
__device__ __forceinline__ void mfma_high_rows(const bf16x8 (&fa)[4][2], const bf16x8 (&fb)[NN][2], f32x4 (&acc)[8][NN]) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int nn = 0; nn < NN; ++nn) {
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            f32x4 c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[m][0], fb[nn][0], FIRST ? zero : acc[4 + m][nn], 0, 0, 0);
            acc[4 + m][nn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[m][1], fb[nn][1], c, 0, 0, 0);
        }
#pragma unroll
    for (int g = 0; g < NN + 4; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // one load
        __builtin_amdgcn_sched_group_barrier(0x008, NN, 0);  // NN MFMAs (what is left follows the last load)
    }
}

// MAX_ONLY: the sampling form (a.tile_max), which keeps nothing but each
// query's best score per tile.
// NN: 16-query column tiles per wave; the query tile is TQ = 64 NN wide (256, 192 or 128),
// so that a batch a little above a multiple of 256 does not pay for 256 more columns.
// ROW_AUX: cache-policy bits of the row loads (2 = nt, streaming: a launch with ONE query
// tile over a shard beyond the caches reads every row byte once).
template <bool MAX_ONLY, int NN, int ROW_AUX = 0>
__global__ __launch_bounds__(kWideThreads, 2) void score_filter_wide_kernel(WideArgs a) {
    constexpr int TQ = 64 * NN;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: the load addresses build on it
    const int wm = wave >> 2, wn = wave & 3;
    // XCD-aware, bijective workgroup remap: ids that land on one XCD become consecutive.
    const int nwg = gridDim.x, xcd = blockIdx.x % 8, per = nwg / 8, rem = nwg % 8;
    const int sched = (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + blockIdx.x / 8;
    // Which (query tile, row list) pair the workgroup takes.  Pairs are numbered in blocks of
    // qgroup query tiles: inside a block the query tile runs fastest, so the workgroups of one
    // XCD (consecutive numbers) share each row list between qgroup query tiles -- a row tile
    // that one of them has brought into the XCD's L2 is a hit for the others -- while their
    // qgroup query tiles (512 KB each) stay resident in that L2 together.
    const int qblock = sched / (a.qgroup * a.n_lists);           // (the last block may hold fewer query tiles)
    const int qfirst = qblock * a.qgroup;
    const int qhere = a.n_qtiles - qfirst < a.qgroup ? a.n_qtiles - qfirst : a.qgroup;
    const int inner = sched - qblock * a.qgroup * a.n_lists;
    const int list = inner / qhere, qt = qfirst + inner % qhere;
    const int wgid = qt * a.n_lists + list;                      // its number in the key lists and counts
    long long my_tiles = a.n_tiles > list ? (a.n_tiles - list + a.n_lists - 1) / a.n_lists : 0;
    long long tile_step = a.n_lists, tile_first = list;
    if (MTR_TIMING(a) & 2) {  // contiguous ranges
        const long long tpr = (a.n_tiles + a.n_lists - 1) / a.n_lists;
        tile_first = list * tpr;
        tile_step = 1;
        my_tiles = a.n_tiles - tile_first < tpr ? a.n_tiles - tile_first : tpr;
        if (my_tiles < 0) my_tiles = 0;
    }
    const int ksteps = a.ld / 64;
    const long long row_bytes = (long long)a.ld * 2;
    const char* qbase = a.queries + (size_t)qt * TQ * row_bytes;
    const long long total = my_tiles * ksteps;

    // This lane's NN queries (accumulator column = lane & 15) and their floors.
    float thr[NN];
#pragma unroll
    for (int nn = 0; nn < NN; ++nn) {
        const int q = qt * TQ + wn * (16 * NN) + nn * 16 + (lane & 15);
        thr[nn] = __builtin_huge_valf();  // padding queries collect nothing
        if (!MAX_ONLY && q < a.b && !(MTR_TIMING(a) & 1)) {
            const unsigned int f = a.floor[q];
            thr[nn] = unorder_f32(f | (f == 0 ? 0x007fffffu : 0u));  // no floor: -inf
        }
    }
    // The floors must have arrived before the first tile load is requested: the
    // loop's counted waits assume that nothing else is in flight.
#pragma unroll
    for (int nn = 0; nn < NN; ++nn) asm volatile("" ::"v"(thr[nn]) : "memory");

    const unsigned int lane_off = lane_piece_offset(lane, (unsigned int)row_bytes);
    // load cursor: the next K step to request
    long long ld_tile = tile_first;
    int ld_ks = 0, ld_buf = 0;
    auto request_step = [&]() {
        char* dst = smem + ld_buf * kWideStepBytes;
        request_queries<NN>(qbase, row_bytes, ld_ks * 128, dst, wave, lane_off);
        request_rows<ROW_AUX>(a.rows, ld_tile * a.tile_stride * kWideTile + wm * 128, a.n - 1, row_bytes, ld_ks * 128,
                     dst + wm * kHalfBytes, wn, lane, lane_off);
        ld_buf ^= 1;
        if (++ld_ks == ksteps) { ld_ks = 0; ld_tile += tile_step; }
    };
    // Two K steps are always in flight, also past a workgroup's last step (the
    // cursor then runs on: rows are clamped to the matrix, and what lands goes to
    // a buffer nobody reads again), so that the loop requests without a branch.
    request_step();
    request_step();

    // acc[m][nn]: rows 16 m + 4 (lane / 16) + r of the wave's 128, query 16 nn + lane % 16 of its 16 NN
    f32x4 acc[8][NN];
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int nn = 0; nn < NN; ++nn)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[m][nn][r] = 0.f;   // (every tile's first K step overwrites them)

    // step 0 has landed for everybody; then group 1 falls one barrier behind
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NN + 4) : "memory");
    barrier_keeping_loads();
    if (wm == 1) barrier_keeping_loads();

    // Each query's scores of a tile sit in eight lanes of the workgroup (wm = 0, 1 x
    // lane / 16 = 0..3); each of them appends to a list of its own and counts in a
    // register.
    const int sub = wm * 4 + (lane >> 4);
    unsigned long long* my_keys = a.keys + ((size_t)wgid * TQ * kWideSubs + sub) * a.cap;
    int cnt[NN];
#pragma unroll
    for (int nn = 0; nn < NN; ++nn) cnt[nn] = 0;
    const int cap_mask = a.cap - 1;  // cap is a power of two
    long long tile = tile_first;
    int ks = 0;
    for (long long s = 0; s < total; ++s) {
        const char* buf = smem + (int)(s & 1) * kWideStepBytes;
        const char* sa = buf + wm * kHalfBytes;
        const char* sb = buf + 2 * kHalfBytes;
        const int qrow = wn * (16 * NN);
        // this group's rows of step s (its queries were waited for before the previous barrier)
        if (s > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NN + 4) : "memory");
        barrier_keeping_loads();                                   // first barrier of the step
        // Fragment of a 16 x 32 block: lane l holds elements 8 (l / 16) .. +7 of row l % 16.
        bf16x8 fb[NN][2], fa[4][2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int nn = 0; nn < NN; ++nn) fb[nn][kk] = read_frag(sb, qrow + nn * 16 + (lane & 15), 4 * kk + (lane >> 4));
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int m = 0; m < 4; ++m) fa[m][kk] = read_frag(sa, m * 16 + (lane & 15), 4 * kk + (lane >> 4));
        barrier_keeping_loads();                                   // second barrier
        // A tile's first K step starts its sums from the constant zero, which
        // spares the epilogue 128 register clears per tile.
        const bool first = ks == 0;
        const bool tile_done = ks + 1 == ksteps;
        __builtin_amdgcn_s_setprio(1);
        if (first) mfma_low_rows<true, NN>(fa, fb, acc, sa, lane);
        else mfma_low_rows<false, NN>(fa, fb, acc, sa, lane);
        __builtin_amdgcn_s_setprio(0);
        // the queries of step s + 1: the older NN of the NN + 4 loads in flight
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        barrier_keeping_loads();                                   // third barrier
        // The buffers of step s are free: step s + 2 is requested, its pieces spread
        // between the MFMAs.  (One branch per form of the MFMAs and nothing else: with
        // a third, request-free copy for a tile's last step the compiler spilled 200
        // registers.)
        __builtin_amdgcn_s_setprio(1);   // the priority change bounds what the scheduler may interleave: requests go inside it
        if (first) {
            request_step();
            mfma_high_rows<true, NN>(fa, fb, acc);
        } else {
            request_step();
            mfma_high_rows<false, NN>(fa, fb, acc);
        }
        __builtin_amdgcn_s_setprio(0);

        if (tile_done) {
            // The tile's scores are complete.  Rows past the end of the matrix (copies
            // of the last row, in the last tile only) are treated like any other: they
            // cannot raise a maximum, and the selection kernel drops their keys.
            const long long ptile = tile * a.tile_stride;
            if constexpr (MAX_ONLY) {
#pragma unroll
                for (int nn = 0; nn < NN; ++nn) {
                    float mx = -__builtin_huge_valf();
#pragma unroll
                    for (int m = 0; m < 8; ++m)
#pragma unroll
                        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, acc[m][nn][r]);
                    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                    if (lane < 16)
                        atomicMax(a.tile_max + ((size_t)qt * TQ + wn * (16 * NN) + nn * 16 + lane) * a.n_tiles + tile, order_f32(mx));
                }
            } else {
                const int row_base = (int)(ptile * kWideTile) + wm * 128 + 4 * (lane >> 4);
                if (!(MTR_TIMING(a) & 4))
#pragma unroll
                for (int nn = 0; nn < NN; ++nn) {
                    const int ql = wn * (16 * NN) + nn * 16 + (lane & 15);
                    unsigned long long* qkeys = my_keys + (size_t)ql * kWideSubs * a.cap;
                    // Three scores per test (v_max3): a branch costs more than the compare,
                    // and with some thousand scores per query reaching the floor, about one
                    // group in three holds one in some lane.  Inside the branch (one lane
                    // active as a rule, some 15 times per wave and tile) the lane's first hit
                    // is picked with selects and stored once; only a lane with two or three
                    // hits in one group, which is rare, goes through a second branch.
                    using u32x2 = __attribute__((ext_vector_type(2))) unsigned int;
#pragma unroll
                    for (int g = 0; g < 32; g += 3) {
                        const float v0 = acc[g >> 2][nn][g & 3];
                        const float v1 = g + 1 < 32 ? acc[(g + 1) >> 2][nn][(g + 1) & 3] : -__builtin_huge_valf();
                        const float v2 = g + 2 < 32 ? acc[(g + 2) >> 2][nn][(g + 2) & 3] : -__builtin_huge_valf();
                        const float mx = fmaxf(fmaxf(v0, v1), v2);
                        if (__builtin_expect(mx >= thr[nn], 0)) {
                            // (row, raw score bits), wrapping around in a full list: the count
                            // tells the selection kernel that the list is incomplete.
                            const int r0 = (g >> 2) * 16 + (g & 3), r1 = ((g + 1) >> 2) * 16 + ((g + 1) & 3),
                                      r2 = ((g + 2) >> 2) * 16 + ((g + 2) & 3);
                            // (the last group holds two scores: its padding must not pass a floor of -inf)
                            const bool h0 = v0 >= thr[nn], h1 = g + 1 < 32 && v1 >= thr[nn], h2 = g + 2 < 32 && v2 >= thr[nn];
                            const float vs = h0 ? v0 : h1 ? v1 : v2;
                            const int rs = h0 ? r0 : h1 ? r1 : r2;
                            *reinterpret_cast<u32x2*>(qkeys + (cnt[nn] & cap_mask)) =
                                u32x2{(unsigned int)(row_base + rs), __float_as_uint(vs)};
                            ++cnt[nn];
                            if (__builtin_expect((h0 && (h1 || h2)) || (h1 && h2), 0)) {
                                if (h0 && h1) {
                                    *reinterpret_cast<u32x2*>(qkeys + (cnt[nn] & cap_mask)) =
                                        u32x2{(unsigned int)(row_base + r1), __float_as_uint(v1)};
                                    ++cnt[nn];
                                }
                                if ((h0 || h1) && h2) {
                                    *reinterpret_cast<u32x2*>(qkeys + (cnt[nn] & cap_mask)) =
                                        u32x2{(unsigned int)(row_base + r2), __float_as_uint(v2)};
                                    ++cnt[nn];
                                }
                            }
                        }
                    }
                }
            }
            ks = 0;
            tile += tile_step;
        } else {
            ++ks;
        }
    }
    if (wm == 0) barrier_keeping_loads();   // group 0 catches up with group 1's last barrier
    // the two requests past the last step must have landed before the LDS is given back
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (!MAX_ONLY) {
#pragma unroll
        for (int nn = 0; nn < NN; ++nn) {
            const int ql = wn * (16 * NN) + nn * 16 + (lane & 15);
            a.counts[((size_t)wgid * TQ + ql) * kWideSubs + sub] = (unsigned int)cnt[nn];
        }
    }
}

// Every floor back to "none" when *flag is set: the general kernel that then
// recomputes the batch is a different instruction mix, whose k-th best scores
// may sit an ulp below floors derived from this kernel's arithmetic.
__global__ void reset_floors_if_kernel(unsigned int* floors, int words, const int* flag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (*flag != 0 && i < words) floors[i] = 0u;
}

}  // namespace

int plan_wide(int dtype, int64_t n, int ld, int b, int k, int expected_hits, int num_cus, WidePlan* plan) {
    if (dtype != 2 || ld % 64 != 0 || ld < 128 || b < 1 || k < 1 || k > 256 || n < 1 || n > 0x7fffffffLL) return -1;
    // The fewest query tiles of at most 256, as narrow (128, 192 or 256) as they can be.
    plan->n_qtiles = (b + kWideTile - 1) / kWideTile;
    const int per_tile = (b + plan->n_qtiles - 1) / plan->n_qtiles;
    plan->tq = per_tile <= 128 ? 128 : per_tile <= 192 ? 192 : 256;
    if (tuning().wide_tile_queries == 128 || tuning().wide_tile_queries == 192 || tuning().wide_tile_queries == 256) {
        plan->tq = tuning().wide_tile_queries;
        plan->n_qtiles = (b + plan->tq - 1) / plan->tq;
    }
    plan->n_tiles = (n + kWideTile - 1) / kWideTile;
    long long lists = num_cus / plan->n_qtiles;  // one workgroup per CU
    if (lists > plan->n_tiles) lists = plan->n_tiles;
    if (lists < 1) lists = 1;
    plan->n_lists = (int)lists;
    plan->grid = plan->n_qtiles * plan->n_lists;
    plan->lds = kWideLds;
    // Slots per list: twice its even share of the hits the floor is expected to
    // let through, a power of two between 128 and 8192.
    long long want = 2LL * expected_hits / ((long long)plan->n_lists * kWideSubs);
    int cap = 128;
    while (cap < want && cap < 8192) cap *= 2;
    // ... and at most 4 GiB of lists in all (large shards have long sample strides;
    // their lists are long enough for a smaller margin to do)
    while (cap > 128 && (size_t)plan->grid * plan->tq * kWideSubs * cap * 8 > ((size_t)4 << 30)) cap /= 2;
    plan->cap = cap;
    plan->key_bytes = (size_t)plan->grid * plan->tq * kWideSubs * cap * 8;
    plan->count_words = (size_t)plan->grid * plan->tq * kWideSubs;
    return 0;
}

hipError_t launch_score_wide(const void* rows, int64_t n, int ld, const void* queries, int b, const WidePlan& plan,
                             const unsigned int* floors, unsigned long long* keys, unsigned int* counts,
                             int tile_stride, unsigned int* tile_max, hipStream_t stream) {
    WideArgs a;
    a.rows = static_cast<const char*>(rows);
    a.n = n;
    a.ld = ld;
    a.queries = static_cast<const char*>(queries);
    a.b = b;
    a.n_lists = plan.n_lists;
    a.n_qtiles = plan.n_qtiles;
    a.qgroup = tuning().wide_qgroup > 0 ? tuning().wide_qgroup : 4;
    if (a.qgroup > plan.n_qtiles) a.qgroup = plan.n_qtiles;
    a.n_tiles = plan.n_tiles;
    a.tile_stride = tile_stride < 1 ? 1 : tile_stride;
    a.tile_max = tile_max;
    a.floor = floors;
    a.keys = keys;
    a.counts = counts;
    a.cap = plan.cap;
    a.debug = 0;
#ifdef MTR_TIMING_SWITCHES
    a.debug = tuning().timing_wide;
#endif
    // the last tile of the launch must exist in the matrix
    if ((plan.n_tiles - 1) * a.tile_stride * kWideTile >= n) return hipErrorInvalidValue;
    auto launch = [&](auto kernel) -> hipError_t {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)plan.lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kernel, dim3(plan.grid), dim3(kWideThreads), plan.lds, stream, a);
        return hipGetLastError();
    };
    // streaming row loads: one query tile, full launch, rows beyond the Infinity Cache
    const int stream_switch = tuning().wide_stream_rows;
    const bool stream_rows = tile_max == nullptr && (stream_switch >= 0 ? stream_switch != 0
                                                                        : plan.n_qtiles == 1 && (double)n * ld * 2 > 192.0 * 1024 * 1024);
    if (stream_rows) {
        if (plan.tq == 128) return launch(&score_filter_wide_kernel<false, 2, 2>);
        if (plan.tq == 192) return launch(&score_filter_wide_kernel<false, 3, 2>);
        if (plan.tq == 256) return launch(&score_filter_wide_kernel<false, 4, 2>);
    }
    if (plan.tq == 128) return tile_max != nullptr ? launch(&score_filter_wide_kernel<true, 2>) : launch(&score_filter_wide_kernel<false, 2>);
    if (plan.tq == 192) return tile_max != nullptr ? launch(&score_filter_wide_kernel<true, 3>) : launch(&score_filter_wide_kernel<false, 3>);
    if (plan.tq != 256) return hipErrorInvalidValue;
    if (tile_max != nullptr) return launch(&score_filter_wide_kernel<true, 4>);
    return launch(&score_filter_wide_kernel<false, 4>);
}

hipError_t launch_reset_floors_if(unsigned int* floors, int words, const int* flag, hipStream_t stream) {
    if (words < 1) return hipSuccess;
    hipLaunchKernelGGL(reset_floors_if_kernel, dim3((words + 255) / 256), dim3(256), 0, stream, floors, words, flag);
    return hipGetLastError();
}

}  // namespace mtr
