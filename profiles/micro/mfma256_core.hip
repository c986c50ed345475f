// Prototype of a 256 x 256 bf16 score tile on the matrix cores with a
// half-tile load ring (research code, not part of the library): eight waves
// (2 along the rows x 4 along the queries), each 128 rows x 64 queries as 4 x 2
// MFMA tiles of 32 x 32, K steps of 64.  LDS holds two K steps, each as four
// half tiles of 128 rows x 128 bytes (rows 0-127, rows 128-255, queries 0-127,
// queries 128-255).  A half tile is refilled with the data of two K steps
// later as soon as every wave has moved its fragments of it to registers, so
// loads have more than a whole K step to land and are issued in two bursts per
// step.  Output: each query's best score per row tile (enough to check the
// arithmetic against a plain kernel).
// build: hipcc --offload-arch=gfx950 -O3 -o profiles/micro/mfma256_core profiles/micro/mfma256_core.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef unsigned short bf16_t;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

constexpr int kHalfBytes = 128 * 128;      // 128 rows x 128 bytes
constexpr int kStepBytes = 4 * kHalfBytes;  // rows lo, rows hi, queries lo, queries hi
constexpr int kThreads = 512;

__device__ __forceinline__ unsigned int order_f32(float f) {
    const unsigned int u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// One half tile: 16 wave instructions of 1 KiB (8 rows x 128 B), two per wave.
__device__ __forceinline__ void stage_half(const char* base, long long row0, long long row_max, long long row_bytes,
                                           int k0_bytes, char* lds, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int t = wave * 2 + i;
        const int r = t * 8 + (lane >> 3);
        long long row = row0 + r;
        row = row < row_max ? row : row_max;
        const int chunk = (lane & 7) ^ (lane >> 3);
        const char* src = base + row * row_bytes + k0_bytes + chunk * 16;
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(lds + t * 1024), 16, 0, 0);
    }
}

__device__ __forceinline__ bf16x8 read_frag(const char* lds, int row, int chunk) {
    return *reinterpret_cast<const bf16x8*>(lds + row * 128 + ((chunk ^ (row & 7)) << 4));
}

__device__ __forceinline__ void barrier_keeping_loads() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// grid = n_qtiles * n_ranges; workgroup (qt, range) sweeps row tiles [range*tpr, ...) against query tile qt.
__global__ __launch_bounds__(kThreads, 2) void core_kernel(const char* rows, long long n, int ld, const char* queries,
                                                           int n_ranges, long long tiles_per_range,
                                                           unsigned int* tile_max /* [n_q][n_tiles] */, int mode) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned int* best = reinterpret_cast<unsigned int*>(smem + 2 * kStepBytes);  // [256]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int nwg = gridDim.x, xcd = blockIdx.x % 8, per = nwg / 8, rem = nwg % 8;
    const int wgid = (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + blockIdx.x / 8;
    const int qt = wgid / n_ranges, range = wgid % n_ranges;
    const long long n_tiles = (n + 255) / 256;
    const long long tile_lo = (long long)range * tiles_per_range;
    long long tile_hi = tile_lo + tiles_per_range;
    if (tile_hi > n_tiles) tile_hi = n_tiles;
    const int ksteps = ld / 64;
    const long long row_bytes = (long long)ld * 2;
    const char* qbase = queries + (size_t)qt * 256 * row_bytes;
    const long long total = (tile_hi > tile_lo ? tile_hi - tile_lo : 0) * ksteps;
    if (tid < 256) best[tid] = 0u;

    // load cursor: next K step whose halves are to be requested
    long long ld_tile = tile_lo;
    int ld_ks = 0, ld_buf = 0;
    auto request_queries = [&]() {  // both query halves of the cursor's step
        char* dst = smem + ld_buf * kStepBytes;
        stage_half(qbase, 0, 255, row_bytes, ld_ks * 128, dst + 2 * kHalfBytes, wave, lane);
        stage_half(qbase, 128, 255, row_bytes, ld_ks * 128, dst + 3 * kHalfBytes, wave, lane);
    };
    auto request_rows = [&]() {  // both row halves, then advance the cursor
        char* dst = smem + ld_buf * kStepBytes;
        stage_half(rows, ld_tile * 256, n - 1, row_bytes, ld_ks * 128, dst, wave, lane);
        stage_half(rows, ld_tile * 256 + 128, n - 1, row_bytes, ld_ks * 128, dst + kHalfBytes, wave, lane);
        ld_buf ^= 1;
        if (++ld_ks == ksteps) { ld_ks = 0; ++ld_tile; }
    };
    if (total > 0) { request_queries(); request_rows(); }
    if (total > 1) { request_queries(); request_rows(); }

    f32x16 acc[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int nn = 0; nn < 2; ++nn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][nn][r] = 0.f;

    long long tile = tile_lo;
    int ks = 0;
    for (long long s = 0; s < total; ++s) {
        const char* buf = smem + (int)(s & 1) * kStepBytes;
        const char* sa = buf + wm * kHalfBytes;                         // this wave's 128 rows
        const char* sb = buf + (2 + (wn >> 1)) * kHalfBytes;            // the half holding its 64 queries
        const int qrow = (wn & 1) * 64;
        // step s has landed when at most the 8 loads of step s + 1 are outstanding
        if (s + 1 < total) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        barrier_keeping_loads();

        bf16x8 fb[2][4], fa[2][4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int nn = 0; nn < 2; ++nn) fb[nn][kk] = read_frag(sb, qrow + nn * 32 + (lane & 31), 2 * kk + (lane >> 5));
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int m = 0; m < 2; ++m) fa[m][kk] = read_frag(sa, m * 32 + (lane & 31), 2 * kk + (lane >> 5));
        barrier_keeping_loads();  // every wave holds its query fragments: the query halves are free
        const bool more = s + 2 < total && !(mode & 2);
        if (more) request_queries();
        if (!(mode & 1)) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int nn = 0; nn < 2; ++nn)
                        acc[m][nn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[m][kk], fb[nn][kk], acc[m][nn], 0, 0, 0);
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int m = 0; m < 2; ++m) fa[m][kk] = read_frag(sa, 64 + m * 32 + (lane & 31), 2 * kk + (lane >> 5));
        barrier_keeping_loads();  // ... and its row fragments: the row halves are free
        if (more) request_rows();
        if (!(mode & 1)) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int nn = 0; nn < 2; ++nn)
                        acc[2 + m][nn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[m][kk], fb[nn][kk], acc[2 + m][nn], 0, 0, 0);
        }

        if (ks + 1 == ksteps) {
#pragma unroll
            for (int nn = 0; nn < 2; ++nn) {
                float mx = -__builtin_huge_valf();
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        mx = fmaxf(mx, acc[m][nn][r]);
                        acc[m][nn][r] = 0.f;
                    }
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                if (lane < 32) atomicMax(&best[wn * 64 + nn * 32 + lane], order_f32(mx));
            }
            barrier_keeping_loads();
            if (tid < 256) {
                tile_max[((size_t)qt * 256 + tid) * n_tiles + tile] = best[tid];
                best[tid] = 0u;
            }
            ks = 0;
            ++tile;
        } else {
            ++ks;
        }
    }
}

// The same tile with its two wave groups (waves 0-3: rows 0-127, waves 4-7: rows
// 128-255) running ONE BARRIER APART, so that one group is on the matrix cores
// while the other reads LDS.  Every wave executes the same barrier sequence,
// three per K step; group 1 executes one extra barrier first and group 0 one
// extra at the end.  In group-0 barrier slots, step s: slot 3s (then it reads
// its query + low row fragments), 3s+1 (MFMA low, reads high rows), 3s+2 (MFMA
// high); group 1 does the same at 3s+1, 3s+2, 3s+3.
//   landed:  the query halves of step s+1 are loaded by all eight waves; each
//            waits for its part before its LAST barrier of step s, i.e. before
//            slot 3s+3 at the latest, and the first reader starts after 3s+3.
//            A row half is loaded and read by ONE group: its waves wait before
//            their first barrier of step s+1.
//   refill:  after its last barrier of step s a wave requests the queries and its
//            group's rows of step s+2 into the buffers step s used: group 0
//            after slot 3s+2, when group 1 finished its query reads (before its
//            second barrier = 3s+2); a group's row reads end before its own last
//            barrier.
__global__ __launch_bounds__(kThreads, 2) void pingpong_kernel(const char* rows, long long n, int ld, const char* queries,
                                                               int n_ranges, long long tiles_per_range,
                                                               unsigned int* tile_max, int mode) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned int* best = reinterpret_cast<unsigned int*>(smem + 2 * kStepBytes);  // [256]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int nwg = gridDim.x, xcd = blockIdx.x % 8, per = nwg / 8, rem = nwg % 8;
    const int wgid = (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + blockIdx.x / 8;
    const int qt = wgid / n_ranges, range = wgid % n_ranges;
    const long long n_tiles = (n + 255) / 256;
    const long long tile_lo = (long long)range * tiles_per_range;
    long long tile_hi = tile_lo + tiles_per_range;
    if (tile_hi > n_tiles) tile_hi = n_tiles;
    const int ksteps = ld / 64;
    const long long row_bytes = (long long)ld * 2;
    const char* qbase = queries + (size_t)qt * 256 * row_bytes;
    const long long total = (tile_hi > tile_lo ? tile_hi - tile_lo : 0) * ksteps;
    if (tid < 256) best[tid] = 0u;

    long long ld_tile = tile_lo;
    int ld_ks = 0, ld_buf = 0;
    // four loads per wave: this wave's share of both query halves ...
    auto request_queries = [&]() {
        char* dst = smem + ld_buf * kStepBytes;
        stage_half(qbase, 0, 255, row_bytes, ld_ks * 128, dst + 2 * kHalfBytes, wave, lane);
        stage_half(qbase, 128, 255, row_bytes, ld_ks * 128, dst + 3 * kHalfBytes, wave, lane);
    };
    // ... and four more: its share of ITS GROUP's row half (16 instructions over 4 waves)
    auto request_rows = [&]() {
        char* dst = smem + ld_buf * kStepBytes + wm * kHalfBytes;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int t = (wave & 3) * 4 + i;
            const int r = t * 8 + (lane >> 3);
            long long row = ld_tile * 256 + wm * 128 + r;
            row = row < n - 1 ? row : n - 1;
            const int chunk = (lane & 7) ^ (lane >> 3);
            const char* src = rows + row * row_bytes + ld_ks * 128 + chunk * 16;
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(dst + t * 1024), 16, 0, 0);
        }
        ld_buf ^= 1;
        if (++ld_ks == ksteps) { ld_ks = 0; ++ld_tile; }
    };
    if (total > 0) { request_queries(); request_rows(); }
    if (total > 1) { request_queries(); request_rows(); }

    f32x16 acc[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int nn = 0; nn < 2; ++nn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][nn][r] = 0.f;

    // step 0 has landed for everybody; then group 1 falls one barrier behind
    if (total > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    barrier_keeping_loads();
    if (wm == 1) barrier_keeping_loads();

    long long tile = tile_lo;
    int ks = 0;
    long long flush_tile = -1;  // a finished tile whose maxima group 1 still has to write out
    for (long long s = 0; s < total; ++s) {
        const char* buf = smem + (int)(s & 1) * kStepBytes;
        const char* sa = buf + wm * kHalfBytes;
        const char* sb = buf + (2 + (wn >> 1)) * kHalfBytes;
        const int qrow = (wn & 1) * 64;
        // this group's rows of step s (its queries were waited for before the previous barrier)
        if (s > 0) {
            if (s + 1 < total) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        barrier_keeping_loads();                                   // first barrier of the step
        if (flush_tile >= 0 && wm == 1) {                            // both groups' maxima of that tile are in
            tile_max[((size_t)qt * 256 + (tid - 256)) * n_tiles + flush_tile] = best[tid - 256];
            best[tid - 256] = 0u;
        }
        flush_tile = -1;
        bf16x8 fb[2][4], fa[2][4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int nn = 0; nn < 2; ++nn) fb[nn][kk] = read_frag(sb, qrow + nn * 32 + (lane & 31), 2 * kk + (lane >> 5));
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int m = 0; m < 2; ++m) fa[m][kk] = read_frag(sa, m * 32 + (lane & 31), 2 * kk + (lane >> 5));
        barrier_keeping_loads();                                   // second barrier
        if (!(mode & 1)) {
            if (mode & 4) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int nn = 0; nn < 2; ++nn)
                        acc[m][nn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[m][kk], fb[nn][kk], acc[m][nn], 0, 0, 0);
            if (mode & 4) __builtin_amdgcn_s_setprio(0);
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int m = 0; m < 2; ++m) fa[m][kk] = read_frag(sa, 64 + m * 32 + (lane & 31), 2 * kk + (lane >> 5));
        // queries of step s + 1: only this step's... no later loads exist yet beyond rows(s+1)
        if (s + 1 < total) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        barrier_keeping_loads();                                   // third barrier
        if (s + 2 < total && !(mode & 2)) {
            request_queries();
            request_rows();
        }
        if (!(mode & 1)) {
            if (mode & 4) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int nn = 0; nn < 2; ++nn)
                        acc[2 + m][nn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[m][kk], fb[nn][kk], acc[2 + m][nn], 0, 0, 0);
            if (mode & 4) __builtin_amdgcn_s_setprio(0);
        }
        if (ks + 1 == ksteps) {
#pragma unroll
            for (int nn = 0; nn < 2; ++nn) {
                float mx = -__builtin_huge_valf();
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        mx = fmaxf(mx, acc[m][nn][r]);
                        acc[m][nn][r] = 0.f;
                    }
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                if (lane < 32) atomicMax(&best[wn * 64 + nn * 32 + lane], order_f32(mx));
            }
            flush_tile = tile;
            ks = 0;
            ++tile;
        } else {
            ++ks;
        }
    }
    if (wm == 0) barrier_keeping_loads();   // group 0 catches up with group 1's last barrier
    barrier_keeping_loads();                // every wave's maxima are in LDS
    if (flush_tile >= 0 && wm == 1) {
        tile_max[((size_t)qt * 256 + (tid - 256)) * n_tiles + flush_tile] = best[tid - 256];
    }
}

// The staggered form with v_mfma_f32_16x16x32_bf16 (8 x 4 tiles of 16 x 16 per
// wave, K blocks of 32): the same cycles per flop and LDS reads as the 32 x 32
// x 16 form; what may differ is the clock the chip holds under the load.
__global__ __launch_bounds__(kThreads, 2) void pingpong16_kernel(const char* rows, long long n, int ld, const char* queries,
                                                               int n_ranges, long long tiles_per_range,
                                                               unsigned int* tile_max, int mode) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned int* best = reinterpret_cast<unsigned int*>(smem + 2 * kStepBytes);  // [256]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int nwg = gridDim.x, xcd = blockIdx.x % 8, per = nwg / 8, rem = nwg % 8;
    const int wgid = (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + blockIdx.x / 8;
    const int qt = wgid / n_ranges, range = wgid % n_ranges;
    const long long n_tiles = (n + 255) / 256;
    const long long tile_lo = (long long)range * tiles_per_range;
    long long tile_hi = tile_lo + tiles_per_range;
    if (tile_hi > n_tiles) tile_hi = n_tiles;
    const int ksteps = ld / 64;
    const long long row_bytes = (long long)ld * 2;
    const char* qbase = queries + (size_t)qt * 256 * row_bytes;
    const long long total = (tile_hi > tile_lo ? tile_hi - tile_lo : 0) * ksteps;
    if (tid < 256) best[tid] = 0u;

    long long ld_tile = tile_lo;
    int ld_ks = 0, ld_buf = 0;
    // four loads per wave: this wave's share of both query halves ...
    auto request_queries = [&]() {
        char* dst = smem + ld_buf * kStepBytes;
        stage_half(qbase, 0, 255, row_bytes, ld_ks * 128, dst + 2 * kHalfBytes, wave, lane);
        stage_half(qbase, 128, 255, row_bytes, ld_ks * 128, dst + 3 * kHalfBytes, wave, lane);
    };
    // ... and four more: its share of ITS GROUP's row half (16 instructions over 4 waves)
    auto request_rows = [&]() {
        char* dst = smem + ld_buf * kStepBytes + wm * kHalfBytes;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int t = (wave & 3) * 4 + i;
            const int r = t * 8 + (lane >> 3);
            long long row = ld_tile * 256 + wm * 128 + r;
            row = row < n - 1 ? row : n - 1;
            const int chunk = (lane & 7) ^ (lane >> 3);
            const char* src = rows + row * row_bytes + ld_ks * 128 + chunk * 16;
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(dst + t * 1024), 16, 0, 0);
        }
        ld_buf ^= 1;
        if (++ld_ks == ksteps) { ld_ks = 0; ++ld_tile; }
    };
    if (total > 0) { request_queries(); request_rows(); }
    if (total > 1) { request_queries(); request_rows(); }

    f32x4 acc[8][4];
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int nn = 0; nn < 4; ++nn)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[m][nn][r] = 0.f;

    // step 0 has landed for everybody; then group 1 falls one barrier behind
    if (total > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    barrier_keeping_loads();
    if (wm == 1) barrier_keeping_loads();

    long long tile = tile_lo;
    int ks = 0;
    long long flush_tile = -1;  // a finished tile whose maxima group 1 still has to write out
    for (long long s = 0; s < total; ++s) {
        const char* buf = smem + (int)(s & 1) * kStepBytes;
        const char* sa = buf + wm * kHalfBytes;
        const char* sb = buf + (2 + (wn >> 1)) * kHalfBytes;
        const int qrow = (wn & 1) * 64;
        // this group's rows of step s (its queries were waited for before the previous barrier)
        if (s > 0) {
            if (s + 1 < total) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        barrier_keeping_loads();                                   // first barrier of the step
        if (flush_tile >= 0 && wm == 1) {                            // both groups' maxima of that tile are in
            tile_max[((size_t)qt * 256 + (tid - 256)) * n_tiles + flush_tile] = best[tid - 256];
            best[tid - 256] = 0u;
        }
        flush_tile = -1;
        bf16x8 fb[4][2], fa[4][2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int nn = 0; nn < 4; ++nn) fb[nn][kk] = read_frag(sb, qrow + nn * 16 + (lane & 15), 4 * kk + (lane >> 4));
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int m = 0; m < 4; ++m) fa[m][kk] = read_frag(sa, m * 16 + (lane & 15), 4 * kk + (lane >> 4));
        barrier_keeping_loads();                                   // second barrier
        if (!(mode & 1)) {
            if (mode & 4) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int nn = 0; nn < 4; ++nn)
                        acc[m][nn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[m][kk], fb[nn][kk], acc[m][nn], 0, 0, 0);
            if (mode & 4) __builtin_amdgcn_s_setprio(0);
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int m = 0; m < 4; ++m) fa[m][kk] = read_frag(sa, 64 + m * 16 + (lane & 15), 4 * kk + (lane >> 4));
        // queries of step s + 1: only this step's... no later loads exist yet beyond rows(s+1)
        if (s + 1 < total) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        barrier_keeping_loads();                                   // third barrier
        if (s + 2 < total && !(mode & 2)) {
            request_queries();
            request_rows();
        }
        if (!(mode & 1)) {
            if (mode & 4) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int nn = 0; nn < 4; ++nn)
                        acc[4 + m][nn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[m][kk], fb[nn][kk], acc[4 + m][nn], 0, 0, 0);
            if (mode & 4) __builtin_amdgcn_s_setprio(0);
        }
        if (ks + 1 == ksteps) {
#pragma unroll
            for (int nn = 0; nn < 4; ++nn) {
                float mx = -__builtin_huge_valf();
#pragma unroll
                for (int m = 0; m < 8; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        mx = fmaxf(mx, acc[m][nn][r]);
                        acc[m][nn][r] = 0.f;
                    }
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                if (lane < 16) atomicMax(&best[wn * 64 + nn * 16 + lane], order_f32(mx));
            }
            flush_tile = tile;
            ks = 0;
            ++tile;
        } else {
            ++ks;
        }
    }
    if (wm == 0) barrier_keeping_loads();   // group 0 catches up with group 1's last barrier
    barrier_keeping_loads();                // every wave's maxima are in LDS
    if (flush_tile >= 0 && wm == 1) {
        tile_max[((size_t)qt * 256 + (tid - 256)) * n_tiles + flush_tile] = best[tid - 256];
    }
}

// The 16x16x32 form with the K-step requests issued unconditionally and spread
// between the MFMAs of the step's second half by sched_group_barrier.
__global__ __launch_bounds__(kThreads, 2) void pingpong16i_kernel(const char* rows, long long n, int ld, const char* queries,
                                                               int n_ranges, long long tiles_per_range,
                                                               unsigned int* tile_max, int mode) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned int* best = reinterpret_cast<unsigned int*>(smem + 2 * kStepBytes);  // [256]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int nwg = gridDim.x, xcd = blockIdx.x % 8, per = nwg / 8, rem = nwg % 8;
    const int wgid = (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + blockIdx.x / 8;
    const int qt = wgid / n_ranges, range = wgid % n_ranges;
    const long long n_tiles = (n + 255) / 256;
    const long long tile_lo = (long long)range * tiles_per_range;
    long long tile_hi = tile_lo + tiles_per_range;
    if (tile_hi > n_tiles) tile_hi = n_tiles;
    const int ksteps = ld / 64;
    const long long row_bytes = (long long)ld * 2;
    const char* qbase = queries + (size_t)qt * 256 * row_bytes;
    const long long total = (tile_hi > tile_lo ? tile_hi - tile_lo : 0) * ksteps;
    if (tid < 256) best[tid] = 0u;

    long long ld_tile = tile_lo;
    int ld_ks = 0, ld_buf = 0;
    // four loads per wave: this wave's share of both query halves ...
    auto request_queries = [&]() {
        char* dst = smem + ld_buf * kStepBytes;
        stage_half(qbase, 0, 255, row_bytes, ld_ks * 128, dst + 2 * kHalfBytes, wave, lane);
        stage_half(qbase, 128, 255, row_bytes, ld_ks * 128, dst + 3 * kHalfBytes, wave, lane);
    };
    // ... and four more: its share of ITS GROUP's row half (16 instructions over 4 waves)
    auto request_rows = [&]() {
        char* dst = smem + ld_buf * kStepBytes + wm * kHalfBytes;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int t = (wave & 3) * 4 + i;
            const int r = t * 8 + (lane >> 3);
            long long row = ld_tile * 256 + wm * 128 + r;
            row = row < n - 1 ? row : n - 1;
            const int chunk = (lane & 7) ^ (lane >> 3);
            const char* src = rows + row * row_bytes + ld_ks * 128 + chunk * 16;
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(dst + t * 1024), 16, 0, 0);
        }
        ld_buf ^= 1;
        if (++ld_ks == ksteps) { ld_ks = 0; ++ld_tile; }
    };
    if (total > 0) { request_queries(); request_rows(); }
    if (total > 1) { request_queries(); request_rows(); }

    f32x4 acc[8][4];
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int nn = 0; nn < 4; ++nn)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[m][nn][r] = 0.f;

    // step 0 has landed for everybody; then group 1 falls one barrier behind
    if (total > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    barrier_keeping_loads();
    if (wm == 1) barrier_keeping_loads();

    long long tile = tile_lo;
    int ks = 0;
    long long flush_tile = -1;  // a finished tile whose maxima group 1 still has to write out
    for (long long s = 0; s < total; ++s) {
        const char* buf = smem + (int)(s & 1) * kStepBytes;
        const char* sa = buf + wm * kHalfBytes;
        const char* sb = buf + (2 + (wn >> 1)) * kHalfBytes;
        const int qrow = (wn & 1) * 64;
        // this group's rows of step s (its queries were waited for before the previous barrier)
        if (s > 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        barrier_keeping_loads();                                   // first barrier of the step
        if (flush_tile >= 0 && wm == 1) {                            // both groups' maxima of that tile are in
            tile_max[((size_t)qt * 256 + (tid - 256)) * n_tiles + flush_tile] = best[tid - 256];
            best[tid - 256] = 0u;
        }
        flush_tile = -1;
        bf16x8 fb[4][2], fa[4][2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int nn = 0; nn < 4; ++nn) fb[nn][kk] = read_frag(sb, qrow + nn * 16 + (lane & 15), 4 * kk + (lane >> 4));
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int m = 0; m < 4; ++m) fa[m][kk] = read_frag(sa, m * 16 + (lane & 15), 4 * kk + (lane >> 4));
        barrier_keeping_loads();                                   // second barrier
        // Low rows: the eight MFMAs of row tile m, then the two reads that replace its
        // fragments with those of the high rows (they land while the next tiles multiply).
        if (mode & 4) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
#pragma unroll
            for (int nn = 0; nn < 4; ++nn)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
                    acc[m][nn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[m][kk], fb[nn][kk], acc[m][nn], 0, 0, 0);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) fa[m][kk] = read_frag(sa, 64 + m * 16 + (lane & 15), 4 * kk + (lane >> 4));
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);  // eight MFMAs
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);  // two LDS reads
        }
        if (mode & 4) __builtin_amdgcn_s_setprio(0);
        // queries of step s + 1: only this step's... no later loads exist yet beyond rows(s+1)
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        barrier_keeping_loads();                                   // third barrier
        // Requests without a branch (past the last step they re-read clamped rows into a
        // buffer nobody reads again), spread between the MFMAs: one 1 KiB piece per four.
        request_queries();
        request_rows();
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int nn = 0; nn < 4; ++nn)
                    acc[4 + m][nn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[m][kk], fb[nn][kk], acc[4 + m][nn], 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // one VMEM read
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);  // four MFMAs
        }
        if (ks + 1 == ksteps) {
#pragma unroll
            for (int nn = 0; nn < 4; ++nn) {
                float mx = -__builtin_huge_valf();
#pragma unroll
                for (int m = 0; m < 8; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        mx = fmaxf(mx, acc[m][nn][r]);
                        acc[m][nn][r] = 0.f;
                    }
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                if (lane < 16) atomicMax(&best[wn * 64 + nn * 16 + lane], order_f32(mx));
            }
            flush_tile = tile;
            ks = 0;
            ++tile;
        } else {
            ++ks;
        }
    }
    if (wm == 0) barrier_keeping_loads();   // group 0 catches up with group 1's last barrier
    barrier_keeping_loads();                // every wave's maxima are in LDS
    if (flush_tile >= 0 && wm == 1) {
        tile_max[((size_t)qt * 256 + (tid - 256)) * n_tiles + flush_tile] = best[tid - 256];
    }
}

// plain check kernel: best score of query q over row tile t
__global__ void check_kernel(const bf16_t* rows, long long n, int ld, const bf16_t* queries, int n_q, long long n_tiles,
                             unsigned int* out) {
    const int q = blockIdx.y;
    const long long t = blockIdx.x;
    float best = -__builtin_huge_valf();
    for (int r = threadIdx.x; r < 256; r += blockDim.x) {
        long long row = t * 256 + r;
        if (row >= n) row = n - 1;
        float s = 0.f;
        for (int e = 0; e < ld; ++e)
            s += __uint_as_float((unsigned int)rows[row * ld + e] << 16) * __uint_as_float((unsigned int)queries[(size_t)q * ld + e] << 16);
        best = fmaxf(best, s);
    }
    __shared__ float red[256];
    red[threadIdx.x] = best;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0) out[(size_t)q * n_tiles + t] = __float_as_uint(red[0]);
}

static unsigned int unorder(unsigned int o) { return (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o; }

int main(int argc, char** argv) {
    const long long n = argc > 1 ? atoll(argv[1]) : 1000000;
    const int ld = 1024, n_q = argc > 2 ? atoi(argv[2]) : 4096;
    const int n_qtiles = n_q / 256;
    const long long n_tiles = (n + 255) / 256;
    int n_ranges = 256 / n_qtiles;
    if (n_ranges > n_tiles) n_ranges = (int)n_tiles;
    const long long tpr = (n_tiles + n_ranges - 1) / n_ranges;
    std::vector<bf16_t> h_rows((size_t)n * ld), h_q((size_t)n_q * ld);
    unsigned int seed = 12345;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return (bf16_t)(((seed >> 9) & 0x7f) | 0x3c00 | ((seed >> 1) & 0x8000)); };  // +-[0.0078, 0.0156)
    // argv[5] = 1: roughly normal values of scale 1/32 (what normalised rows of 1024 look like)
    // instead of one binade of uniform ones: the clock the chip holds depends on the data.
    const bool gaussian = argc > 5 && atoi(argv[5]) == 1;
    auto gauss = [&]() {
        float acc = 0.f;
        for (int i = 0; i < 4; ++i) { seed = seed * 1664525u + 1013904223u; acc += (float)(seed >> 8) / 16777216.f - 0.5f; }
        const float x = acc * 1.7320508f / 32.f;
        unsigned int u;
        memcpy(&u, &x, 4);
        return (bf16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
    };
    for (auto& v : h_rows) v = gaussian ? gauss() : rnd();
    for (auto& v : h_q) v = gaussian ? gauss() : rnd();
    bf16_t *d_rows, *d_q;
    unsigned int *d_out, *d_ref;
    if (hipMalloc(reinterpret_cast<void**>(&d_rows), h_rows.size() * 2) != hipSuccess) return 1;
    if (hipMalloc(reinterpret_cast<void**>(&d_q), h_q.size() * 2) != hipSuccess) return 1;
    if (hipMalloc(reinterpret_cast<void**>(&d_out), (size_t)n_q * n_tiles * 4) != hipSuccess) return 1;
    if (hipMalloc(reinterpret_cast<void**>(&d_ref), (size_t)n_q * n_tiles * 4) != hipSuccess) return 1;
    (void)hipMemcpy(d_rows, h_rows.data(), h_rows.size() * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_q, h_q.data(), h_q.size() * 2, hipMemcpyHostToDevice);
    const size_t lds = 2 * kStepBytes + 256 * 4;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(core_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pingpong_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const bool pingpong = argc > 4 && atoi(argv[4]) >= 1;
    const bool m16 = argc > 4 && atoi(argv[4]) >= 2;
    const bool m16i = argc > 4 && atoi(argv[4]) == 3;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pingpong16i_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pingpong16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    auto kernel = m16i ? pingpong16i_kernel : m16 ? pingpong16_kernel : pingpong ? pingpong_kernel : core_kernel;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const double flops = 2.0 * n_q * (double)n * ld;
    for (int mode : {0, 1, 2, 4}) {
        float sum = 0.f;
        const int reps = 5;
        for (int r = 0; r < reps + 1; ++r) {
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL(kernel, dim3(n_qtiles * n_ranges), dim3(kThreads), lds, 0, reinterpret_cast<const char*>(d_rows), n,
                               ld, reinterpret_cast<const char*>(d_q), n_ranges, tpr, d_out, mode);
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (r >= 1) sum += ms;
        }
        printf("{\"kernel\": \"%s\", \"mode\": \"%s\", \"ms\": %.4f, \"TFLOPs\": %.1f}\n", m16i ? "pingpong_16x16x32_interleaved" : m16 ? "pingpong_16x16x32" : pingpong ? "pingpong" : "lockstep", mode == 0 ? "full" : mode == 1 ? "loads_only" : mode == 2 ? "mfma_only" : "full_setprio",
               sum / reps, flops / (sum / reps) / 1e9);
        if (hipGetLastError() != hipSuccess) { printf("launch error\n"); return 1; }
    }
    if (argc > 3 && atoi(argv[3]) == 0) return 0;  // skip the (slow) comparison with the plain kernel
    hipLaunchKernelGGL(kernel, dim3(n_qtiles * n_ranges), dim3(kThreads), lds, 0, reinterpret_cast<const char*>(d_rows), n, ld,
                       reinterpret_cast<const char*>(d_q), n_ranges, tpr, d_out, 0);
    hipLaunchKernelGGL(check_kernel, dim3((unsigned)n_tiles, n_q), dim3(256), 0, 0, d_rows, n, ld, d_q, n_q, n_tiles, d_ref);
    std::vector<unsigned int> out((size_t)n_q * n_tiles), ref((size_t)n_q * n_tiles);
    (void)hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(ref.data(), d_ref, ref.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0.0;
    for (size_t i = 0; i < out.size(); ++i) {
        unsigned int u = unorder(out[i]);
        float a, b;
        memcpy(&a, &u, 4);
        memcpy(&b, &ref[i], 4);
        const double e = fabs((double)a - (double)b);
        if (e > worst) worst = e;
    }
    printf("{\"max_abs_error_vs_plain_kernel\": %.3e, \"compared\": %zu}\n", worst, out.size());
    return 0;
}
