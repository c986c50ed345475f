// Launch shapes of the single-query scan with its full arithmetic, side by side
// on one box: 10M rows x 2048 B of random bf16, streaming loads, sixteen lanes per
// row (one load instruction = 256 B of each of four rows), query chunks from LDS,
// DPP row sums, a threshold test per group.
//   WAVES   waves per workgroup (one workgroup per CU)
//   QUADS   row quads per wave group (rows per group = 4 * QUADS)
//   SWEEPS  sweeps requested together (loads per batch = SWEEPS * QUADS)
//   SETS    1: a batch is requested, then consumed;  2: two register sets, the
//           next batch is requested before the current one is consumed
// Every kernel carries __launch_bounds__ for its own workgroup size: without
// it the compiler assumes 1024 threads, caps a wave at 128 registers and spills
// the second register set (which is what made every pipelined variant of
// scan_steps.hip run at 1-2 TB/s: a scratch store behind every load, each
// behind s_waitcnt vmcnt(0)).
// Also reports when the workgroups finished (100 MHz wall clock): the spread is
// what a static split of the rows leaves on the table.
// build: hipcc --offload-arch=gfx950 -O3 -o profiles/micro/scan_forms profiles/micro/scan_forms.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
using f32x4 = __attribute__((ext_vector_type(4))) float;

__global__ void fill_random(unsigned int* p, size_t words) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < words; i += stride) {
        unsigned int x = (unsigned int)i * 2654435761u + 0x9e3779b9u;
        x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
        const unsigned int lo = (x & 0x807fu) | 0x3d00u, hi = ((x >> 16) & 0x807fu) | 0x3d00u;
        p[i] = lo | (hi << 16);
    }
}

template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_move<0xB1>(v);
    v += dpp_move<0x4E>(v);
    v += dpp_move<0x124>(v);
    v += dpp_move<0x128>(v);
    return v;
}

// A unit of work: SWEEPS sweeps (from sweep s0) of the QUADS row quads of group g.
struct Unit {
    long long g;
    int s0;
};

template <int QUADS, int SWEEPS>
__device__ __forceinline__ void request(const char* rows, long long n, int row_bytes, int nit, Unit u, int col, int lrow,
                                        u32x4 (&raw)[SWEEPS][QUADS]) {
    const char* rp[QUADS];
#pragma unroll
    for (int qd = 0; qd < QUADS; ++qd) {
        long long row = u.g * (4 * QUADS) + qd * 4 + lrow;
        row = row < n ? row : n - 1;
        rp[qd] = rows + row * row_bytes;
    }
#pragma unroll
    for (int s = 0; s < SWEEPS; ++s) {
        // no branch around a load: nit is a multiple of SWEEPS here (a branch makes the
        // compiler wait for every load in flight at each join)
        const int chunk = (u.s0 + s) * 16 + col;
#pragma unroll
        for (int qd = 0; qd < QUADS; ++qd)
            raw[s][qd] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(rp[qd] + (size_t)chunk * 16));
    }
}

template <int QUADS, int SWEEPS>
__device__ __forceinline__ void consume(const float* q_lds, int nit, Unit u, int col, const u32x4 (&raw)[SWEEPS][QUADS],
                                        float (&acc)[QUADS]) {
#pragma unroll
    for (int s = 0; s < SWEEPS; ++s) {
        {
            const int chunk = (u.s0 + s) * 16 + col;
            const f32x4* qp = reinterpret_cast<const f32x4*>(q_lds + (size_t)chunk * 8);
            const f32x4 a = qp[0], b = qp[1];
            const float q[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
#pragma unroll
            for (int qd = 0; qd < QUADS; ++qd)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[qd] = __builtin_fmaf(__uint_as_float(raw[s][qd][i] << 16), q[2 * i], acc[qd]);
                    acc[qd] = __builtin_fmaf(__uint_as_float(raw[s][qd][i] & 0xffff0000u), q[2 * i + 1], acc[qd]);
                }
        }
    }
}

template <int QUADS>
__device__ __forceinline__ void finish_group(float (&acc)[QUADS], int col, float& best, unsigned int& hits) {
    float total = 0.f;
#pragma unroll
    for (int qd = 0; qd < QUADS; ++qd) {
        const float t = row16_sum(acc[qd]);
        if (col == qd) total = t;
        acc[qd] = 0.f;
    }
    const bool pass = col < QUADS && total > best;
    if (__ballot(pass) != 0ull) {   // rare once best has settled
        float m = pass ? total : best;
#pragma unroll
        for (int x = 32; x >= 1; x >>= 1) m = fmaxf(m, __shfl_xor(m, x, 64));
        best = m;
        hits += 1;
    }
}

template <int WAVES, int QUADS, int SWEEPS, int SETS>
__global__ __launch_bounds__(WAVES * 64) void scan_form(const char* __restrict__ rows, long long n, int row_bytes,
                                                          unsigned int* out, unsigned long long* stamps) {
    extern __shared__ float q_lds[];
    const int lane = threadIdx.x & 63;
    // scalar wave index: the unit counters and loop tests then live in scalar registers and the
    // loop is plain scalar control flow (with a vector wave index every test is an exec-mask
    // region and the compiler's wait counts turn conservative at the joins)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = lane & 15, lrow = lane >> 4;
    const int nit = row_bytes / 256;
    const long long ngroups = (n + 4 * QUADS - 1) / (4 * QUADS);
    const long long gstride = (long long)gridDim.x * WAVES;
    const unsigned long long t0 = wall_clock64();

    Unit cur{(long long)blockIdx.x * WAVES + wave, 0};
    auto next_of = [&](Unit u) {
        u.s0 += SWEEPS;
        if (u.s0 >= nit) { u.s0 = 0; u.g += gstride; }
        return u;
    };
    u32x4 ra[SWEEPS][QUADS], rb[SWEEPS][QUADS];
    if (cur.g < ngroups) request<QUADS, SWEEPS>(rows, n, row_bytes, nit, cur, col, lrow, ra);

    for (int i = threadIdx.x; i < row_bytes / 2; i += WAVES * 64) q_lds[i] = 0.001f * (float)(i % 97) - 0.04f;
    __syncthreads();

    float acc[QUADS];
#pragma unroll
    for (int qd = 0; qd < QUADS; ++qd) acc[qd] = 0.f;
    float best = -1e30f;
    unsigned int hits = 0;

    if constexpr (SETS == 1) {
        bool first = true;
        while (cur.g < ngroups) {
            if (!first) request<QUADS, SWEEPS>(rows, n, row_bytes, nit, cur, col, lrow, ra);
            first = false;
            consume<QUADS, SWEEPS>(q_lds, nit, cur, col, ra, acc);
            if (cur.s0 + SWEEPS >= nit) finish_group<QUADS>(acc, col, best, hits);
            cur = next_of(cur);
        }
    } else {
        // ra holds (or awaits) unit cur.  The loop body takes two units and has no
        // exit in its middle (a break between the halves joins the back edge in the
        // compiler's control flow, and its wait counts then assume the other set's
        // loads may still be in flight at the loop head: vmcnt(0) before every
        // request).  A request past the wave's last unit re-reads that unit (in
        // bounds, never consumed); an odd last unit is consumed behind the loop.
        const int upg = nit / SWEEPS;   // units per group
        const long long my_groups = cur.g < ngroups ? (ngroups - cur.g + gstride - 1) / gstride : 0;
        const long long total = my_groups * upg;
        long long i = 0;
        for (; i + 2 <= total; i += 2) {
            const Unit nx = next_of(cur);
            request<QUADS, SWEEPS>(rows, n, row_bytes, nit, nx, col, lrow, rb);
            __builtin_amdgcn_sched_barrier(0);   // or the scheduler moves the consumption in front of the requests
            consume<QUADS, SWEEPS>(q_lds, nit, cur, col, ra, acc);
            if (cur.s0 + SWEEPS >= nit) finish_group<QUADS>(acc, col, best, hits);
            cur = next_of(nx);
            request<QUADS, SWEEPS>(rows, n, row_bytes, nit, i + 2 < total ? cur : nx, col, lrow, ra);
            __builtin_amdgcn_sched_barrier(0);
            consume<QUADS, SWEEPS>(q_lds, nit, nx, col, rb, acc);
            if (nx.s0 + SWEEPS >= nit) finish_group<QUADS>(acc, col, best, hits);
        }
        if (i < total) {
            consume<QUADS, SWEEPS>(q_lds, nit, cur, col, ra, acc);
            finish_group<QUADS>(acc, col, best, hits);
        }
    }
    if (hits == 0xffffffffu) out[blockIdx.x] = __float_as_uint(best);
    __syncthreads();
    if (threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = t0;
        stamps[2 * blockIdx.x + 1] = wall_clock64();
    }
}

// Dynamic split of the rows: the groups are cut into chunks that shrink towards the end (CA
// groups each over the first part, CB over the next, single groups over the rest); a wave takes
// two chunks by its index and claims every further one from a device-wide counter, one chunk
// ahead of need, so that CUs the memory system serves faster end up with more rows and all of
// them finish together.
struct DynMap {
    long long G, GA, GB;   // groups; [0, GA) in chunks of CA, [GA, GB) in chunks of CB, [GB, G) singly
    int CA, CB;
    long long nA, nB;      // slots of the first two parts
    long long J, J0;       // static part (J0 = J * W: first group of the dynamic part): every wave first takes J groups, wid + j * W (then GA counts from J * W)
};
__device__ __forceinline__ bool chunk_of(const DynMap& m, long long slot, long long& g0, long long& g1) {
    if (slot < m.nA) { g0 = m.J0 + slot * m.CA; g1 = g0 + m.CA; return true; }
    slot -= m.nA;
    if (slot < m.nB) { g0 = m.GA + slot * m.CB; g1 = g0 + m.CB; return true; }
    slot -= m.nB;
    g0 = m.GB + slot; g1 = g0 + 1;
    return g0 < m.G;
}

template <int WAVES, int QUADS, int SWEEPS>
__global__ __launch_bounds__(WAVES * 64) void scan_dyn(const char* __restrict__ rows, long long n, int row_bytes,
                                                         unsigned int* out, unsigned long long* stamps, unsigned int* counter,
                                                         DynMap m) {
    extern __shared__ float q_lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = lane & 15, lrow = lane >> 4;
    const int nit = row_bytes / 256;
    const long long W = (long long)gridDim.x * WAVES;
    const long long wid = (long long)blockIdx.x * WAVES + wave;
    const unsigned long long t0 = wall_clock64();

    // iterator over this wave's units, one unit ahead of the consumption
    long long g = 0, g_end = 0, next_slot = W + wid, j_static = 0;
    int s0 = 0;
    bool valid;
    if (m.J > 0) { g = wid; g_end = wid + 1; valid = true; next_slot = -1; }
    else valid = chunk_of(m, wid, g, g_end);
    unsigned int tick = 0;
    unsigned int opaque_zero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(opaque_zero));
    auto claim = [&]() {
        // only lane 0's value is ever read (readfirstlane); leaving the other lanes' value
        // undefined keeps the compiler from merging with the old value at the join, which
        // would need the atomic's result at once (s_waitcnt vmcnt(0) right behind it)
        // (and an address the compiler cannot prove uniform keeps its atomic optimizer away,
        // which would turn the add into a wave reduction that reads the result at once)
        unsigned int r;
        if (lane == 0) r = __hip_atomic_fetch_add(counter + opaque_zero, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tick = r;
    };
    claim();
    auto advance = [&]() {
        if (!valid) return;
        s0 += SWEEPS;
        if (s0 < nit) return;
        s0 = 0;
        g += 1;
        if (g < g_end) return;
        if (m.J > 0) {
            if (++j_static < m.J) { g = wid + j_static * W; g_end = g + 1; return; }
            if (next_slot < 0) next_slot = (long long)__builtin_amdgcn_readfirstlane(tick), claim();
        }
        long long a, b;
        if (chunk_of(m, next_slot, a, b)) {
            g = a; g_end = b;
            next_slot = (m.J > 0 ? 0 : 2 * W) + (long long)__builtin_amdgcn_readfirstlane(tick);
            claim();
        } else {
            valid = false;
        }
    };

    u32x4 ra[SWEEPS][QUADS], rb[SWEEPS][QUADS];
    Unit cur{g, s0};
    bool cur_ok = valid;
    if (cur_ok) request<QUADS, SWEEPS>(rows, n, row_bytes, nit, cur, col, lrow, ra);

    for (int i = threadIdx.x; i < row_bytes / 2; i += WAVES * 64) q_lds[i] = 0.001f * (float)(i % 97) - 0.04f;
    __syncthreads();

    float acc[QUADS];
#pragma unroll
    for (int qd = 0; qd < QUADS; ++qd) acc[qd] = 0.f;
    float best = -1e30f;
    unsigned int hits = 0, groups_done = 0;

    while (cur_ok) {
        advance();
        const bool nx_ok = valid;
        const Unit nx = nx_ok ? Unit{g, s0} : cur;
        request<QUADS, SWEEPS>(rows, n, row_bytes, nit, nx, col, lrow, rb);
        __builtin_amdgcn_sched_barrier(0);
        consume<QUADS, SWEEPS>(q_lds, nit, cur, col, ra, acc);
        if (cur.s0 + SWEEPS >= nit) { finish_group<QUADS>(acc, col, best, hits); groups_done += 1; }
        advance();
        const bool c2_ok = valid;
        const Unit c2 = c2_ok ? Unit{g, s0} : cur;
        request<QUADS, SWEEPS>(rows, n, row_bytes, nit, c2, col, lrow, ra);
        __builtin_amdgcn_sched_barrier(0);
        // consumed even when there was no next unit (a copy of the last one, whose sums nobody
        // reads): a branch around it would leave the compiler unable to count what is in flight
        consume<QUADS, SWEEPS>(q_lds, nit, nx, col, rb, acc);
        if (nx_ok && nx.s0 + SWEEPS >= nit) { finish_group<QUADS>(acc, col, best, hits); groups_done += 1; }
        cur = c2;
        cur_ok = c2_ok;
    }
    if (hits == 0xffffffffu) out[blockIdx.x] = __float_as_uint(best);
    if (lane == 0) atomicAdd(&out[1024], groups_done);
    __syncthreads();
    if (threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = t0;
        stamps[2 * blockIdx.x + 1] = wall_clock64();
    }
}

template <int WAVES, int QUADS, int SWEEPS>
void run_dyn(const char* buf, long long n_rows, int row_bytes, unsigned int* out, unsigned long long* stamps, unsigned int* counter,
             int cus, int CA, int CB, double fa, double fb, hipEvent_t e0, hipEvent_t e1, double fstatic = 0.0) {
    float sum = 0.f, best = 1e9f;
    const int reps = 10;
    const int grid = cus;
    DynMap m;
    m.G = (n_rows + 4 * QUADS - 1) / (4 * QUADS);
    m.CA = CA; m.CB = CB;
    const long long W = (long long)grid * WAVES;
    m.J = (long long)(fstatic * m.G) / W;
    m.J0 = m.J * W;
    m.GA = (long long)(fa * m.G) < m.J0 ? 0 : ((long long)(fa * m.G) - m.J0) / CA * CA;   // relative to J0
    m.GB = m.GA + ((long long)(fb * m.G) - m.J0 - m.GA) / CB * CB;
    m.GA += m.J0; m.GB += m.J0;
    m.nA = (m.GA - m.J0) / CA;
    m.nB = (m.GB - m.GA) / CB;
    unsigned int done = 0;
    for (int r = 0; r < reps + 2; ++r) {
        (void)hipMemsetAsync(counter, 0, 4, 0);
        (void)hipMemsetAsync(out + 1024, 0, 4, 0);
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((scan_dyn<WAVES, QUADS, SWEEPS>), dim3(grid), dim3(WAVES * 64), row_bytes * 2, 0, buf, n_rows, row_bytes,
                           out, stamps, counter, m);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (r >= 2) { sum += ms; best = ms < best ? ms : best; }
    }
    (void)hipMemcpy(&done, out + 1024, 4, hipMemcpyDeviceToHost);
    unsigned int claims = 0;
    (void)hipMemcpy(&claims, counter, 4, hipMemcpyDeviceToHost);
    std::vector<unsigned long long> h(2 * grid);
    (void)hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long first = ~0ull;
    for (int i = 0; i < grid; ++i) first = std::min(first, h[2 * i]);
    std::vector<double> ends(grid);
    for (int i = 0; i < grid; ++i) ends[i] = (double)(h[2 * i + 1] - first) * 1e-5;
    std::sort(ends.begin(), ends.end());
    const double bytes = (double)n_rows * row_bytes;
    printf("{\"dynamic\": 1, \"waves\": %d, \"rows_per_group\": %d, \"loads_per_batch\": %d, \"static\": %.3f, \"chunks\": [%d, %d, 1], \"upto\": [%.3f, %.3f], "
           "\"claims\": %u, \"groups_ok\": %s, \"ms\": %.4f, \"ms_min\": %.4f, \"GBps\": %.1f, "
           "\"wg_end_ms\": {\"min\": %.4f, \"p10\": %.4f, \"p50\": %.4f, \"p90\": %.4f, \"max\": %.4f}}\n",
           WAVES, 4 * QUADS, SWEEPS * QUADS, fstatic, CA, CB, fa, fb, claims, done == (unsigned int)m.G ? "true" : "false", sum / reps, best,
           bytes / (sum / reps) / 1e6, ends[0], ends[grid / 10], ends[grid / 2], ends[grid * 9 / 10], ends[grid - 1]);
    fflush(stdout);
}

template <int WAVES, int QUADS, int SWEEPS, int SETS>
void run(const char* buf, long long n_rows, int row_bytes, unsigned int* out, unsigned long long* stamps, int cus, int per_cu,
         hipEvent_t e0, hipEvent_t e1) {
    float sum = 0.f, best = 1e9f;
    const int reps = 10;
    const int grid = cus * per_cu;
    for (int r = 0; r < reps + 2; ++r) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((scan_form<WAVES, QUADS, SWEEPS, SETS>), dim3(grid), dim3(WAVES * 64), row_bytes * 2, 0, buf, n_rows,
                           row_bytes, out, stamps);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (r >= 2) { sum += ms; best = ms < best ? ms : best; }
    }
    std::vector<unsigned long long> h(2 * grid);
    (void)hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long first = ~0ull, last = 0;
    for (int i = 0; i < grid; ++i) { first = std::min(first, h[2 * i]); last = std::max(last, h[2 * i + 1]); }
    std::vector<double> ends(grid);
    for (int i = 0; i < grid; ++i) ends[i] = (double)(h[2 * i + 1] - first) * 1e-5;   // ms at 100 MHz
    std::sort(ends.begin(), ends.end());
    const double bytes = (double)n_rows * row_bytes;
    printf("{\"waves\": %d, \"rows_per_group\": %d, \"loads_per_batch\": %d, \"sets\": %d, \"wg_per_cu\": %d, \"ms\": %.4f, "
           "\"ms_min\": %.4f, \"GBps\": %.1f, \"wg_end_ms\": {\"min\": %.4f, \"p10\": %.4f, \"p50\": %.4f, \"p90\": %.4f, \"max\": %.4f}}\n",
           WAVES, 4 * QUADS, SWEEPS * QUADS, SETS, per_cu, sum / reps, best, bytes / (sum / reps) / 1e6, ends[0], ends[grid / 10],
           ends[grid / 2], ends[grid * 9 / 10], ends[grid - 1]);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int row_bytes = 2048;
    const long long n_rows = 10000000;
    char* buf = nullptr;
    unsigned int* out = nullptr;
    unsigned long long* stamps = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&buf), n_rows * row_bytes) != hipSuccess) return 1;
    if (hipMalloc(reinterpret_cast<void**>(&out), 1 << 20) != hipSuccess) return 1;
    if (hipMalloc(reinterpret_cast<void**>(&stamps), 1 << 20) != hipSuccess) return 1;
    hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, reinterpret_cast<unsigned int*>(buf), (size_t)n_rows * row_bytes / 4);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    unsigned int* counter = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&counter), 256) != hipSuccess) return 1;
    const int only_dyn = argc > 1 ? atoi(argv[1]) : 0;
    for (int pass = 0; pass < 2; ++pass) {
        run_dyn<4, 4, 4>(buf, n_rows, row_bytes, out, stamps, counter, cus, 16, 4, 0.80, 0.95, e0, e1);
        run_dyn<4, 4, 4>(buf, n_rows, row_bytes, out, stamps, counter, cus, 8, 2, 0.85, 0.97, e0, e1);
        run_dyn<4, 4, 4>(buf, n_rows, row_bytes, out, stamps, counter, cus, 32, 4, 0.90, 0.98, e0, e1);
        run_dyn<4, 4, 2>(buf, n_rows, row_bytes, out, stamps, counter, cus, 16, 4, 0.80, 0.95, e0, e1);
        run_dyn<8, 2, 4>(buf, n_rows, row_bytes, out, stamps, counter, cus, 16, 4, 0.80, 0.95, e0, e1);
        run_dyn<8, 2, 4>(buf, n_rows, row_bytes, out, stamps, counter, cus, 32, 8, 0.90, 0.98, e0, e1);
        run_dyn<8, 4, 2>(buf, n_rows, row_bytes, out, stamps, counter, cus, 16, 4, 0.80, 0.95, e0, e1);
        run_dyn<4, 4, 4>(buf, n_rows, row_bytes, out, stamps, counter, cus, 4, 2, 0.97, 0.99, e0, e1, 0.90);
        run_dyn<4, 4, 4>(buf, n_rows, row_bytes, out, stamps, counter, cus, 4, 2, 0.98, 0.995, e0, e1, 0.94);
        run_dyn<4, 4, 4>(buf, n_rows, row_bytes, out, stamps, counter, cus, 2, 2, 0.98, 0.995, e0, e1, 0.96);
        run_dyn<4, 4, 4>(buf, n_rows, row_bytes, out, stamps, counter, cus, 8, 2, 0.97, 0.99, e0, e1, 0.85);
        run_dyn<8, 2, 4>(buf, n_rows, row_bytes, out, stamps, counter, cus, 8, 4, 0.97, 0.99, e0, e1, 0.90);
        run_dyn<8, 2, 4>(buf, n_rows, row_bytes, out, stamps, counter, cus, 4, 2, 0.98, 0.995, e0, e1, 0.94);
        run_dyn<8, 4, 2>(buf, n_rows, row_bytes, out, stamps, counter, cus, 4, 2, 0.97, 0.99, e0, e1, 0.90);
        if (only_dyn) continue;
        run<8, 2, 4, 1>(buf, n_rows, row_bytes, out, stamps, cus, 1, e0, e1);   // the shipped shape
        run<8, 2, 4, 2>(buf, n_rows, row_bytes, out, stamps, cus, 1, e0, e1);
        run<8, 2, 2, 2>(buf, n_rows, row_bytes, out, stamps, cus, 1, e0, e1);
        run<8, 1, 4, 2>(buf, n_rows, row_bytes, out, stamps, cus, 1, e0, e1);
        run<4, 4, 4, 1>(buf, n_rows, row_bytes, out, stamps, cus, 1, e0, e1);
        run<4, 4, 4, 2>(buf, n_rows, row_bytes, out, stamps, cus, 1, e0, e1);
        run<4, 4, 2, 2>(buf, n_rows, row_bytes, out, stamps, cus, 1, e0, e1);
        run<4, 2, 4, 2>(buf, n_rows, row_bytes, out, stamps, cus, 1, e0, e1);
        run<4, 2, 8, 2>(buf, n_rows, row_bytes, out, stamps, cus, 1, e0, e1);
        run<4, 2, 4, 2>(buf, n_rows, row_bytes, out, stamps, cus, 2, e0, e1);
        run<2, 4, 4, 2>(buf, n_rows, row_bytes, out, stamps, cus, 2, e0, e1);
        run<4, 4, 4, 2>(buf, n_rows, row_bytes, out, stamps, cus, 2, e0, e1);
    }
    return 0;
}
