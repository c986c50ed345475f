// Read bandwidth of tile-sweep access patterns (the matrix-core kernels' way of
// walking the rows): a four-wave workgroup owns a contiguous range of row tiles
// and walks each tile in K steps; one step reads SEG contiguous bytes of each of
// the tile's TM rows (TM * SEG = 16 KiB per step), D steps in flight per wave.
// SEG = 128 is what score_mfma.hip requests per step; larger SEG means fewer,
// longer pieces per row.  Loads go to registers and are xor-ed away.
// build: hipcc --offload-arch=gfx950 -O3 -o profiles/micro/tile_read_patterns profiles/micro/tile_read_patterns.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;

template <int SEG, int D, int POLICY>
__global__ __launch_bounds__(256) void tile_kernel(const char* __restrict__ src, long long n_rows, int row_bytes,
                                                   long long tiles_per_wg, unsigned int* out) {
    constexpr int TM = 16384 / SEG;          // rows per tile
    constexpr int LPR = SEG / 16;            // lanes per row
    constexpr int RPI = 64 / LPR;            // rows per wave instruction
    constexpr int L = TM / RPI / 4;          // instructions per wave and step
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long n_tiles = n_rows / TM;
    const long long t0 = (long long)blockIdx.x * tiles_per_wg;
    long long t1 = t0 + tiles_per_wg;
    if (t1 > n_tiles) t1 = n_tiles;
    const int ksteps = row_bytes / SEG;
    const long long total = (t1 > t0 ? t1 - t0 : 0) * ksteps;
    if (total == 0) return;  // a workgroup past the last tile has nothing it may read
    // lane's fixed offset inside a step: row (wave*L + i)*RPI + lane/LPR, byte (lane%LPR)*16
    const long long lane_off = (long long)(wave * L * RPI + lane / LPR) * row_bytes + (lane % LPR) * 16;
    auto addr = [&](long long s, int i) -> const u32x4* {
        const long long tile = t0 + s / ksteps;
        const int ks = (int)(s % ksteps);
        return reinterpret_cast<const u32x4*>(src + tile * TM * (long long)row_bytes + lane_off +
                                              (long long)i * RPI * row_bytes + (long long)ks * SEG);
    };
    u32x4 v[D][L];
    unsigned int acc = 0;
#pragma unroll
    for (int j = 0; j < D; ++j)
#pragma unroll
        for (int i = 0; i < L; ++i) {
            const u32x4* p = addr(j < total ? j : 0, i);
            v[j][i] = POLICY == 1 ? __builtin_nontemporal_load(p) : *p;
        }
    for (long long s = 0; s < total; s += D) {
#pragma unroll
        for (int j = 0; j < D; ++j) {
#pragma unroll
            for (int i = 0; i < L; ++i) acc ^= v[j][i][0] ^ v[j][i][1] ^ v[j][i][2] ^ v[j][i][3];
            const long long nx = s + j + D < total ? s + j + D : 0;
#pragma unroll
            for (int i = 0; i < L; ++i) {
                const u32x4* p = addr(nx, i);
                v[j][i] = POLICY == 1 ? __builtin_nontemporal_load(p) : *p;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < D; ++j)
#pragma unroll
        for (int i = 0; i < L; ++i) acc ^= v[j][i][0];
    if (acc == 0x12345678u) out[blockIdx.x] = acc;
}

// The same walk with 16-byte direct-to-LDS loads (what the matrix-core kernels
// issue), cache-policy bits AUX; nothing reads the LDS image.  Two steps in
// flight per wave, waited for with counted vmcnt.
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
template <int AUX>
__global__ __launch_bounds__(256) void tile_lds_kernel(const char* __restrict__ src, long long n_rows, int row_bytes,
                                                       long long tiles_per_wg, unsigned int* out) {
    constexpr int SEG = 128, TM = 128, LPR = 8, RPI = 8, L = 4;
    __shared__ __attribute__((aligned(16))) char lds[2 * 16384];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long n_tiles = n_rows / TM;
    const long long t0 = (long long)blockIdx.x * tiles_per_wg;
    long long t1 = t0 + tiles_per_wg;
    if (t1 > n_tiles) t1 = n_tiles;
    const int ksteps = row_bytes / SEG;
    const long long total = (t1 > t0 ? t1 - t0 : 0) * ksteps;
    if (total == 0) return;
    const long long lane_off = (long long)(wave * L * RPI + lane / LPR) * row_bytes + (lane % LPR) * 16;
    auto request = [&](long long s) {
        const long long tile = t0 + s / ksteps;
        const int ks = (int)(s % ksteps);
        char* dst = lds + (int)(s & 1) * 16384 + wave * L * 1024;
#pragma unroll
        for (int i = 0; i < L; ++i) {
            const char* p = src + tile * TM * (long long)row_bytes + lane_off + (long long)i * RPI * row_bytes + (long long)ks * SEG;
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)p, (lds_ptr_t)(dst + i * 1024), 16, 0, AUX);
        }
    };
    request(0);
    for (long long s = 0; s < total; ++s) {
        if (s + 1 < total) {
            request(s + 1);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    if (lds[threadIdx.x] == 0x7f && lds[threadIdx.x + 16384] == 0x7e) out[blockIdx.x] = 1;
}

template <int AUX>
void run_lds(const char* buf, long long n_rows, int row_bytes, unsigned int* out, int cus, hipEvent_t e0, hipEvent_t e1) {
    for (int per_cu : {2, 3, 4}) {
        const long long n_tiles = n_rows / 128;
        const int grid = cus * per_cu;
        const long long tpw = (n_tiles + grid - 1) / grid;
        float sum = 0.f;
        const int reps = 4;
        for (int r = 0; r < reps + 1; ++r) {
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL((tile_lds_kernel<AUX>), dim3(grid), dim3(256), 0, 0, buf, n_rows, row_bytes, tpw, out);
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (r >= 1) sum += ms;
        }
        const double bytes = (double)(n_tiles * 128) * row_bytes;
        printf("{\"row_bytes\": %d, \"direct_to_lds\": 1, \"aux\": %d, \"wg_per_cu\": %d, \"ms\": %.4f, \"GBps\": %.1f}\n", row_bytes, AUX,
               per_cu, sum / reps, bytes / (sum / reps) / 1e6);
        fflush(stdout);
    }
}

template <int SEG, int D, int POLICY>
void run(const char* buf, long long n_rows, int row_bytes, unsigned int* out, int cus, hipEvent_t e0, hipEvent_t e1) {
    constexpr int TM = 16384 / SEG;
    for (int per_cu : {1, 2, 3, 4, 6}) {
        const long long n_tiles = n_rows / TM;
        const int grid = cus * per_cu;
        const long long tpw = (n_tiles + grid - 1) / grid;
        float sum = 0.f;
        const int reps = 4;
        for (int r = 0; r < reps + 1; ++r) {
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL((tile_kernel<SEG, D, POLICY>), dim3(grid), dim3(256), 0, 0, buf, n_rows, row_bytes, tpw, out);
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (r >= 1) sum += ms;
        }
        const double bytes = (double)(n_tiles * TM) * row_bytes;
        printf("{\"row_bytes\": %d, \"seg_bytes\": %d, \"tile_rows\": %d, \"steps_in_flight\": %d, \"nontemporal\": %d, \"wg_per_cu\": %d, "
               "\"ms\": %.4f, \"GBps\": %.1f}\n", row_bytes, SEG, TM, D, POLICY, per_cu, sum / reps, bytes / (sum / reps) / 1e6);
        fflush(stdout);
    }
}

int main(int argc, char** argv) {
    const int row_bytes = argc > 1 ? atoi(argv[1]) : 1536;
    const long long n_rows = argc > 2 ? atoll(argv[2]) : 12500000;
    char* buf = nullptr;
    unsigned int* out = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&buf), n_rows * row_bytes) != hipSuccess) return 1;
    if (hipMalloc(reinterpret_cast<void**>(&out), 1 << 20) != hipSuccess) return 1;
    (void)hipMemset(buf, 1, n_rows * row_bytes);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
#define RUN_D(SEG, P) run<SEG, 1, P>(buf, n_rows, row_bytes, out, cus, e0, e1); run<SEG, 2, P>(buf, n_rows, row_bytes, out, cus, e0, e1); \
                      run<SEG, 3, P>(buf, n_rows, row_bytes, out, cus, e0, e1);
    if (argc > 3) {  // direct-to-LDS loads under every cache-policy combination (sc0 = 1, nt = 2, sc1 = 16)
        run_lds<0>(buf, n_rows, row_bytes, out, cus, e0, e1);
        run_lds<1>(buf, n_rows, row_bytes, out, cus, e0, e1);
        run_lds<2>(buf, n_rows, row_bytes, out, cus, e0, e1);
        run_lds<3>(buf, n_rows, row_bytes, out, cus, e0, e1);
        run_lds<16>(buf, n_rows, row_bytes, out, cus, e0, e1);
        run_lds<17>(buf, n_rows, row_bytes, out, cus, e0, e1);
        run_lds<18>(buf, n_rows, row_bytes, out, cus, e0, e1);
        run_lds<19>(buf, n_rows, row_bytes, out, cus, e0, e1);
        return 0;
    }
    RUN_D(128, 0) RUN_D(256, 0) RUN_D(512, 0)
    RUN_D(128, 1) RUN_D(256, 1) RUN_D(512, 1)
    return 0;
}
