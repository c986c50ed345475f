// Read bandwidth of MFMA-fragment-shaped global loads: can rows be streamed from
// HBM straight into registers in the operand layout of v_mfma_f32_16x16x32_bf16
// (lane l holds 16 bytes of row l % 16: every quarter-wave touches 16 rows) at the
// rate of row-contiguous loads?  10M rows x 2048 B of random data, streaming loads.
//   pattern 0: scan-like, lane l -> row l / 16 (4 rows), 16-byte chunk l % 16 (256 B per row)
//   pattern 1: fragment, lane l -> row l % 16 (16 rows), chunk l / 16 + 4 j (64 B per row and load)
//   pattern 2: fragment 32, lane l -> row l % 32 (32 rows), chunk l / 32 + 2 j (32 B per row and load)
// Each wave owns blocks of rows (grid-strided) and walks the whole row width, LOADS in flight.
// build: hipcc --offload-arch=gfx950 -O3 -o profiles/micro/frag_read profiles/micro/frag_read.hip
#include <hip/hip_runtime.h>
#include <cstdio>

using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;

__global__ void fill_random(unsigned int* p, size_t words) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < words; i += (size_t)gridDim.x * blockDim.x) {
        unsigned int x = (unsigned int)i * 2654435761u + 0x9e3779b9u;
        x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13;
        p[i] = (x & 0x807f807fu) | 0x3d003d00u;
    }
}

template <int PATTERN, int LOADS>
__global__ void read_kernel(const char* __restrict__ src, long long n_rows, int row_bytes, unsigned int* out) {
    constexpr int ROWS = PATTERN == 0 ? 4 : PATTERN == 1 ? 16 : 32;      // rows per load instruction
    constexpr int SEG = 1024 / ROWS;                                     // bytes of a row per load
    const int lane = threadIdx.x & 63, waves = blockDim.x >> 6;
    const int lrow = PATTERN == 0 ? lane / 16 : PATTERN == 1 ? lane % 16 : lane % 32;
    const int lchunk = PATTERN == 0 ? lane % 16 : PATTERN == 1 ? lane / 16 : lane / 32;
    const long long nblocks = n_rows / ROWS;
    const int steps = row_bytes / SEG;   // loads per row block
    unsigned int x = 0;
    for (long long g = (long long)blockIdx.x * waves + (threadIdx.x >> 6); g < nblocks; g += (long long)gridDim.x * waves) {
        const char* base = src + (g * ROWS + lrow) * (long long)row_bytes + lchunk * 16;
        for (int s0 = 0; s0 < steps; s0 += LOADS) {
            u32x4 v[LOADS];
#pragma unroll
            for (int u = 0; u < LOADS; ++u)
                v[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(base + (long long)(s0 + u) * SEG));
#pragma unroll
            for (int u = 0; u < LOADS; ++u) x ^= v[u][0] ^ v[u][1] ^ v[u][2] ^ v[u][3];
        }
    }
    if (x == 0x12345678u) out[blockIdx.x] = x;
}

template <int PATTERN, int LOADS>
void run(const char* buf, long long n_rows, int row_bytes, unsigned int* out, int cus, int waves, int per_cu) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float sum = 0.f;
    const int reps = 8;
    for (int r = 0; r < reps + 2; ++r) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((read_kernel<PATTERN, LOADS>), dim3(cus * per_cu), dim3(waves * 64), 0, 0, buf, n_rows, row_bytes, out);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (r >= 2) sum += ms;
    }
    printf("{\"pattern\": %d, \"loads_in_flight\": %d, \"waves\": %d, \"wg_per_cu\": %d, \"ms\": %.4f, \"GBps\": %.1f}\n", PATTERN, LOADS, waves,
           per_cu, sum / reps, (double)n_rows * row_bytes / (sum / reps) / 1e6);
    fflush(stdout);
}

int main() {
    const int row_bytes = 2048;
    const long long n_rows = 10000000 / 32 * 32;
    char* buf = nullptr;
    unsigned int* out = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&buf), n_rows * row_bytes) != hipSuccess) return 1;
    if (hipMalloc(reinterpret_cast<void**>(&out), 1 << 20) != hipSuccess) return 1;
    hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, reinterpret_cast<unsigned int*>(buf), (size_t)n_rows * row_bytes / 4);
    (void)hipDeviceSynchronize();
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    for (int per_cu : {1, 2}) {
        run<0, 8>(buf, n_rows, row_bytes, out, cus, 8, per_cu);
        run<1, 8>(buf, n_rows, row_bytes, out, cus, 8, per_cu);
        run<1, 16>(buf, n_rows, row_bytes, out, cus, 8, per_cu);
        run<1, 32>(buf, n_rows, row_bytes, out, cus, 8, per_cu);
        run<2, 8>(buf, n_rows, row_bytes, out, cus, 8, per_cu);
        run<2, 16>(buf, n_rows, row_bytes, out, cus, 8, per_cu);
        run<2, 32>(buf, n_rows, row_bytes, out, cus, 8, per_cu);
    }
    run<1, 16>(buf, n_rows, row_bytes, out, cus, 4, 1);
    run<1, 32>(buf, n_rows, row_bytes, out, cus, 4, 1);
    run<1, 16>(buf, n_rows, row_bytes, out, cus, 16, 1);
    return 0;
}
