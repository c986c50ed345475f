// Read-bandwidth of scan-like access patterns: a wave reads R consecutive rows
// of ROWB bytes (lane l takes 16-byte chunk it*64 + l of each row), wave groups
// are grid-strided.  Sweeps R, waves per workgroup, workgroups per CU and the
// load policy; prints one JSON line per variant.
// build: hipcc --offload-arch=gfx950 -O3 -o profiles/micro/read_patterns profiles/micro/read_patterns.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;

template <int R, int POLICY>
__global__ void rows_kernel(const char* __restrict__ src, long long n_rows, int row_bytes, unsigned int* out) {
    const int lane = threadIdx.x & 63;
    const int waves = blockDim.x >> 6;
    const long long ngroups = n_rows / R;
    const int nit = row_bytes / 1024;
    unsigned int acc = 0;
    for (long long g = (long long)blockIdx.x * waves + (threadIdx.x >> 6); g < ngroups; g += (long long)gridDim.x * waves) {
        const char* base = src + g * R * (long long)row_bytes;
        for (int it = 0; it < nit; ++it) {
            u32x4 v[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const u32x4* p = reinterpret_cast<const u32x4*>(base + (long long)r * row_bytes + (it * 64 + lane) * 16);
                v[r] = POLICY == 1 ? __builtin_nontemporal_load(p) : *p;
            }
#pragma unroll
            for (int r = 0; r < R; ++r) acc ^= v[r][0] ^ v[r][1] ^ v[r][2] ^ v[r][3];
        }
    }
    if (acc == 0x12345678u) out[blockIdx.x] = acc;
}

template <int R, int POLICY>
void run(const char* buf, long long n_rows, int row_bytes, unsigned int* out, int cus, hipEvent_t e0, hipEvent_t e1) {
    for (int waves : {2, 4, 8, 16})
        for (int per_cu : {1, 2, 4, 8}) {
            if (waves * per_cu > 32) continue;
            float sum = 0.f;
            const int reps = 6;
            for (int r = 0; r < reps + 1; ++r) {
                (void)hipEventRecord(e0, 0);
                hipLaunchKernelGGL((rows_kernel<R, POLICY>), dim3(cus * per_cu), dim3(waves * 64), 0, 0, buf, n_rows, row_bytes, out);
                (void)hipEventRecord(e1, 0);
                (void)hipEventSynchronize(e1);
                float ms = 0.f;
                (void)hipEventElapsedTime(&ms, e0, e1);
                if (r >= 1) sum += ms;
            }
            const double bytes = (double)(n_rows / R * R) * row_bytes;
            printf("{\"rows_per_wave\": %d, \"nontemporal\": %d, \"waves\": %d, \"wg_per_cu\": %d, \"ms\": %.4f, \"GBps\": %.1f}\n", R, POLICY,
                   waves, per_cu, sum / reps, bytes / (sum / reps) / 1e6);
        }
}

int main() {
    const int row_bytes = 2048;
    const long long n_rows = 10000000;
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
    run<2, 1>(buf, n_rows, row_bytes, out, cus, e0, e1);
    run<4, 1>(buf, n_rows, row_bytes, out, cus, e0, e1);
    run<8, 1>(buf, n_rows, row_bytes, out, cus, e0, e1);
    run<16, 1>(buf, n_rows, row_bytes, out, cus, e0, e1);
    run<8, 0>(buf, n_rows, row_bytes, out, cus, e0, e1);
    return 0;
}
