// Read-bandwidth ceiling of this GPU for a streaming kernel: every lane reads
// 16 bytes per load, a wave 1 KiB contiguous per instruction, a workgroup of
// 16 waves walks the buffer grid-strided, eight loads in flight per lane; the
// values are folded into one word per workgroup so that nothing is optimised
// away.  Prints GB/s for default-policy and non-temporal loads.
// build: hipcc --offload-arch=gfx950 -O3 -o profiles/micro/read_ceiling profiles/micro/read_ceiling.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;

template <bool NT>
__global__ __launch_bounds__(1024) void read_kernel(const u32x4* __restrict__ src, long long n_vec, unsigned int* out) {
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    unsigned int acc = 0;
    long long i = tid;
    for (; i + 7 * stride < n_vec; i += 8 * stride) {
        u32x4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
            v[j] = NT ? __builtin_nontemporal_load(src + i + j * stride) : src[i + j * stride];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc ^= v[j][0] ^ v[j][1] ^ v[j][2] ^ v[j][3];
    }
    for (; i < n_vec; i += stride) {
        const u32x4 v = src[i];
        acc ^= v[0] ^ v[1] ^ v[2] ^ v[3];
    }
    if (acc == 0x12345678u) out[blockIdx.x] = acc;  // practically never: keeps the loads alive
}

int main(int argc, char** argv) {
    const double gb = argc > 1 ? atof(argv[1]) : 20.48;
    const long long bytes = (long long)(gb * 1e9) / 16 * 16;
    void* buf = nullptr;
    unsigned int* out = nullptr;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&out), 1 << 20) != hipSuccess) return 1;
    (void)hipMemset(buf, 1, bytes);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    for (int nt = 0; nt < 2; ++nt)
        for (int per_cu = 1; per_cu <= 2; ++per_cu) {
            const int grid = prop.multiProcessorCount * per_cu;
            float best = 1e30f, sum = 0.f;
            const int reps = 10;
            for (int r = 0; r < reps + 2; ++r) {
                (void)hipEventRecord(e0, 0);
                if (nt) hipLaunchKernelGGL(read_kernel<true>, dim3(grid), dim3(1024), 0, 0, static_cast<const u32x4*>(buf), bytes / 16, out);
                else hipLaunchKernelGGL(read_kernel<false>, dim3(grid), dim3(1024), 0, 0, static_cast<const u32x4*>(buf), bytes / 16, out);
                (void)hipEventRecord(e1, 0);
                (void)hipEventSynchronize(e1);
                float ms = 0.f;
                (void)hipEventElapsedTime(&ms, e0, e1);
                if (r >= 2) {
                    sum += ms;
                    best = ms < best ? ms : best;
                }
            }
            printf("{\"bytes\": %lld, \"nontemporal\": %d, \"workgroups_per_cu\": %d, \"avg_ms\": %.4f, \"best_ms\": %.4f, \"avg_GBps\": %.1f, \"best_GBps\": %.1f}\n",
                   bytes, nt, per_cu, sum / reps, best, bytes / (sum / reps) / 1e6, bytes / best / 1e6);
        }
    return 0;
}
