// Where the single-query scan loses bandwidth against a bare read of the same
// access pattern: the scan's loop is rebuilt step by step on random bf16 data
// (10M rows x 2048 B, streaming loads) and each step is timed.
//   level 0  loads only (xor of the loaded words)
//   level 1  + bf16 unpack and 8 FMAs per 16-byte chunk, query chunk in registers
//   level 2  + query chunk read from LDS per sweep
//   level 3  + butterfly reduction of the R row sums and a threshold test per group
// PF = 1: the loads of group g + 1 are issued before group g is consumed.
// build: hipcc --offload-arch=gfx950 -O3 -o profiles/micro/scan_steps profiles/micro/scan_steps.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
using f32x4 = __attribute__((ext_vector_type(4))) float;

__global__ void fill_random(unsigned int* p, size_t words) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < words; i += stride) {
        unsigned int x = (unsigned int)i * 2654435761u + 0x9e3779b9u;
        x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
        // two bf16 values of moderate magnitude: keep exponents in a narrow band around 2^-5
        const unsigned int lo = (x & 0x807fu) | 0x3d00u, hi = ((x >> 16) & 0x807fu) | 0x3d00u;
        p[i] = lo | (hi << 16);
    }
}

template <int R>
__device__ __forceinline__ void load_group(const char* base, int row_bytes, int it, int lane, u32x4 (&v)[R]) {
#pragma unroll
    for (int r = 0; r < R; ++r)
        v[r] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(base + (long long)r * row_bytes + (it * 64 + lane) * 16));
}

template <int R, int LEVEL>
__device__ __forceinline__ void consume(const u32x4 (&v)[R], const float* q_lds, int it, int lane, float (&acc)[R], unsigned int& x) {
    if constexpr (LEVEL == 0) {
#pragma unroll
        for (int r = 0; r < R; ++r) x ^= v[r][0] ^ v[r][1] ^ v[r][2] ^ v[r][3];
    } else {
        float q[8];
        if constexpr (LEVEL >= 2) {
            const f32x4* qp = reinterpret_cast<const f32x4*>(q_lds + (it * 64 + lane) * 8);
            const f32x4 a = qp[0], b = qp[1];
            q[0] = a[0]; q[1] = a[1]; q[2] = a[2]; q[3] = a[3]; q[4] = b[0]; q[5] = b[1]; q[6] = b[2]; q[7] = b[3];
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) q[e] = 0.01f * (float)(lane + e + 1);
        }
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[r] = __builtin_fmaf(__uint_as_float(v[r][i] << 16), q[2 * i], acc[r]);
                acc[r] = __builtin_fmaf(__uint_as_float(v[r][i] & 0xffff0000u), q[2 * i + 1], acc[r]);
            }
    }
}

template <int R>
__device__ __forceinline__ float butterfly(float (&vals)[R], int lane) {
    int half = R / 2;
    for (int mask = 32; half >= 1; mask >>= 1, half >>= 1) {
        const bool upper = (lane & mask) != 0;
#pragma unroll
        for (int j = 0; j < R / 2; ++j) {
            if (j < half) {
                const float keep = upper ? vals[j + half] : vals[j];
                const float send = upper ? vals[j] : vals[j + half];
                vals[j] = keep + __shfl_xor(send, mask, 64);
            }
        }
    }
    int m = 32;
    for (int r = R; r > 1; r >>= 1) m >>= 1;
    for (; m >= 1; m >>= 1) vals[0] += __shfl_xor(vals[0], m, 64);
    return vals[0];
}

template <int R, int LEVEL, int PF>
__global__ void scan_like(const char* __restrict__ src, long long n_rows, int row_bytes, unsigned int* out) {
    extern __shared__ float q_lds[];
    const int lane = threadIdx.x & 63;
    const int waves = blockDim.x >> 6;
    for (int i = threadIdx.x; i < row_bytes / 2; i += blockDim.x) q_lds[i] = 0.001f * (float)(i % 97);
    __syncthreads();
    const long long ngroups = n_rows / R;
    const int nit = row_bytes / 1024;   // 2 here
    const long long gstride = (long long)gridDim.x * waves;
    unsigned int x = 0;
    float best = 1e30f;
    long long g = (long long)blockIdx.x * waves + (threadIdx.x >> 6);
    if constexpr (PF == 0) {
        for (; g < ngroups; g += gstride) {
            const char* base = src + g * R * (long long)row_bytes;
            float acc[R];
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] = 0.f;
            for (int it = 0; it < nit; ++it) {
                u32x4 v[R];
                load_group<R>(base, row_bytes, it, lane, v);
                consume<R, LEVEL>(v, q_lds, it, lane, acc, x);
            }
            if constexpr (LEVEL >= 3) {
                const float t = butterfly<R>(acc, lane);
                if (__any(t > best)) x += 1;
            } else if constexpr (LEVEL >= 1) {
#pragma unroll
                for (int r = 0; r < R; ++r) x ^= __float_as_uint(acc[r]);
            }
        }
    } else {
        // two sweeps per row (row_bytes = 2048): buffers A (sweep 0) and B (sweep 1); the next
        // group's sweep is requested as soon as the buffer's previous contents have been consumed
        u32x4 va[R], vb[R];
        if (g < ngroups) {
            load_group<R>(src + g * R * (long long)row_bytes, row_bytes, 0, lane, va);
            load_group<R>(src + g * R * (long long)row_bytes, row_bytes, 1, lane, vb);
        }
        for (; g < ngroups; g += gstride) {
            const long long gn = g + gstride < ngroups ? g + gstride : g;   // the last group re-reads itself
            const char* nbase = src + gn * R * (long long)row_bytes;
            float acc[R];
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] = 0.f;
            consume<R, LEVEL>(va, q_lds, 0, lane, acc, x);
            load_group<R>(nbase, row_bytes, 0, lane, va);
            consume<R, LEVEL>(vb, q_lds, 1, lane, acc, x);
            load_group<R>(nbase, row_bytes, 1, lane, vb);
            if constexpr (LEVEL >= 3) {
                const float t = butterfly<R>(acc, lane);
                if (__any(t > best)) x += 1;
            } else if constexpr (LEVEL >= 1) {
#pragma unroll
                for (int r = 0; r < R; ++r) x ^= __float_as_uint(acc[r]);
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) x ^= va[r][0] ^ vb[r][0];
    }
    if (x == 0x12345678u) out[blockIdx.x] = x;
}

template <int R, int LEVEL, int PF>
void run(const char* buf, long long n_rows, int row_bytes, unsigned int* out, int cus, int waves, int per_cu, hipEvent_t e0, hipEvent_t e1) {
    float sum = 0.f, best = 1e9f;
    const int reps = 10;
    for (int r = 0; r < reps + 2; ++r) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((scan_like<R, LEVEL, PF>), dim3(cus * per_cu), dim3(waves * 64), row_bytes * 2, 0, buf, n_rows, row_bytes, out);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (r >= 2) { sum += ms; best = ms < best ? ms : best; }
    }
    const double bytes = (double)(n_rows / R * R) * row_bytes;
    printf("{\"rows_per_wave\": %d, \"level\": %d, \"prefetch\": %d, \"waves\": %d, \"wg_per_cu\": %d, \"ms\": %.4f, \"ms_min\": %.4f, \"GBps\": %.1f}\n",
           R, LEVEL, PF, waves, per_cu, sum / reps, best, bytes / (sum / reps) / 1e6);
    fflush(stdout);
}

template <int R, int PF>
void levels(const char* buf, long long n_rows, int row_bytes, unsigned int* out, int cus, int waves, int per_cu, hipEvent_t e0, hipEvent_t e1) {
    run<R, 0, PF>(buf, n_rows, row_bytes, out, cus, waves, per_cu, e0, e1);
    run<R, 1, PF>(buf, n_rows, row_bytes, out, cus, waves, per_cu, e0, e1);
    run<R, 2, PF>(buf, n_rows, row_bytes, out, cus, waves, per_cu, e0, e1);
    run<R, 3, PF>(buf, n_rows, row_bytes, out, cus, waves, per_cu, e0, e1);
}

int main(int argc, char** argv) {
    const int row_bytes = 2048;
    const long long n_rows = 10000000;
    const int constant_data = argc > 1 ? atoi(argv[1]) : 0;
    char* buf = nullptr;
    unsigned int* out = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&buf), n_rows * row_bytes) != hipSuccess) return 1;
    if (hipMalloc(reinterpret_cast<void**>(&out), 1 << 20) != hipSuccess) return 1;
    if (constant_data) (void)hipMemset(buf, 1, n_rows * row_bytes);
    else hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, reinterpret_cast<unsigned int*>(buf), (size_t)n_rows * row_bytes / 4);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("{\"data\": \"%s\"}\n", constant_data ? "constant" : "random bf16");
    levels<8, 0>(buf, n_rows, row_bytes, out, cus, 8, 1, e0, e1);
    levels<8, 1>(buf, n_rows, row_bytes, out, cus, 8, 1, e0, e1);
    levels<8, 1>(buf, n_rows, row_bytes, out, cus, 4, 1, e0, e1);
    levels<16, 0>(buf, n_rows, row_bytes, out, cus, 4, 1, e0, e1);
    levels<16, 1>(buf, n_rows, row_bytes, out, cus, 4, 1, e0, e1);
    levels<16, 0>(buf, n_rows, row_bytes, out, cus, 8, 1, e0, e1);
    levels<4, 1>(buf, n_rows, row_bytes, out, cus, 8, 1, e0, e1);
    levels<4, 1>(buf, n_rows, row_bytes, out, cus, 16, 1, e0, e1);
    return 0;
}
