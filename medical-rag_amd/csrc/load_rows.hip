// Row loader: turns caller rows (fp32 / fp64 / bf16) into the stored form --
// finite-checked, optionally L2-normalised, cast to fp32 or bf16, row stride
// padded with zeros.  Also prepares query batches (same arithmetic, fp32 out).
//
// This is the one-time replacement for what the reference redoes on every
// query: sklearn's normalize() of all N rows inside cosine_similarity
// (pipeline/utils/local_rag.py:77).  The norm follows scikit-learn: float64
// sqrt of the float64 sum of squares, and norms below 10*DBL_EPSILON count
// as 1 so zero rows stay zero (score 0, never NaN).
#include <float.h>
#include "mtr_device.h"

namespace mtr {

namespace {

template <typename T> __device__ __forceinline__ double to_f64(T v);
template <> __device__ __forceinline__ double to_f64<float>(float v) { return (double)v; }
template <> __device__ __forceinline__ double to_f64<double>(double v) { return v; }
template <> __device__ __forceinline__ double to_f64<bf16_t>(bf16_t v) { return (double)bf16_to_f32(v); }

template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return f32_to_bf16(v); }

__device__ __forceinline__ double wave_sum_f64(double x) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        int lo = __double2loint(x), hi = __double2hiint(x);
        lo = __shfl_xor(lo, m, 64);
        hi = __shfl_xor(hi, m, 64);
        x += __hiloint2double(hi, lo);
    }
    return x;
}

// One wave per row, grid-strided.
template <typename SrcT, typename DstT>
__global__ __launch_bounds__(256) void load_rows_kernel(const SrcT* __restrict__ src, long long n_rows, int d,
                                                         DstT* __restrict__ dst, int ld, int normalize,
                                                         NonfiniteFlag nonfinite_flag) {
    const int lane = threadIdx.x & 63;
    const long long wave_id = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long long nwaves = (long long)gridDim.x * (blockDim.x >> 6);
    for (long long row = wave_id; row < n_rows; row += nwaves) {
        const SrcT* s = src + row * d;
        double ss = 0.0;
        bool bad = false;
        // A lane's loads go out kBatch at a time and are summed in element order
        // afterwards (one at a time they cost a memory round trip each: 33 us for
        // 34 query rows of 4096, most of a small search).
        constexpr int kBatch = 16;
        for (int e0 = lane; e0 < d; e0 += 64 * kBatch) {
            SrcT v[kBatch];
#pragma unroll
            for (int u = 0; u < kBatch; ++u) {
                const int e = e0 + 64 * u;
                v[u] = s[e < d ? e : lane];   // lane < d whenever this loop runs
            }
#pragma unroll
            for (int u = 0; u < kBatch; ++u) {
                if (e0 + 64 * u < d) {
                    const double x = to_f64<SrcT>(v[u]);
                    bad |= !(fabs(x) <= DBL_MAX);  // NaN or infinity
                    ss += x * x;
                }
            }
        }
        if (__ballot(bad) != 0ull) {
            if (lane == 0) atomicMax(nonfinite_flag.word, nonfinite_flag.value);
        }
        double inv = 1.0;
        if (normalize) {
            double norm = sqrt(wave_sum_f64(ss));
            if (!(norm >= 10.0 * DBL_EPSILON)) norm = 1.0;
            inv = norm;
        }
        DstT* o = dst + row * ld;
        for (int e0 = lane; e0 < ld; e0 += 64 * kBatch) {
            SrcT v[kBatch];
#pragma unroll
            for (int u = 0; u < kBatch; ++u) {
                const int e = e0 + 64 * u;
                v[u] = s[e < d ? e : 0];   // d >= 1
            }
#pragma unroll
            for (int u = 0; u < kBatch; ++u) {
                const int e = e0 + 64 * u;
                if (e < ld) {
                    float r = 0.f;
                    if (e < d) {
                        const double x = to_f64<SrcT>(v[u]);
                        r = (float)(normalize ? x / inv : x);
                    }
                    o[e] = from_f32<DstT>(r);
                }
            }
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void read_rows_kernel(const T* __restrict__ rows, long long n_rows, int d, int ld,
                                                         float* __restrict__ dst) {
    const long long total = n_rows * d;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / d;
        const int e = (int)(i - r * d);
        if constexpr (sizeof(T) == 2) dst[i] = bf16_to_f32(rows[r * ld + e]);
        else dst[i] = rows[r * ld + e];
    }
}

__global__ __launch_bounds__(256) void fill_u32_kernel(unsigned int* __restrict__ dst, unsigned int value, size_t n_words) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = value;
}

template <typename SrcT>
hipError_t launch_src(const void* src, int64_t n_rows, int d, void* dst, int dst_dtype, int ld, int normalize,
                      NonfiniteFlag flag, hipStream_t stream) {
    long long blocks = (n_rows + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    if (dst_dtype == 0)
        hipLaunchKernelGGL((load_rows_kernel<SrcT, float>), dim3((unsigned)blocks), dim3(256), 0, stream,
                           static_cast<const SrcT*>(src), (long long)n_rows, d, static_cast<float*>(dst), ld,
                           normalize, flag);
    else if (dst_dtype == 2)
        hipLaunchKernelGGL((load_rows_kernel<SrcT, bf16_t>), dim3((unsigned)blocks), dim3(256), 0, stream,
                           static_cast<const SrcT*>(src), (long long)n_rows, d, static_cast<bf16_t*>(dst), ld,
                           normalize, flag);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace

hipError_t launch_load_rows(const void* src, int src_dtype, int64_t n_rows, int d, void* dst, int dst_dtype, int ld,
                            int normalize, NonfiniteFlag nonfinite_flag, hipStream_t stream) {
    if (n_rows <= 0) return hipSuccess;
    switch (src_dtype) {
        case 0: return launch_src<float>(src, n_rows, d, dst, dst_dtype, ld, normalize, nonfinite_flag, stream);
        case 1: return launch_src<double>(src, n_rows, d, dst, dst_dtype, ld, normalize, nonfinite_flag, stream);
        case 2: return launch_src<bf16_t>(src, n_rows, d, dst, dst_dtype, ld, normalize, nonfinite_flag, stream);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_fill_u32(void* dst, unsigned int value, size_t n_words, hipStream_t stream) {
    if (n_words == 0) return hipSuccess;
    size_t blocks = (n_words + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(fill_u32_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, static_cast<unsigned int*>(dst), value,
                       n_words);
    return hipGetLastError();
}

hipError_t launch_read_rows(const void* rows, int dtype, int64_t n_rows, int d, int ld, float* dst,
                            hipStream_t stream) {
    if (n_rows <= 0) return hipSuccess;
    long long blocks = (n_rows * d + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (dtype == 0)
        hipLaunchKernelGGL((read_rows_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, stream,
                           static_cast<const float*>(rows), (long long)n_rows, d, ld, dst);
    else
        hipLaunchKernelGGL((read_rows_kernel<bf16_t>), dim3((unsigned)blocks), dim3(256), 0, stream,
                           static_cast<const bf16_t*>(rows), (long long)n_rows, d, ld, dst);
    return hipGetLastError();
}

}  // namespace mtr
