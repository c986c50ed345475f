// Encoder epilogue: masked mean over tokens followed by L2 normalisation, one
// pass over the hidden states, written as fp32 rows ready for the search.
//
// Replaces the five torch ops at pipeline/utils/local_rag.py:61-65 (and
// pipeline/utils/embed.py:68-72 on the index-build side):
//     expanded_mask = mask.unsqueeze(-1).expand(h.size()).float()
//     sum = (h * expanded_mask).sum(1);  cnt = clamp(expanded_mask.sum(1), min=1e-9)
//     v = sum / cnt;  v = F.normalize(v, p=2, dim=1)          (eps 1e-12)
// The encoder forward itself stays a PyTorch(-ROCm) call.
//
// One workgroup per sequence.  Threads own columns (coalesced over d) and walk
// the T tokens; masked-out tokens are skipped without loading.  Accumulation is
// fp32 in token order, which is the order torch.sum reduces a (T, d) slice for
// short T; results agree with the torch expression to ~1e-6.
//
// Algorithmic bytes: sum over sequences of (unmasked tokens) * d * sizeof(T).
#include "mtr_device.h"

namespace mtr {

namespace {

template <typename T> __device__ __forceinline__ float load_as_f32(const T* p);
template <> __device__ __forceinline__ float load_as_f32<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float load_as_f32<bf16_t>(const bf16_t* p) { return bf16_to_f32(*p); }
template <> __device__ __forceinline__ float load_as_f32<_Float16>(const _Float16* p) { return (float)*p; }

constexpr int kPoolThreads = 256;
constexpr int kMaxCols = 32;  // columns per thread: d <= 256 * 32

template <typename T, int NC>
__global__ __launch_bounds__(kPoolThreads) void pool_normalize_kernel(const T* __restrict__ hidden,
                                                                       const long long* __restrict__ mask, int t, int d,
                                                                       float* __restrict__ out) {
    __shared__ float red[kPoolThreads / 64];
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const T* h = hidden + (size_t)b * t * d;
    const long long* m = mask + (size_t)b * t;
    float acc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] = 0.f;
    float count = 0.f;
    for (int tok = 0; tok < t; ++tok) {
        const float w = (float)m[tok];  // wave-uniform; the reference multiplies by the mask value
        if (w == 0.f) continue;
        count += w;
        const T* row = h + (size_t)tok * d;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int col = c * kPoolThreads + tid;
            if (col < d) acc[c] = __builtin_fmaf(load_as_f32<T>(row + col), w, acc[c]);
        }
    }
    const float denom = fmaxf(count, 1e-9f);
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        acc[c] = acc[c] / denom;
        ss += acc[c] * acc[c];
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) ss += __shfl_xor(ss, off, 64);
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    float total = 0.f;
#pragma unroll
    for (int w = 0; w < kPoolThreads / 64; ++w) total += red[w];
    const float inv = 1.f / fmaxf(sqrtf(total), 1e-12f);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int col = c * kPoolThreads + tid;
        if (col < d) out[(size_t)b * d + col] = acc[c] * inv;
    }
}

template <typename T>
void launch_typed(const void* hidden, const long long* mask, int b, int t, int d, float* out, hipStream_t stream) {
    const T* h = static_cast<const T*>(hidden);
    if (d <= kPoolThreads * 4)
        hipLaunchKernelGGL((pool_normalize_kernel<T, 4>), dim3(b), dim3(kPoolThreads), 0, stream, h, mask, t, d, out);
    else if (d <= kPoolThreads * 16)
        hipLaunchKernelGGL((pool_normalize_kernel<T, 16>), dim3(b), dim3(kPoolThreads), 0, stream, h, mask, t, d, out);
    else
        hipLaunchKernelGGL((pool_normalize_kernel<T, kMaxCols>), dim3(b), dim3(kPoolThreads), 0, stream, h, mask, t, d,
                           out);
}

}  // namespace

hipError_t launch_pool_normalize(const void* hidden, int dtype, const long long* mask, int b, int t, int d, float* out,
                                 hipStream_t stream) {
    if (b < 1 || t < 1 || d < 1 || d > kPoolThreads * kMaxCols) return hipErrorInvalidValue;
    switch (dtype) {
        case 0: launch_typed<float>(hidden, mask, b, t, d, out, stream); break;
        case 2: launch_typed<bf16_t>(hidden, mask, b, t, d, out, stream); break;
        case 3: launch_typed<_Float16>(hidden, mask, b, t, d, out, stream); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace mtr
